cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02ad
mkdir -p $O
for A in 16 17 18; do
PRV_ABLATE=$A python3 nerf_prv_amd/build.py --force > $O/build.log 2>&1
python3 scripts/kbench.py --reps 5 --tag ablate$A 2>&1 | grep ablate | sed 's/eval_exact=[0-9]* //' | tee -a $O/kbench.txt
done
PRV_ABLATE= python3 nerf_prv_amd/build.py --force > /dev/null 2>&1
