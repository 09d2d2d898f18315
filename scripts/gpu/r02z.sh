cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02z
mkdir -p $O
for A in 32 64 128 160 16 48 176; do
PRV_ABLATE=$A python3 nerf_prv_amd/build.py --force > $O/build.log 2>&1
python3 scripts/kbench.py --reps 5 --tag ablate$A 2>&1 | grep ablate | sed 's/eval_exact.*render=/render=/' | tee -a $O/kbench2.txt
done
PRV_ABLATE= python3 nerf_prv_amd/build.py --force > /dev/null 2>&1
