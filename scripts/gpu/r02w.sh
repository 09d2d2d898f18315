cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02w
mkdir -p $O
PRV_ABLATE=8 python3 nerf_prv_amd/build.py --force > $O/build.log 2>&1
python3 scripts/kbench.py --reps 5 --tag noloop 2>&1 | grep noloop | tee -a $O/kbench.txt
PRV_ABLATE= python3 nerf_prv_amd/build.py --force > /dev/null 2>&1
