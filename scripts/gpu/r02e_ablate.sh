cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02e
mkdir -p $O
for ABL in 0 4 2 1 6; do
  PRV_ABLATE=$ABL python3 nerf_prv_amd/build.py --force > $O/build_$ABL.log 2>&1
  for i in 1 2; do python3 scripts/kbench.py --reps 5 --tag ablate$ABL >> $O/kbench.txt 2>&1; done
done
PRV_ABLATE= python3 nerf_prv_amd/build.py --force > /dev/null 2>&1
grep ablate $O/kbench.txt
