cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bt
mkdir -p $O
python3 scripts/trainablate.py --save /tmp/state.prvf --rays 65536 2>&1 | tail -1
PMC_SCRIPT=scripts/trainablate.py bash scripts/pmc.sh $O/pmc 8 --load /tmp/state.prvf --rays 65536 --tag pmc > $O/pmc.log 2>&1
grep -A10 "train_tile_kernel<4, false, 2>" $O/pmc/summary.txt | head -12
