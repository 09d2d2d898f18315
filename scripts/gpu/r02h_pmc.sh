cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02h
mkdir -p $O
export PRV_BLOCKS_PER_CU=3
python3 scripts/kbench.py --reps 1 --tag f256 > $O/kbench_256.txt 2>&1
bash scripts/pmc.sh $O/pmc256 1,2,3 > $O/pmc256.log 2>&1
cat $O/kbench_256.txt | grep f256
grep -A30 "render_queue64" $O/pmc256/summary.txt
