set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02b
mkdir -p $O
python3 scripts/kbench.py --reps 1 --tag f256 > $O/kbench_256.txt 2>&1
python3 scripts/kbench.py --reps 1 --field 512 --tag f512 > $O/kbench_512.txt 2>&1
bash scripts/pmc.sh $O/pmc256 1,2,3,4,5,6 > $O/pmc256.log 2>&1
bash scripts/pmc.sh $O/pmc512 1,2,4,5,6 --field 512 > $O/pmc512.log 2>&1
cat $O/kbench_256.txt $O/kbench_512.txt
tail -30 $O/pmc256/summary.txt
