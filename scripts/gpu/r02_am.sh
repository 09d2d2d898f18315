cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02am
mkdir -p $O
python3 scripts/trainbench.py --rays 65536 --steps 1500 > $O/train.txt 2>&1; tail -4 $O/train.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 scripts/trainbench.py --rays 65536 --steps 800 > $O/train_prof.txt 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-170; cp "$f" $O/train_kernel_stats.csv
