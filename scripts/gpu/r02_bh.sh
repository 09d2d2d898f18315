cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bh
mkdir -p $O
env | grep -i "rocp\|preload" | head
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-extras --no-training > $O/bench_prof.json 2> $O/bench_prof.err; echo "rc=$?"
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -4 "$f"; cp "$f" $O/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -- python3 -c "import os; print({k:v for k,v in os.environ.items() if 'ROCP' in k.upper() or 'PRELOAD' in k})" 2>/dev/null | tail -2
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
