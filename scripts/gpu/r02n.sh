cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02n
mkdir -p $O
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err; echo "rc=$?" >> $O/bench_prof.err
PRV_FORCE_DIST=1 timeout 600 python3 bench.py --no-cpu-baseline --no-training --no-extras --steps 5 > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "rc=$?" >> $O/bench_dist1.err
wc -l $O/bench_dist1.json
find $O/prof -name "*kernel_stats.csv" | head
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -12 "$f"; cp "$f" $O/kernel_stats.csv
python3 -m pytest tests/test_gpu_sweep.py -m gpu -q --timeout 600 2>&1 | tail -3
