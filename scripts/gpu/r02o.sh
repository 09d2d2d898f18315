cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02o
mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-extras --no-training > $O/bench_prof.json 2> $O/bench_prof.err; echo "rc=$?" >> $O/bench_prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -8 "$f"; cp "$f" $O/kernel_stats.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof512 -- python3 bench.py --field 512 --no-cpu-baseline --no-extras --no-training > $O/bench_prof512.json 2> $O/bench_prof512.err; echo "rc=$?" >> $O/bench_prof512.err
f=$(find $O/prof512 -name "*kernel_stats.csv" | head -1); head -5 "$f"; cp "$f" $O/kernel_stats_512.csv
PRV_FORCE_DIST=1 timeout 600 python3 bench.py --no-cpu-baseline --no-training --no-extras --steps 5 > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "rc=$?" >> $O/bench_dist1.err
wc -l $O/bench_dist1.json
