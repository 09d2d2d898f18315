cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bw
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep "passed\|failed" $O/pytest.log | tail -3
python3 scripts/kbench.py --reps 1 --tag f256 > $O/kbench_256.txt 2>&1
bash scripts/pmc.sh $O/pmc256 1,2,3,4,5,6 > $O/pmc256.log 2>&1
grep f256 $O/kbench_256.txt
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-extras --no-training > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -4 "$f"; cp "$f" $O/kernel_stats.csv
