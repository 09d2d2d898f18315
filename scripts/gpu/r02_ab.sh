cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02ab
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep "passed\|failed" $O/pytest.log | tail -3
for i in 1 2 3; do
timeout 600 python3 bench.py --no-cpu-baseline --no-training --no-extras > $O/bench_$i.json 2> $O/bench_$i.err
python3 -c "
import json;d=json.load(open('$O/bench_$i.json'));print(round(d['value']/1e9,2),round(d['ms_per_step'],3),round(d['roofline']['avg_launch_ms'],3),round(d['roofline']['march_avg_launch_ms'],3))"
done
