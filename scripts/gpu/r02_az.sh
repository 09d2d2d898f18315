cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02az
mkdir -p $O
for i in 1 2 3; do
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > $O/pytest_$i.log 2>&1; echo "run $i rc=$? $(grep -o '[0-9]* passed\|[0-9]* failed' $O/pytest_$i.log | tr '\n' ' ')"
done
timeout 600 python3 bench.py --no-cpu-baseline --no-training --no-extras > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 -c "
import json;d=json.load(open('$O/bench.json'));print(round(d['value']/1e9,2),round(d['ms_per_step'],3))"
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
