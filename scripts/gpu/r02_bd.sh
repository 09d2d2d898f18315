cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bd
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep "passed\|failed" $O/pytest.log | tail -3; grep -B30 "^FAILED\|Error" $O/pytest.log | head -60
timeout 600 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 -c "
import json;d=json.load(open('$O/bench.json'));print(round(d['value']/1e9,2),round(d['ms_per_step'],3));print(d['training'])"
python3 scripts/trainbench.py --rays 65536 --steps 2500 2>&1 | tail -2
python3 scripts/trainbench.py --rays 4096 --steps 2500 --members 5 2>&1 | tail -3
python3 scripts/nbvbench.py > $O/nbv.txt 2>&1; tail -2 $O/nbv.txt
