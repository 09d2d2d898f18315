cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bv
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > $O/pytest.log 2>&1; grep "passed\|failed" $O/pytest.log | tail -2
timeout 600 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 -c "
import json;d=json.load(open('$O/bench.json'));print(round(d['value']/1e9,2),round(d['ms_per_step'],3), d['training']['steps_per_s'], d['training']['samples_per_s']/1e6, d['field512']['value']/1e9)"
python3 scripts/nbvbench.py 2>&1 | tail -2 | head -1
