#!/bin/bash
# the whole GPU suite, then the default bench line        usage: scripts/gpu/r06_suite.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06e}; mkdir -p $O
timeout 2400 python3 -m pytest tests -x -q -m gpu -s 2>&1 | grep -v amdgpu.ids | tail -25 | tee $O/pytest_gpu.txt
timeout 1200 python3 bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc $?"; tail -c 400 $O/bench_err.txt
python3 - $O/bench_line.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in d if (k.startswith("full_loop") and k != "full_loop") or k.startswith("training_")}, d.get("value"), d.get("ms_per_step"), d.get("reference_round_ms"))
print(json.dumps(d.get("full_loop"))[:1600]); print(json.dumps(d.get("training"))[:2400])
PY
