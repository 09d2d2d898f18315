#!/bin/bash
# Round 6, first measurement: the trainer at upstream's batch (2^18-sample target, 2^16-ray cap) -- per-kernel times, atomic
# requests per composited sample, the patch-sampler study at THAT batch, then the default bench line (full_loop at the default batch).
#   usage: scripts/gpu/r06_big_batch.sh <tag>     -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06a}; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o tb -- python3 $GRAFT_REPO_ROOT/scripts/trainbench.py --rays 65536 --steps 600 --chunk 200 > $O/trainbench_trace.txt 2>&1)
tail -3 $O/trainbench_trace.txt
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_65536rays.csv; rm -rf $O/prof
head -14 $O/kernel_stats_65536rays.csv | cut -c1-160
bash scripts/gpu/train_patch_study.sh ${1:-r06a}/patch 1x1:65536 2x1:65536 2x2:65536 4x4:65536
timeout 900 python3 bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc $?"; tail -c 600 $O/bench_err.txt
python3 - $O/bench_line.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in d if k.startswith("full_loop") and k != "full_loop"}, d.get("value"), d.get("ms_per_step"), d.get("training_steps_per_s"), d.get("reference_round_ms"))
print(json.dumps(d.get("full_loop"))[:1500]); print(json.dumps(d.get("training"))[:1200])
PY
