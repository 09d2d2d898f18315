cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bb
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_sweep.py -m gpu -q -x --timeout 600 -k "train" 2>&1 | tail -2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 scripts/trainbench.py --rays 65536 --steps 400 > $O/train.txt 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
for r in list(csv.DictReader(open('$f')))[:6]:
    print(f"{r['Name'][:60]:60s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
tail -2 $O/train.txt
