cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02br
mkdir -p $O
python3 scripts/refbench.py 2>&1 | tail -3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 scripts/refbench.py > $O/ref.txt 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
for r in list(csv.DictReader(open('$f')))[:12]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
