cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r06m; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o tb -- python3 $GRAFT_REPO_ROOT/scripts/trainprofile.py --rule ngp --rays 65536 --members 1 --steps 1500 --chunk 500 > $O/trace_ngp.txt 2>&1)
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_loop1_ngp.csv
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 300 steps: per-kernel mean duration and mean gap to the previous kernel's end
tail = rows[-300 * 7:]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None
for r in tail:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("prv::", "").split("(")[0][:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[k].append(e - s)
    if prev_end is not None: gap[k].append(s - prev_end)
    prev_end = e
for k in dur:
    print(f"{k:42s} n={len(dur[k]):5d} mean {sum(dur[k])/len(dur[k])/1e3:8.1f} us   gap before {sum(gap[k])/max(1,len(gap[k]))/1e3:6.1f} us")
PY
rm -rf $O/prof
