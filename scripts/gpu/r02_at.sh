cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02at
mkdir -p $O
python3 scripts/trainablate.py --save /tmp/state.prvf --rays 65536 2>&1 | tail -1
for HK in 0 1; do
if [ $HK = 1 ]; then export PRV_TRAIN_BWD_HALF=1; fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$HK -- python3 scripts/trainablate.py --load /tmp/state.prvf --rays 65536 --tag half$HK > $O/abl$HK.txt 2>&1
f=$(find $O/prof$HK -name "*kernel_stats.csv" | head -1); cp "$f" $O/abl_kernel_stats_$HK.csv
grep "^half" $O/abl$HK.txt
python3 - <<PY
import csv
for r in list(csv.DictReader(open('$O/abl_kernel_stats_$HK.csv')))[:2]:
    print(f"half=$HK {r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
done
