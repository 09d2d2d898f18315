cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bf
mkdir -p $O
python3 scripts/trainablate.py --save /tmp/state.prvf --rays 65536 2>&1 | tail -1
for HK in 0 70000 200000 600000; do
if [ $HK != 0 ]; then export PRV_TRAIN_REP_HACK=$HK; fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$HK -- python3 scripts/trainablate.py --load /tmp/state.prvf --rays 65536 --tag rep$HK > $O/abl$HK.txt 2>&1
f=$(find $O/prof$HK -name "*kernel_stats.csv" | head -1)
grep "^rep" $O/abl$HK.txt
python3 - <<PY
import csv
for r in list(csv.DictReader(open('$f')))[:1]:
    print(f"rep=$HK {r['Name'][:70]:70s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
done
