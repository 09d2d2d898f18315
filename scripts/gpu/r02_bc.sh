cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bc
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_sweep.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -k "train" > $O/pytest.log 2>&1; tail -15 $O/pytest.log
for K in 1 0; do
export PRV_TRAIN_REG_CHAIN=$K
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$K -- python3 scripts/trainbench.py --rays 65536 --steps 600 > $O/train$K.txt 2>&1
f=$(find $O/prof$K -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
for r in list(csv.DictReader(open('$f')))[:3]:
    print(f"chain=$K {r['Name'][:60]:60s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
grep "steps in" $O/train$K.txt
done
