cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02ap
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 900 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 -k "train" >> $O/pytest2.log 2>&1; tail -3 $O/pytest2.log
for K in 1 0; do
PRV_TRAIN_KEEP_ACT=$K python3 scripts/trainbench.py --rays 65536 --steps 1500 2>&1 | tail -2 | sed "s/^/[keep=$K 65536] /" | tee -a $O/train.txt
PRV_TRAIN_KEEP_ACT=$K python3 scripts/trainbench.py --rays 4096 --steps 2500 --members 5 2>&1 | tail -3 | sed "s/^/[keep=$K 4096] /" | tee -a $O/train.txt
done
