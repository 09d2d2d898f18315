cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bi
mkdir -p $O
for i in $(seq 1 12); do
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_sweep.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q --timeout 600 -k "train or configs" -p no:cacheprovider > $O/pytest_$i.log 2>&1; echo "run $i rc=$? $(grep -o '[0-9]* passed\|[0-9]* failed' $O/pytest_$i.log | tr '\n' ' ')"
done
