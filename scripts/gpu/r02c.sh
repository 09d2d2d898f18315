set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02c
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_comm.py tests/test_gpu_planner.py tests/test_gpu_configs.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
