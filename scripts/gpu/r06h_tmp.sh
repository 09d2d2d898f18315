cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r06h; mkdir -p $O
timeout 1800 python3 -m pytest tests/test_gpu_comm.py tests/test_gpu_planner.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | grep -v amdgpu.ids | tail -30 | tee $O/pytest.txt
