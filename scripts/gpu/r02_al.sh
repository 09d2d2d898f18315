cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "awkward" 2>&1 | tail -15
