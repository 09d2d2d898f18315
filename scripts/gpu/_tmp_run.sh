cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -3
python3 scripts/trainprofile.py --views 5 --members 1 --steps 1500 --chunk 500 2>&1 | grep -v amdgpu.ids | tail -2
python3 scripts/trainprofile.py --views 5 --members 5 --steps 1500 --chunk 500 2>&1 | grep -v amdgpu.ids | tail -2
python3 scripts/trainbench.py --rays 65536 --steps 300 --chunk 100 2>&1 | grep -v amdgpu.ids | tail -3
