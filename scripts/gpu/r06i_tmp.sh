cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r06i; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_train.py tests/test_gpu_sweep.py -x -q 2>&1 | tail -3 | tee $O/pytest_train.txt
for v in "" "PRV_TRAIN_CHAIN_ROTATE=0" "PRV_TRAIN_TILE_SKIP=0"; do
echo "== $v" | tee -a $O/ab.txt
env $v timeout 300 python3 scripts/trainprofile.py --rule ngp --rays 65536 --members 1 --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/ab.txt
env $v timeout 300 python3 scripts/trainprofile.py --rule ngp --rays 65536 --members 5 --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/ab.txt
done
timeout 300 python3 scripts/trainbench.py --rule ngp --eval-rule ngp --rays 65536 --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/ab.txt
timeout 300 python3 scripts/trainbench.py --rule fixed --eval-rule ngp --rays 65536 --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/ab.txt
