cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r06l; mkdir -p $O
for v in "" "PRV_TRAIN_BWD_BLOCKS=512" "PRV_TRAIN_BWD_BLOCKS=384" "PRV_TRAIN_FWD_BLOCKS=1024" "PRV_TRAIN_FWD_BLOCKS=256"; do
echo "== $v" | tee -a $O/ab.txt
env $v timeout 300 python3 scripts/trainprofile.py --rule ngp --rays 65536 --members 5 --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/ab.txt
env $v timeout 300 python3 scripts/trainprofile.py --rule ngp --rays 65536 --members 1 --steps 1500 --chunk 500 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $O/ab.txt
done
