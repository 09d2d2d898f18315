cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r06g; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -3 | tee $O/pytest_train.txt
for rule in ngp fixed; do
timeout 300 python3 scripts/trainprofile.py --rule $rule --rays 65536 --members 5 --steps 2500 --chunk 500 --eval 2>&1 | grep -v amdgpu.ids | tee $O/loop_$rule.txt | tail -7
timeout 300 python3 scripts/trainprofile.py --rule $rule --rays 65536 --members 1 --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tee $O/loop1_$rule.txt | tail -2
timeout 300 python3 scripts/trainbench.py --rule $rule --eval-rule ngp --rays 65536 --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tee $O/trainbench_$rule.txt | tail -2
done
W=/tmp/loopwork; mkdir -p $W
sed -e "s#pre_path: \"[^\"]*\"#pre_path: \"$W/\"#" -e "s#model_path: \"[^\"]*\"#model_path: \"$W/models/\"#" -e "s#viewspace_path: \"[^\"]*\"#viewspace_path: \"$GRAFT_REPO_ROOT/tests/golden/hemisphere/\"#" configs/TrainInLoop.yaml > $W/cfg.yaml
echo -e "21\nobject_0\n-1" | PRV_PLANNER_TIMING=1 PRV_TRAIN_TIMING=1 timeout 900 nerf_prv_amd/prv_planner $W/cfg.yaml > $O/loop_stdout.txt 2> $O/loop_stderr.txt
grep -E "train_members" $O/loop_stderr.txt | cut -c1-200 | head -24
