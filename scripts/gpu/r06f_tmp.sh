cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r06f; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_gpu_planner.py tests/test_gpu_sweep.py tests/test_gpu_train.py tests/test_gpu_wholeview.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/pytest_gpu_rest.txt
W=/tmp/loopwork; mkdir -p $W
sed -e "s#pre_path: \"[^\"]*\"#pre_path: \"$W/\"#" -e "s#model_path: \"[^\"]*\"#model_path: \"$W/models/\"#" -e "s#viewspace_path: \"[^\"]*\"#viewspace_path: \"$GRAFT_REPO_ROOT/tests/golden/hemisphere/\"#" configs/TrainInLoop.yaml > $W/cfg.yaml
echo -e "21\nobject_0\n-1" | PRV_PLANNER_TIMING=1 PRV_TRAIN_TIMING=1 timeout 900 nerf_prv_amd/prv_planner $W/cfg.yaml > $O/loop_stdout.txt 2> $O/loop_stderr.txt
grep -E "train_members|last batch of slot 0|host enqueue" $O/loop_stderr.txt | head -70
