cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02cc
mkdir -p $O
rm -rf prv_out
export PRV_PLANNER_TIMING=1 PRV_TRAIN_TIMING=1
echo -e "21\nsynthetic_object\n-1" | timeout 900 nerf_prv_amd/prv_planner configs/TrainInLoop.yaml > $O/loop.txt 2> $O/loop.err; echo "rc=$?"
grep "prv_train_steps" $O/loop.err | sed -n '1p;5p;10p;20p'
PRV_TRAIN_TIMING=1 python3 scripts/trainbench.py --rays 4096 --steps 2500 --members 5 2>&1 | grep "prv_train_steps\|ensemble" | tail -4
