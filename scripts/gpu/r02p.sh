cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02p
mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt; tail -2 $O/smoke.txt
timeout 600 python3 scripts/nbvbench.py > $O/nbvbench.txt 2>&1; tail -5 $O/nbvbench.txt
( time (echo -e "21\nsynthetic_object\n-1" | timeout 900 nerf_prv_amd/prv_planner configs/TrainInLoop.yaml) ) > $O/loop.txt 2>&1; tail -8 $O/loop.txt
timeout 300 python3 scripts/refbench.py > $O/refbench.txt 2>&1; tail -4 $O/refbench.txt
