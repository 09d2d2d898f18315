cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r06final; mkdir -p $O
timeout 3000 python3 -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
bash scripts/gpu/collect_profiles.sh r06final/prof > $O/collect.log 2>&1; tail -3 $O/collect.log
