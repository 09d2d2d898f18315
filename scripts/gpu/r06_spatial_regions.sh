#!/bin/bash
# Round 6: the queue region of a wave's rays = the octant of its first live ray (PRV_SPATIAL_REGIONS=1, the default) against the block
# index modulo the region count (0: rounds 1-5) -- every bench kernel, the reference's round, field_hbm; then the parity tests.
#   usage: scripts/gpu/r06_spatial_regions.sh <tag>     -> gpurun_out/<tag>/ab.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06n}; mkdir -p $O
for v in 1 0 1 0; do
  echo "== PRV_SPATIAL_REGIONS=$v" | tee -a $O/ab.txt
  for cfg in "256 baseline" "256 dense" "512 baseline" "512 dense"; do set -- $cfg
    PRV_SPATIAL_REGIONS=$v python3 scripts/kbench.py --reps 5 --field $1 --scene $2 --tag "f$1_$2" 2>&1 | grep -v amdgpu.ids | cut -c1-260 | tee -a $O/ab.txt
  done
  PRV_SPATIAL_REGIONS=$v python3 scripts/kbench.py --reps 3 --field hbm --views 16 --tag hbm 2>&1 | grep -v amdgpu.ids | cut -c1-260 | tee -a $O/ab.txt
  PRV_SPATIAL_REGIONS=$v python3 scripts/kbench.py --reps 5 --field 256 --scene baseline --step ngp --tag ngp256 2>&1 | grep -v amdgpu.ids | cut -c1-260 | tee -a $O/ab.txt
  PRV_SPATIAL_REGIONS=$v python3 scripts/refbench.py --step ngp --reps 3 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tee -a $O/ab.txt
done
timeout 2400 python3 -m pytest tests/test_gpu_sweep.py tests/test_gpu_parity.py tests/test_gpu_wholeview.py tests/test_gpu_ngp_step.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/pytest.txt
