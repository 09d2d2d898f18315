cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r06s; mkdir -p $O
for v in 1 0 1 0; do
  echo "== PRV_SPATIAL_REGIONS=$v" | tee -a $O/ab.txt
  for cfg in "256 baseline" "256 dense" "512 baseline"; do set -- $cfg
    PRV_SPATIAL_REGIONS=$v python3 scripts/kbench.py --reps 5 --field $1 --scene $2 --tag "f$1_$2" 2>&1 | grep -v amdgpu.ids | cut -c1-40,100-180 | tee -a $O/ab.txt
  done
  PRV_SPATIAL_REGIONS=$v python3 scripts/kbench.py --reps 5 --field 256 --scene baseline --step ngp --tag ngp256 2>&1 | grep -v amdgpu.ids | cut -c1-40,100-180 | tee -a $O/ab.txt
  PRV_SPATIAL_REGIONS=$v python3 scripts/refbench.py --step ngp --reps 3 2>&1 | grep -v amdgpu.ids | cut -c40-110,160-290 | tee -a $O/ab.txt
done
