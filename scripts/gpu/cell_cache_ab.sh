# A/B of the render kernel's per-lane corner cache (PRV_CELL_CACHE) and the relocation knobs on the reference's round
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r04cache}
mkdir -p $O
run() { echo "== $*" | tee -a $O/ab.txt; env "$@" python3 scripts/refbench.py --step ngp --reps 3 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt; }
run PRV_CELL_CACHE=0
run PRV_CELL_CACHE=1
run PRV_CELL_CACHE=1 PRV_MERGE_MAX=16 PRV_POOL=1
run PRV_CELL_CACHE=1 PRV_MERGE_MAX=24 PRV_POOL=1
run PRV_CELL_CACHE=1 PRV_MERGE_MAX=31 PRV_POOL=1
run PRV_CELL_CACHE=1 PRV_MERGE_MAX=16 PRV_POOL=0
run PRV_CELL_CACHE=1 PRV_MERGE_MAX=16 PRV_POOL=1 PRV_BLOCKS_PER_CU=1
echo "== fixed-S, cache on + merge" | tee -a $O/ab.txt
PRV_CELL_CACHE=1 PRV_MERGE_MAX=16 PRV_POOL=1 python3 scripts/refbench.py --step fixed --reps 3 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
PRV_CELL_CACHE=0 PRV_MERGE_MAX=16 PRV_POOL=1 python3 scripts/refbench.py --step fixed --reps 3 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
