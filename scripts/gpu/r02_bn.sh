cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bn
mkdir -p $O
F='s/eval_exact=[0-9]* //; s/RM=8 rounds.*render=/render=/; s/kernel_rate.*ns/ns/'
run() { python3 scripts/kbench.py --reps 5 --field 512 "$@" 2>&1 | grep "BPC=" | sed "$F" | tee -a $O/f512.txt; }
run --tag base_auto
PRV_MERGE_MAX=16 PRV_POOL=1 run --tag merge_pool
PRV_MERGE_MAX=16 PRV_POOL=0 run --tag merge_only
for W in 1 2; do
PRV_R64_WAVES=$W python3 nerf_prv_amd/build.py --force > $O/build.log 2>&1
run --tag waves_per_eu_$W
PRV_BLOCKS_PER_CU=2 run --tag waves_per_eu_${W}_bpc2
done
python3 nerf_prv_amd/build.py --force > /dev/null 2>&1
