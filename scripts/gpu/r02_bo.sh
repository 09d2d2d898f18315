cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bo
mkdir -p $O
F='s/eval_exact=[0-9]* //; s/RM=8 rounds.*render=/render=/; s/kernel_rate.*ns/ns/'
run() { python3 scripts/kbench.py --reps 5 --views 540 --width 80 --height 45 --spp 16 --min-t 0.01 "$@" 2>&1 | grep "BPC=" | sed "$F" | tee -a $O/ref.txt; }
run --tag base_auto
PRV_MERGE_MAX=16 PRV_POOL=1 run --tag merge_pool
PRV_MERGE_MAX=16 PRV_POOL=0 run --tag merge_only
PRV_MERGE_MAX=8 PRV_POOL=0 run --tag merge8_only
PRV_REFILL_MIN=16 run --tag refill16
PRV_REFILL_MIN=1 run --tag refill1
