cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bm
mkdir -p $O
F='s/eval_exact=[0-9]* //; s/RM=8 rounds.*render=/render=/; s/kernel_rate.*ns/ns/'
run() { python3 scripts/kbench.py --reps 5 "$@" 2>&1 | grep "BPC=" | sed "$F" | tee -a $O/bpc.txt; }
for B in auto 1 2 3 4; do
if [ $B = auto ]; then unset PRV_BLOCKS_PER_CU; else export PRV_BLOCKS_PER_CU=$B; fi
run --tag "256^3_800x800"
run --field 512 --tag "512^3_800x800"
run --views 540 --width 80 --height 45 --spp 16 --min-t 0.01 --tag "256^3_80x45x16spp_540views"
run --views 64 --size 320 --tag "256^3_320x320"
run --bias 0 --tag "256^3_800x800_density_bias_0"
done
unset PRV_BLOCKS_PER_CU
timeout 600 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q -x --timeout 600 2>&1 | tail -2
