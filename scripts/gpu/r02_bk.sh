cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bk
mkdir -p $O
F='s/eval_exact=[0-9]* //; s/RM=8 rounds.*render=/render=/; s/kernel_rate.*ns/ns/'
for i in 1 2; do python3 scripts/kbench.py --reps 8 --tag base 2>&1 | grep "^base" | sed "$F" | tee -a $O/kbench.txt; done
for N in 24 16; do
PRV_W_REGS=$N PRV_R64_WAVES=2 python3 nerf_prv_amd/build.py --force > $O/build.log 2>&1
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 2>&1 | tail -1
for i in 1 2; do PRV_BLOCKS_PER_CU=2 python3 scripts/kbench.py --reps 8 --tag wregs$N 2>&1 | grep "^wregs" | sed "$F" | tee -a $O/kbench.txt; done
PRV_BLOCKS_PER_CU=2 python3 scripts/kbench.py --reps 5 --field 512 --tag wregs${N}_512 2>&1 | grep "^wregs" | sed "$F" | tee -a $O/kbench.txt
done
python3 nerf_prv_amd/build.py --force > /dev/null 2>&1
python3 scripts/kbench.py --reps 5 --field 512 --tag base512 2>&1 | grep "^base" | sed "$F" | tee -a $O/kbench.txt
