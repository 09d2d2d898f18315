cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_sweep.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
for R in 0 1; do for B in 2 3 4; do
PRV_RENDER64=$R PRV_BLOCKS_PER_CU=$B python3 scripts/kbench.py --reps 5 --tag r64=$R 2>&1 | grep "r64=" | tee -a $O/kbench.txt
done; done
PRV_RENDER64=0 python3 scripts/kbench.py --reps 5 --field 512 --tag f512_r64=0 2>&1 | grep "r64=" | tee -a $O/kbench.txt
PRV_RENDER64=1 PRV_BLOCKS_PER_CU=2 python3 scripts/kbench.py --reps 5 --field 512 --tag f512_r64=1 2>&1 | grep "r64=" | tee -a $O/kbench.txt
