cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02k
mkdir -p $O
export PRV_BLOCKS_PER_CU=3
PRV_MERGE_MAX=12 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sweep.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for M in 0 4 8 12 16 24 31; do
PRV_MERGE_MAX=$M python3 scripts/kbench.py --reps 5 --tag merge=$M 2>&1 | grep "merge=" | tee -a $O/kbench.txt
done
for M in 0 8 16; do
PRV_MERGE_MAX=$M python3 scripts/kbench.py --reps 5 --field 512 --tag f512_merge=$M 2>&1 | grep "merge=" | tee -a $O/kbench.txt
done
