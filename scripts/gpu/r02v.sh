cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02v
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sweep.py -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -2 $O/pytest.log
for i in 1 2; do python3 scripts/kbench.py --reps 5 --tag every4 2>&1 | grep "every4" | tee -a $O/kbench.txt; done
python3 scripts/kbench.py --reps 1 --tag f256 > $O/kbench_256.txt 2>&1
bash scripts/pmc.sh $O/pmc256 1,2,3,4,5,6 > $O/pmc256.log 2>&1
grep f256 $O/kbench_256.txt
