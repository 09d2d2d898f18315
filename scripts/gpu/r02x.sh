cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02x
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sweep.py -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -2 $O/pytest.log
for i in 1 2; do python3 scripts/kbench.py --reps 5 --tag march2 2>&1 | grep "march2" | tee -a $O/kbench.txt; done
python3 scripts/kbench.py --reps 5 --views 540 --width 80 --height 45 --spp 16 --min-t 0.01 --tag ref_march2 2>&1 | grep "march2" | tee -a $O/kbench.txt
