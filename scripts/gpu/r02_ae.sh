cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02ae
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sweep.py tests/test_gpu_configs.py tests/test_gpu_ingp.py -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -2 $O/pytest.log
for i in 1 2 3; do python3 scripts/kbench.py --reps 5 --tag pk 2>&1 | grep "^pk" | sed 's/eval_exact=[0-9]* //' | tee -a $O/kbench.txt; done
python3 scripts/kbench.py --reps 5 --field 512 --tag pk512 2>&1 | grep "^pk" | sed 's/eval_exact=[0-9]* //' | tee -a $O/kbench.txt
python3 scripts/kbench.py --reps 5 --views 540 --width 80 --height 45 --spp 16 --min-t 0.01 --tag pk_ref 2>&1 | grep "^pk" | sed 's/eval_exact=[0-9]* //' | tee -a $O/kbench.txt
