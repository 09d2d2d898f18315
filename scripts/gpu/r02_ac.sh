cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02ac
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sweep.py tests/test_gpu_configs.py -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -2 $O/pytest.log
for i in 1 2; do python3 scripts/kbench.py --reps 5 --tag hashfirst 2>&1 | grep "hashfirst" | sed 's/eval_exact.*render=/render=/' | tee -a $O/kbench.txt; done
python3 scripts/kbench.py --reps 5 --field 512 --tag hashfirst512 2>&1 | grep "hashfirst" | sed 's/eval_exact.*render=/render=/' | tee -a $O/kbench.txt
python3 scripts/kbench.py --reps 5 --views 540 --width 80 --height 45 --spp 16 --min-t 0.01 --tag hashfirst_ref 2>&1 | grep "hashfirst" | sed 's/eval_exact.*render=/render=/' | tee -a $O/kbench.txt
PRV_HASH_FIRST=0 python3 nerf_prv_amd/build.py --force > $O/build.log 2>&1
for i in 1 2; do python3 scripts/kbench.py --reps 5 --tag levelorder 2>&1 | grep "levelorder" | sed 's/eval_exact.*render=/render=/' | tee -a $O/kbench.txt; done
python3 scripts/kbench.py --reps 5 --field 512 --tag levelorder512 2>&1 | grep "levelorder" | sed 's/eval_exact.*render=/render=/' | tee -a $O/kbench.txt
python3 scripts/kbench.py --reps 5 --views 540 --width 80 --height 45 --spp 16 --min-t 0.01 --tag levelorder_ref 2>&1 | grep "levelorder" | sed 's/eval_exact.*render=/render=/' | tee -a $O/kbench.txt
