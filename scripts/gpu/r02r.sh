cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02r
mkdir -p $O
PRV_POOL=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_sweep.py -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for PM in "0 16" "1 4" "1 8" "1 12" "1 16" "1 24"; do set -- $PM
PRV_POOL=$1 PRV_MERGE_MAX=$2 timeout 120 python3 scripts/kbench.py --reps 5 --tag pool=$1_m=$2 2>&1 | grep "pool=" | tee -a $O/kbench.txt
done
PRV_POOL=1 PRV_MERGE_MAX=16 timeout 120 python3 scripts/kbench.py --reps 5 --field 512 --tag f512_pool=1_m=16 2>&1 | grep "pool=" | tee -a $O/kbench.txt
PRV_POOL=1 PRV_MERGE_MAX=16 timeout 120 python3 scripts/kbench.py --reps 5 --views 540 --width 80 --height 45 --spp 16 --min-t 0.01 --tag ref_pool=1 2>&1 | grep "pool=" | tee -a $O/kbench.txt
