cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r06j; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_wholeview.py tests/test_gpu_parity.py tests/test_gpu_train.py -x -q 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/pytest.txt
timeout 300 python3 scripts/kbench.py --field hbm --views 16 --reps 3 --tag hbm 2>&1 | grep -v amdgpu.ids | tee $O/kbench_hbm.txt
bash scripts/pmc.sh $O/pmc_hbm 4,5,6 --field hbm --views 16 > $O/pmc_hbm.log 2>&1; cp $O/pmc_hbm/summary.txt $O/pmc_hbm_summary.txt; rm -rf $O/pmc_hbm/p*/; cat $O/pmc_hbm_summary.txt
