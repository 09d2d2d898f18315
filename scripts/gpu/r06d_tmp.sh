cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r06d; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -15 | tee $O/pytest_train.txt
for rule in fixed ngp; do
  timeout 300 python3 scripts/trainbench.py --rule $rule --eval-rule ngp --rays 65536 --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tee $O/trainbench_$rule.txt | tail -3
  timeout 300 python3 scripts/trainprofile.py --rule $rule --rays 65536 --members 5 --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tee $O/loop_$rule.txt | tail -2
  timeout 300 python3 scripts/trainprofile.py --rule $rule --rays 65536 --members 1 --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tee $O/loop1_$rule.txt | tail -2
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o tb -- python3 $GRAFT_REPO_ROOT/scripts/trainprofile.py --rule ngp --rays 65536 --members 1 --steps 1000 --chunk 500 > $O/trace_ngp.txt 2>&1)
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_loop1_ngp.csv; rm -rf $O/prof
head -8 $O/kernel_stats_loop1_ngp.csv | cut -c1-150
