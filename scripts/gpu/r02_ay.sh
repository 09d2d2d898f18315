cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export PRV_SEGV_TRACE=1
O=gpurun_out/r02ay
mkdir -p $O
CFG=$(python3 scripts/gpu/r02_ay.py /tmp/repro | tail -1)
echo cfg $CFG
python3 scripts/kbench.py --reps 100000 --tag bg > $O/bg.txt 2>&1 &
BG=$!
sleep 20
fail=0
for i in $(seq 1 1100); do
  printf "21\nobjA\n-1\n" | nerf_prv_amd/prv_planner $CFG > $O/out.txt 2> $O/err.txt; rc=$?
  if [ $rc -ne 0 ]; then fail=$((fail+1)); echo "run $i rc=$rc"; tail -40 $O/err.txt; cp $O/err.txt $O/err_$i.txt; fi
  if [ $i -eq 1 ]; then ls /tmp/repro; head -3 $O/out.txt | tail -1; fi
  find /tmp/repro -mindepth 1 -maxdepth 1 ! -name models ! -name cfg.yaml -exec rm -rf {} +
done
echo "failures: $fail / 1100"
kill $BG 2>/dev/null; wait $BG 2>/dev/null
