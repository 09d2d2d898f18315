# exits of prv_planner while another process keeps the GPU busy (round 2 saw ~1 crash in 1000 static teardowns):
#   usage: scripts/gpu/exit_stress.sh <runs of the ordinary return> <runs of the plain return> [<runs of the default exit>]
# normal = prv_runtime_shutdown (hipDeviceReset before main returns) then an ordinary return (round 3's default; the default
# is now the same shutdown followed by _exit, which cannot die in the runtime's exit handlers); noreset = plain return
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export PRV_SEGV_TRACE=1
O=gpurun_out/${EXIT_TAG:-r04exit}
mkdir -p $O
CFG=$(python3 scripts/gpu/repro_planner_inputs.py /tmp/repro | tail -1)
echo cfg $CFG
python3 scripts/kbench.py --reps 100000000 --tag bg > $O/bg.txt 2>&1 &
BG=$!
sleep 25
loop() { # mode, runs
  local fail=0 t0=$(date +%s)
  for i in $(seq 1 $2); do
    printf "21\nobjA\n-1\n" | PRV_PLANNER_EXIT=$1 nerf_prv_amd/prv_planner $CFG > $O/out.txt 2> $O/err.txt; rc=$?
    if [ $rc -ne 0 ]; then fail=$((fail+1)); echo "$1 run $i rc=$rc"; tail -30 $O/err.txt; cp $O/err.txt $O/err_$1_$i.txt; fi
    if [ $i -eq 1 ]; then grep chosen_nbvs $O/out.txt; fi
    find /tmp/repro -mindepth 1 -maxdepth 1 ! -name models ! -name cfg.yaml -exec rm -rf {} +
  done
  echo "$1: failures $fail / $2 in $(( $(date +%s) - t0 )) s" | tee -a $O/summary.txt
}
[ "${1:-0}" -gt 0 ] && loop normal $1
[ "${2:-0}" -gt 0 ] && loop noreset $2
[ "${3:-0}" -gt 0 ] && loop quick $3   # the default: ordered shutdown, flush, _exit
kill $BG 2>/dev/null; wait $BG 2>/dev/null
cat $O/summary.txt
