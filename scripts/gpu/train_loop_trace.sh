#!/bin/bash
# per-kernel time of the ensemble's training step in the planner loop's configuration (scripts/trainprofile.py:
# 5 members side by side, 5 views at 1280x720, 4096 rays, ~30 K samples per member-step): rocprofv3 --kernel-trace --stats
#   usage: scripts/gpu/train_loop_trace.sh <tag>   -> gpurun_out/<tag>/train_loop_kernel_stats.csv + a per-round summary
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04tl}; mkdir -p $O
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o tl -- python3 $GRAFT_REPO_ROOT/scripts/trainprofile.py --views 5 --steps 1000 --members ${MEMBERS:-5} > $O/trainprofile.txt 2>&1)
tail -3 $O/trainprofile.txt
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/train_loop_kernel_stats.csv
python3 - $O/train_loop_kernel_stats.csv <<'PY' | tee $O/train_loop_per_round.txt
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
import os
print(f"kernel time per round of {os.environ.get('MEMBERS', '5')} member-steps (1000 rounds), us; kernels overlap across the members' streams")
for r in rows[:16]:
    name = re.sub(r"\(.*", "", r["Name"]).replace("prv::", "").replace("(anonymous namespace)::", "")[:60]
    m = re.search(r"(train_\w+|adam_\w+|prepack_\w+|density_\w+)(<[^>]*>)?", r["Name"]); name = m.group(0)[:60] if m else name
    print(f"  {name:60s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:8.1f} us  per round {float(r['TotalDurationNs'])/1000/1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
print(f"  sum of kernel time per round {tot/1000/1e3:.1f} us")
PY
rm -rf $O/prof
