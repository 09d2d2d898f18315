#!/bin/bash
# Round 6: the trainer under both sampling rules of a training ray (prv_train_opts.step_mode) at upstream's batch: parity tests,
# held-out quality and ms/step on trainbench and on the planner loop's 5-view case, the backward launch's atomic requests per
# composited sample, per-kernel times.     usage: scripts/gpu/r06_train_rules.sh <tag>     -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06b}; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -15 | tee $O/pytest_train.txt
for rule in fixed ngp; do
  timeout 300 python3 scripts/trainbench.py --rule $rule --rays 65536 --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tee $O/trainbench_$rule.txt | tail -3
  timeout 300 python3 scripts/trainprofile.py --rule $rule --rays 65536 --members 5 --steps 2500 --chunk 500 --eval 2>&1 | grep -v amdgpu.ids | tee $O/loop_$rule.txt | tail -3
  timeout 300 python3 scripts/trainprofile.py --rule $rule --rays 65536 --members 1 --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tee $O/loop1_$rule.txt | tail -2
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_ATOMIC_sum --output-format csv -d $O/pmc_$rule -o pmc -- python3 $GRAFT_REPO_ROOT/scripts/trainprofile.py --rule $rule --rays 65536 --members 1 --steps 600 --chunk 100 --tail-single 100 > $O/pmc_$rule.txt 2>&1)
  python3 - $O/pmc_$rule $O/pmc_$rule.txt <<'PY' | tee $O/atomics_$rule.txt
import csv, glob, sys, re
rows = [r for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
by = {}
rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("::")[-1]
    by.setdefault(k, []).append(float(r["Counter_Value"]))
m = re.search(r"composited samples: ([\d ]+) mean", open(sys.argv[2]).read())
samples = [int(x) for x in m.group(1).split()] if m else []
for k, v in by.items():
    if "train_tile_kernel" in k and "false" in k and samples:
        tail = v[-len(samples):]
        print(f"{k}: {len(v)} launches; last {len(tail)}: TCC_EA0_ATOMIC per launch {sum(tail)/len(tail):.0f}, composited samples per step {sum(samples)/len(samples):.0f}, requests per composited sample {sum(tail)/sum(samples):.2f}")
PY
  rm -rf $O/pmc_$rule
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o tb -- python3 $GRAFT_REPO_ROOT/scripts/trainprofile.py --rule $rule --rays 65536 --members 1 --steps 1000 --chunk 500 > $O/trace_$rule.txt 2>&1)
  cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_loop1_$rule.csv; rm -rf $O/prof
  head -8 $O/kernel_stats_loop1_$rule.csv | cut -c1-150
done
