#!/bin/bash
# Patch sampler study (round 5): parity tests, then per patch shape the planner loop's training configuration
# (scripts/trainprofile.py: 5 views 1280x720, 4096-ray cap, 2500 steps, 5 members side by side): ms per round of five
# member-steps, held-out PSNR / SSIM, and the backward launch's atomic requests (TCC_EA0_ATOMIC_sum) per composited sample.
#   usage: scripts/gpu/train_patch_study.sh <tag> [shapes...]     -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05patch}; mkdir -p $O; shift
SHAPES=${@:-1x1 2x2 4x2 4x4}   # each WxH or WxH:rays
timeout 900 python3 -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -5 | tee $O/pytest_train.txt
for tok in $SHAPES; do
  p=${tok%%:*}; R=4096; [[ $tok == *:* ]] && R=${tok##*:}; T=${p}_$R
  timeout 300 python3 scripts/trainprofile.py --patch $p --rays $R --members 5 --steps 2500 --chunk 500 --tail-single 50 --eval 2>&1 | grep -v amdgpu.ids | tee $O/loop_$T.txt | tail -3
  timeout 300 python3 scripts/trainprofile.py --patch $p --rays $R --members 1 --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tee $O/loop1_$T.txt | tail -2
  timeout 300 python3 scripts/trainbench.py --patch $p --rays $R --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tee $O/trainbench_$T.txt | tail -3
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_ATOMIC_sum --output-format csv -d $O/pmc_$T -o pmc -- python3 $GRAFT_REPO_ROOT/scripts/trainprofile.py --patch $p --rays $R --members 1 --steps 600 --chunk 100 --tail-single 100 > $O/pmc_$T.txt 2>&1)
  python3 - $O/pmc_$T $O/pmc_$T.txt <<'PY' | tee $O/atomics_$T.txt
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
  rm -rf $O/pmc_$T
done
