#!/bin/bash
# Patch sampler study (round 5): parity tests, then per patch shape the planner loop's training configuration
# (scripts/trainprofile.py: 5 views 1280x720, 4096-ray cap, 2500 steps, 5 members side by side): ms per round of five
# member-steps, held-out PSNR / SSIM, and the backward launch's atomic requests (TCC_EA0_ATOMIC_sum) per composited sample.
#   usage: scripts/gpu/train_patch_study.sh <tag> [shapes...]     -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05patch}; mkdir -p $O; shift
SHAPES=${@:-1x1 2x2 4x2 4x4}
timeout 900 python3 -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -5 | tee $O/pytest_train.txt
for p in $SHAPES; do
  timeout 300 python3 scripts/trainprofile.py --patch $p --members 5 --steps 2500 --chunk 500 --eval 2>&1 | grep -v amdgpu.ids | tee $O/loop_$p.txt | tail -3
  timeout 300 python3 scripts/trainprofile.py --patch $p --members 1 --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tee $O/loop1_$p.txt | tail -2
  timeout 300 python3 scripts/trainbench.py --patch $p --rays 4096 --steps 2500 --chunk 500 2>&1 | grep -v amdgpu.ids | tee $O/trainbench_$p.txt | tail -3
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_ATOMIC_sum --output-format csv -d $O/pmc_$p -o pmc -- python3 $GRAFT_REPO_ROOT/scripts/trainprofile.py --patch $p --members 1 --steps 600 --chunk 100 > $O/pmc_$p.txt 2>&1)
  python3 - $O/pmc_$p $O/pmc_$p.txt <<'PY' | tee $O/atomics_$p.txt
import csv, glob, sys, re
rows = [r for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
by = {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0]
    by.setdefault(k, []).append(float(r["Counter_Value"]))
samples = [int(m.group(1)) for m in re.finditer(r"member 0:\s+(\d+) samples", open(sys.argv[2]).read())]
for k, v in by.items():
    if "train_tile_kernel" in k and "false" in k:
        tail = v[-300:]
        s = sum(samples[-3:]) / max(len(samples[-3:]), 1)
        print(f"{k}: {len(v)} launches, TCC_EA0_ATOMIC per launch (last 300) {sum(tail)/len(tail):.0f}; composited samples per step ~{s:.0f}; requests per composited sample {sum(tail)/len(tail)/max(s,1):.2f}")
PY
  rm -rf $O/pmc_$p
done
