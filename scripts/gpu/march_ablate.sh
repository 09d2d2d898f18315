#!/bin/bash
# where the march pass of the reference's own round goes (dev): refbench --step ngp on timing builds of the march kernel
#   PRV_ABLATE bits this loop builds: 64 = no dead-pixel writes | 128 = no mask-extension (chunk) writes | 256 = no record copy-out | 448 = all three
#   (other bits of the march kernels, not built here: 8 = no cheap rejection test | 16 = ray set-up, no occupancy walk | 32 = neither)
# An ablated library renders WRONG pixels: the default build is restored on every way out (also when interrupted).
trap 'python3 -c "from nerf_prv_amd import build as b; b.build_hip(True)" > /dev/null 2>&1' EXIT
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${1:-march_ablate}; mkdir -p $O
for ab in 64 128 256 448; do
  PRV_ABLATE=$ab python3 -c "from nerf_prv_amd import build as b; b.build_hip(True)" > /dev/null 2>&1
  echo "== PRV_ABLATE=$ab" | tee -a $O/out.txt
  python3 scripts/refbench.py --step ngp --reps 3 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
done
