#!/bin/bash
# Round 6: where the backward tile launch goes at upstream's batch under the engine's marcher (262 K samples per step, frozen fields):
# phase time sums of block 0 (PRV_TRAIN_ABLATE=48) and the ablation builds (1 no scatter, 2 no dW, 8 no dX chain, 64 no backward tiles).
#   usage: scripts/gpu/r06_bwd_phases.sh <tag> [rule]    -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06c}; mkdir -p $O; RULE=${2:-ngp}
trap 'python3 -c "from nerf_prv_amd import build as b; b.build_hip(True)" > /dev/null 2>&1' EXIT
python3 scripts/trainprofile.py --rule $RULE --rays 65536 --members 1 --steps 2500 --chunk 500 --save-state /tmp/st 2>&1 | grep -v amdgpu.ids | tail -2 | tee $O/save.txt
for ab in 0 48 1 2 8 3 11 64; do
  if [ $ab != 0 ]; then PRV_TRAIN_ABLATE=$ab python3 -c "from nerf_prv_amd import build as b; b.build_hip(True)" > /dev/null 2>&1; fi
  echo "== PRV_TRAIN_ABLATE=$ab" | tee -a $O/phases.txt
  STAMP_SUMS=1 timeout 300 python3 scripts/trainprofile.py --rule $RULE --rays 65536 --members 1 --steps 300 --chunk 100 --load-state /tmp/st 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O/phases.txt
done
