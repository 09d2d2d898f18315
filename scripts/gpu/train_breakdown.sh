#!/bin/bash
# Where a training step goes at the loop's batch size (4096 rays): per-kernel trace of the real build, then the
# backward tile kernel's phase sums (stamps) and its ablations (no scatter / no dW) on timing builds.
#   usage: scripts/gpu/train_breakdown.sh <tag>     -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r04train}; mkdir -p $O
python3 scripts/trainablate.py --save /tmp/state.prvf --rays 4096 > $O/save.txt 2>&1; tail -1 $O/save.txt
python3 scripts/trainablate.py --load /tmp/state.prvf --rays 4096 --tag real 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o tb -- python3 $GRAFT_REPO_ROOT/scripts/trainbench.py --rays 4096 --steps 400 --chunk 200 --members 5 > $O/trainbench.txt 2>&1)
tail -4 $O/trainbench.txt
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python3 scripts/train_trace_summary.py $f | tee $O/trace_summary.txt
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/prof
for ab in 48 1 2 4; do
  PRV_TRAIN_ABLATE=$ab python3 -c "from nerf_prv_amd import build as b; b.build_hip(True)" > /dev/null 2>&1
  STAMP_SUMS=1 python3 scripts/trainablate.py --load /tmp/state.prvf --rays 4096 --tag "ablate=$ab" 2>&1 | grep -v amdgpu.ids | tee -a $O/ablate.txt
done
