#!/bin/bash
# per-kernel time of the ensemble training step (rocprofv3 --kernel-trace --stats on scripts/trainbench.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/tk; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o tb -- python3 $GRAFT_REPO_ROOT/scripts/trainbench.py --rays 4096 --steps 400 --chunk 200 --members 5 > $GRAFT_REPO_ROOT/$O/out.txt 2>&1
cd $GRAFT_REPO_ROOT
tail -3 $O/out.txt
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
head -25 $O/kernel_stats.csv | cut -c1-200
find $O/prof -name "*.db" -delete 2>/dev/null; find $O/prof -name "*kernel_trace.csv" -size +20M -delete
