#!/bin/bash
# how the members' kernels share the GPU: kernel trace of the 5-member training round -> scripts/train_concurrency.py
#   usage: scripts/gpu/train_concurrency_trace.sh <tag> [env assignments...]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05conc}; mkdir -p $O; shift
for v in "$@"; do export "$v"; done
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/prof -o tl -- python3 $GRAFT_REPO_ROOT/scripts/trainprofile.py --views 5 --steps 600 --members 5 > $O/trainprofile.txt 2>&1)
grep "500..  600" $O/trainprofile.txt
python3 scripts/train_concurrency.py $(find $O/prof -name "*kernel_trace.csv" | head -1) 100 5 | tee $O/concurrency.txt
rm -rf $O/prof
