cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02be
mkdir -p $O
python3 scripts/trainablate.py --save /tmp/state.prvf --rays 65536 2>&1 | tail -1
PRV_TRAIN_ABLATE=48 python3 nerf_prv_amd/build.py --force > $O/build.log 2>&1
STAMP_SUMS=1 python3 scripts/trainablate.py --load /tmp/state.prvf --rays 65536 --tag "sums" 2>&1 | tail -2 | tee -a $O/stamps.txt
PRV_TRAIN_ABLATE=49 python3 nerf_prv_amd/build.py --force > $O/build.log 2>&1
STAMP_SUMS=1 python3 scripts/trainablate.py --load /tmp/state.prvf --rays 65536 --tag "sums_noatomics" 2>&1 | tail -2 | tee -a $O/stamps.txt
PRV_TRAIN_ABLATE= python3 nerf_prv_amd/build.py --force > /dev/null 2>&1
