cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bs
mkdir -p $O
python3 scripts/trainablate.py --save /tmp/state.prvf --rays 65536 2>&1 | tail -1
for A in 48 112 176 304 496; do
PRV_TRAIN_ABLATE=$A python3 nerf_prv_amd/build.py --force > $O/build.log 2>&1
STAMP_SUMS=1 python3 scripts/trainablate.py --load /tmp/state.prvf --rays 65536 --tag "ablate$A" 2>&1 | tail -2 | tr '\n' ' ' | sed 's/(fwd + composite + bwd + grad clear), //' | tee -a $O/stamps.txt; echo
done
PRV_TRAIN_ABLATE= python3 nerf_prv_amd/build.py --force > /dev/null 2>&1
