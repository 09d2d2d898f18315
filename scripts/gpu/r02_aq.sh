cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02aq
mkdir -p $O
for A in 0 1 3; do
PRV_TRAIN_ABLATE=$A python3 nerf_prv_amd/build.py --force > $O/build.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$A -- python3 scripts/trainbench.py --rays 65536 --steps 600 > $O/train_prof$A.txt 2>&1
f=$(find $O/prof$A -name "*kernel_stats.csv" | head -1); echo "== ablate $A"; head -6 "$f" | cut -d, -f1-4 | cut -c1-150; cp "$f" $O/train_kernel_stats_$A.csv
tail -1 $O/train_prof$A.txt
done
PRV_TRAIN_ABLATE= python3 nerf_prv_amd/build.py --force > /dev/null 2>&1
