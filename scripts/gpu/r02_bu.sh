cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02bu
mkdir -p $O
bash scripts/pmc.sh $O/pmc 8 > $O/pmc.log 2>&1
grep -A10 "render_queue64\|march" $O/pmc/summary.txt | head -30
