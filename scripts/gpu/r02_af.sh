cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02af
mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -o "\b\(TA\|TCP\|TD\)_[A-Za-z0-9_]*" | sort -u > $O/avail_ta_tcp.txt
wc -l $O/avail_ta_tcp.txt
bash scripts/pmc.sh $O/pmc 7,9,10,11 > $O/pmc.log 2>&1
grep -A40 "render_queue64" $O/pmc/summary.txt | head -60
