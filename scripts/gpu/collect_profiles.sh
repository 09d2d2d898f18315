# The round's evidence in one GPU call: kernel-level lines, PMC passes (one rocprofv3 pass per counter set, scripts/pmc.sh),
# rocprofv3 --kernel-trace --stats of the bench command, and the bench line itself.
#   usage: scripts/gpu/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r03prof}
mkdir -p $O
for cfg in "256 baseline" "256 dense" "512 baseline"; do
  set -- $cfg
  python3 scripts/kbench.py --reps 1 --field $1 --scene $2 --tag "f$1_$2" 2>&1 | grep -v amdgpu.ids > $O/kbench_$1_$2.txt
  bash scripts/pmc.sh $O/pmc_$1_$2 1,2,3,4,5,6 --field $1 --scene $2 > $O/pmc_$1_$2.log 2>&1
  cat $O/kbench_$1_$2.txt
done
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-extras --no-training > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -5 "$f"; cp "$f" $O/bench_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof512 -- python3 bench.py --field 512 --no-cpu-baseline --no-extras --no-training > $O/bench_prof512.json 2> $O/bench_prof512.err
f=$(find $O/prof512 -name "*kernel_stats.csv" | head -1); head -4 "$f"; cp "$f" $O/bench_kernel_stats_field512.csv
rm -rf $O/prof $O/prof512 $O/pmc_*/p*/
