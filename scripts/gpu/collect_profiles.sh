# The round's evidence in one GPU call: kernel-level lines, PMC passes (one rocprofv3 pass per counter set, scripts/pmc.sh),
# rocprofv3 --kernel-trace --stats of the bench command, the reference round under the profiler, and the bench line itself.
#   usage: scripts/gpu/collect_profiles.sh <tag>      -> gpurun_out/<tag>/...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r06prof}
O=gpurun_out/$TAG
mkdir -p $O
for cfg in "256 baseline" "256 dense" "512 baseline"; do
  set -- $cfg
  python3 scripts/kbench.py --reps 1 --field $1 --scene $2 --tag "f$1_$2" 2>&1 | grep -v amdgpu.ids > $O/kbench_$1_$2.txt
  bash scripts/pmc.sh $O/pmc_$1_$2 1,2,3,4,5,6 --field $1 --scene $2 > $O/pmc_$1_$2.log 2>&1
  cp $O/pmc_$1_$2/summary.txt $O/pmc_$1_$2_summary.txt
  cat $O/kbench_$1_$2.txt
done
python3 scripts/pmc_to_json.py "64<4, 5> baseline" $O/pmc_256_baseline/summary.txt $O/kbench_256_baseline.txt "round 6 final: 64 views 800x800 S=128, field 256^3, scene baseline" > /dev/null
python3 scripts/pmc_to_json.py "64<4, 5> dense" $O/pmc_256_dense/summary.txt $O/kbench_256_dense.txt "round 6 final: 64 views 800x800 S=128, field 256^3, scene dense" > /dev/null
python3 scripts/pmc_to_json.py "64<2, 10> baseline" $O/pmc_512_baseline/summary.txt $O/kbench_512_baseline.txt "round 6 final: 64 views 800x800 S=128, field 512^3, scene baseline" > /dev/null
cp profiles/r06_round_cost.json profiles/r06_pmc_traffic.json $O/
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-extras --no-training > $O/bench_prof.json 2> $O/bench_prof.err
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); head -5 "$f"; cp "$f" $O/bench_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof512 -- python3 bench.py --field 512 --no-cpu-baseline --no-extras --no-training > $O/bench_prof512.json 2> $O/bench_prof512.err
f=$(find $O/prof512 -name "*kernel_stats.csv" | head -1); head -4 "$f"; cp "$f" $O/bench_kernel_stats_field512.csv
rm -rf $O/prof $O/prof512 $O/pmc_*/p*/
bash scripts/gpu/profile_reference_round.sh $TAG/ref > $O/ref.log 2>&1; tail -5 $O/ref.log
