# the PMC passes alone (scripts/pmc.sh, one rocprofv3 pass per counter set) for the three bench kernels and the reference
# round, and the two JSON files bench.py reads.   usage: scripts/gpu/collect_pmc.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r06pmc}
mkdir -p $O
for cfg in "256 baseline" "256 dense" "512 baseline"; do
  set -- $cfg
  python3 scripts/kbench.py --reps 1 --field $1 --scene $2 --tag "f$1_$2" 2>&1 | grep -v amdgpu.ids > $O/kbench_$1_$2.txt
  bash scripts/pmc.sh $O/pmc_$1_$2 1,2,3,4,5,6 --field $1 --scene $2 > $O/pmc_$1_$2.log 2>&1
  cp $O/pmc_$1_$2/summary.txt $O/pmc_$1_$2_summary.txt
  rm -rf $O/pmc_$1_$2/p*/
done
python3 scripts/pmc_to_json.py "64<4, 5> baseline" $O/pmc_256_baseline_summary.txt $O/kbench_256_baseline.txt "round 6 final: 64 views 800x800 S=128, field 256^3, scene baseline" > /dev/null
python3 scripts/pmc_to_json.py "64<4, 5> dense" $O/pmc_256_dense_summary.txt $O/kbench_256_dense.txt "round 6 final: 64 views 800x800 S=128, field 256^3, scene dense" > /dev/null
python3 scripts/pmc_to_json.py "64<2, 10> baseline" $O/pmc_512_baseline_summary.txt $O/kbench_512_baseline.txt "round 6 final: 64 views 800x800 S=128, field 512^3, scene baseline" > /dev/null
cp profiles/r06_round_cost.json profiles/r06_pmc_traffic.json $O/
PMC_SCRIPT=scripts/refbench.py bash scripts/pmc.sh $O/pmc_ref_ngp 1,2,3,4,5,6 --step ngp --reps 1 > $O/pmc_ref_ngp.log 2>&1
cp $O/pmc_ref_ngp/summary.txt $O/reference_round_ngp_pmc_summary.txt; rm -rf $O/pmc_ref_ngp/p*/
grep -A30 "render_queue" $O/reference_round_ngp_pmc_summary.txt | head -34
cat profiles/r04_round_cost.json | head -12
