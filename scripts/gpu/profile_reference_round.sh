# The reference's own scoring round (540 views x 80x45 x 16 spp x 5 members, scripts/refbench.py) under the profiler:
# rocprofv3 --kernel-trace --stats of both stepping rules, and the PMC sets (one pass per set, scripts/pmc.sh) of the engine's rule.
#   usage: scripts/gpu/profile_reference_round.sh <tag>      -> gpurun_out/<tag>/...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-r04ref}
mkdir -p $O
python3 scripts/refbench.py --step both --reps 3 2>&1 | grep -v amdgpu.ids > $O/refbench.txt
cat $O/refbench.txt
for mode in ngp fixed; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$mode -- python3 scripts/refbench.py --step $mode --reps 3 > $O/refbench_prof_$mode.txt 2> $O/refbench_prof_$mode.err
  f=$(find $O/prof_$mode -name "*kernel_stats.csv" | head -1); head -6 "$f"; cp "$f" $O/reference_round_${mode}_kernel_stats.csv
  rm -rf $O/prof_$mode
done
PMC_SCRIPT=scripts/refbench.py bash scripts/pmc.sh $O/pmc_ngp 1,2,3,4,5,6 --step ngp --reps 1 > $O/pmc_ngp.log 2>&1
cp $O/pmc_ngp/summary.txt $O/reference_round_ngp_pmc_summary.txt
rm -rf $O/pmc_ngp/p*/
tail -40 $O/reference_round_ngp_pmc_summary.txt
