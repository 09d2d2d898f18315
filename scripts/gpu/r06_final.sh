#!/bin/bash
# the round's last GPU call: the whole -m gpu suite, smoke, then the evidence of the final device code (scripts/gpu/collect_profiles.sh)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06final}; mkdir -p $O
timeout 3000 python3 -m pytest tests -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/pytest_gpu.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
bash scripts/gpu/collect_profiles.sh ${1:-r06final}/prof > $O/collect.log 2>&1; tail -3 $O/collect.log
python3 scripts/kbench.py --reps 3 --field hbm --views 16 --tag hbm 2>&1 | grep -v amdgpu.ids > $O/kbench_hbm.txt
bash scripts/pmc.sh $O/pmc_hbm 4,5,6 --field hbm --views 16 > $O/pmc_hbm.log 2>&1; cp $O/pmc_hbm/summary.txt $O/pmc_hbm_summary.txt; rm -rf $O/pmc_hbm/p*/
python3 scripts/pmc_to_json.py "hbm baseline" $O/pmc_hbm_summary.txt $O/kbench_hbm.txt "round 6 final: 16 views 800x800 S=128, field_hbm (L=16 F=2 log2T=24 finest 2048), scene baseline" > /dev/null
cp profiles/r06_pmc_traffic.json profiles/r06_round_cost.json $O/
timeout 900 python3 bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc $?"
