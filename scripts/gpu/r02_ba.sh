cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02ba
mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do
timeout 600 python3 bench.py --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err; rc=$?
python3 -c "
import json;d=json.load(open('$O/bench_$i.json'));print('run $i rc=$rc', round(d['value']/1e9,2),round(d['ms_per_step'],3),round(d['roofline']['shader_clock_ghz_measured'],3), round(d['training']['steps_per_s']))"
done
PRV_FORCE_DIST=1 timeout 600 python3 bench.py --no-cpu-baseline --no-extras --no-training > $O/bench_dist.json 2> $O/bench_dist.err; echo "dist rc=$?"
python3 -c "
import json;d=json.load(open('$O/bench_dist.json'));print(d.get('rccl_ranks'), d.get('cxx_rccl_round'))"
