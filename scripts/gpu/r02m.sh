cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02m
mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
PRV_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --no-training --steps 5 > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "rc=$?" >> $O/bench_dist1.err
timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-extras --no-training > $O/bench_2.json 2> $O/bench_2.err; echo "rc=$?" >> $O/bench_2.err
timeout 600 python bench.py --mode strong --views-total 256 --field 512 --steps 3 --no-cpu-baseline --no-training --no-extras > $O/bench_strong.json 2> $O/bench_strong.err; echo "rc=$?" >> $O/bench_strong.err
tail -3 $O/bench.err $O/bench_dist1.err $O/bench_2.err $O/bench_strong.err
python - <<'PY'
import json
for f in ("bench","bench_dist1","bench_strong"):
    try:
        d=json.load(open(f"gpurun_out/r02m/{f}.json"))
        r=d["roofline"]
        print(f, d["value"]/1e9, d["ms_per_step"], r["bound"], round(r["frac"],3), r["avg_launch_ms"], d.get("rccl_ranks"), d.get("cxx_rccl_round"))
    except Exception as e:
        print(f, "ERR", e)
PY
