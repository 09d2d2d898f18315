cd $GRAFT_REPO_ROOT
O=gpurun_out/r02u
mkdir -p $O
for i in 1 2 3; do
timeout 600 python3 bench.py --no-cpu-baseline --no-training > $O/bench_$i.json 2> $O/bench_$i.err
done
python3 - <<'PY'
import json
for i in (1,2,3):
    d=json.load(open(f"gpurun_out/r02u/bench_{i}.json"))
    x=d["field512"]; y=d["scene_baseline"]
    print(i, round(d["value"]/1e9,2), round(d["ms_per_step"],3), round(d["roofline"]["avg_launch_ms"],3), "| 512:", round(x["value"]/1e9,2), round(x["ms_per_step"],2), round(x["roofline"]["avg_launch_ms"],2), "| base:", round(y["value"]/1e9,2), round(y["ms_per_step"],2), round(y["roofline"]["avg_launch_ms"],2))
PY
