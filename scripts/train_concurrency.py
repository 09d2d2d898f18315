#!/usr/bin/env python3
"""How the members' training steps share the GPU (dev tool): from a rocprofv3 kernel trace of scripts/trainprofile.py
--members N, over the last ROUNDS rounds -- wall time per round, the time at least one kernel runs, the mean number of
kernels running, and per stream (= member) the time between the end of a kernel and the start of the next one.
   python scripts/train_concurrency.py <kernel_trace.csv> [rounds] [members]"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 200
key = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prepack = [i for i, r in enumerate(rows) if "adam_mlp_kernel" in r["Kernel_Name"]]  # a step's last kernel
streams = sorted({rows[i][key] for i in prepack[-50:]})  # hardware queues the members' streams sit on (5 members: 5 with a queue each, 3 when pooled)
n_members = int(sys.argv[3]) if len(sys.argv) > 3 else len(streams)  # members stepping side by side (one adam_mlp launch per member-step)
lo = prepack[-(rounds * n_members + 1)]
sel = rows[lo + 1:]
t0, t1 = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)
ev = []
for r in sel:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
busy = 0
area = 0
depth = 0
last = t0
for t, d in ev:
    if depth > 0:
        busy += t - last
    area += depth * (t - last)
    depth += d
    last = t
print(f"{n_members} members on {len(streams)} hardware queues ({key}), last {rounds} rounds: wall {(t1 - t0) / rounds / 1e3:.1f} us per round, at least one kernel running "
      f"{busy / rounds / 1e3:.1f} us ({100.0 * busy / (t1 - t0):.0f} %), kernels running on average {area / max(busy, 1):.2f} while any runs, "
      f"sum of kernel durations {area / rounds / 1e3:.1f} us per round")
# per kernel: the share of the wall time at least one launch of it runs, and how many of them run on average while one does
by = collections.defaultdict(list)
for r in sel:
    n = r["Kernel_Name"]
    n = n[n.find("train_") if "train_" in n else (n.find("adam_") if "adam_" in n else 0):].split("(")[0][:40]
    by[n].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for n, iv in sorted(by.items(), key=lambda kv: -sum(b - a for a, b in kv[1]))[:8]:
    ev2 = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
    cover = area2 = depth2 = 0
    last2 = ev2[0][0]
    for t, d in ev2:
        if depth2 > 0:
            cover += t - last2
        area2 += depth2 * (t - last2)
        depth2 += d
        last2 = t
    print(f"  {n:40s} runs {100.0 * cover / (t1 - t0):5.1f} % of the wall time, {area2 / max(cover, 1):.2f} launches at once while it does, "
          f"{area2 / len(iv) / 1e3:.1f} us per launch")
gaps = collections.defaultdict(list)
prev = {}
for r in sel:
    q = r[key]
    if q in prev:
        gaps[q].append(int(r["Start_Timestamp"]) - prev[q])
    prev[q] = int(r["End_Timestamp"])
for q in streams:
    g = gaps.get(q, [])
    if g:
        g2 = sorted(g)
        print(f"  stream {q}: {len(g)} kernel-to-kernel gaps, median {g2[len(g2) // 2] / 1e3:.2f} us, mean {sum(g) / len(g) / 1e3:.2f} us, "
              f"sum per round {sum(g) / rounds / 1e3:.1f} us")
