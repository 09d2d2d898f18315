#!/usr/bin/env python3
"""steady-state per-kernel time of the training step from a rocprofv3 kernel trace of scripts/trainbench.py --members N
(dev tool): the last 100 steps.   python scripts/train_trace_summary.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    m = re.search(r"(\w+_kernel(_m)?)(<[^>]*>)?", n)
    return m.group(1) + (m.group(3) or "") if m else n[:40]


for tag, pat in (("training steps", "adam_mlp_kernel"),):  # a step's last kernel
    idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
    if len(idx) < 101:
        continue
    lo, hi = idx[-101], idx[-1]
    d = collections.defaultdict(list)
    for r in rows[lo + 1:hi + 1]:
        d[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(tag)
    tot = 0
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {k:44s} n={len(v):4d} avg {sum(v)/len(v)/1e3:8.1f} us   per step {sum(v)/100/1e3:8.1f}")
        tot += sum(v)
    print(f"  kernel time per step {tot/100/1e3:.1f} us, wall per step {(int(rows[hi]['End_Timestamp']) - int(rows[lo]['End_Timestamp']))/100/1e3:.1f} us")
