#!/usr/bin/env python3
"""Register / LDS / occupancy table of the HIP kernels (dev tool): compiles one .hip file with
-Rpass-analysis=kernel-resource-usage and prints one line per kernel.
  python scripts/kernel_resources.py [prv_kernels.hip|prv_train.hip] [name filter]"""
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_prv_amd import build as b  # noqa: E402

src = sys.argv[1] if len(sys.argv) > 1 else "prv_kernels.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
flags = [f for f in b.HIP_FLAGS if f not in ("-shared", "-fPIC")]
cmd = [b.hipcc()] + flags + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(b.CSRC, src), "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("void prv::", "")}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'Vspill':>6s} {'Sspill':>6s} {'LDS':>7s} {'waves':>5s}")
for r in rows:
    if flt and flt not in r["name"]:
        continue
    print(f"{r['name'][:58]:58s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} "
          f"{r.get('VGPRs Spill','?'):>6s} {r.get('SGPRs Spill','?'):>6s} {r.get('LDS Size [bytes/block]','?'):>7s} "
          f"{r.get('Occupancy [waves/SIMD]','?'):>5s}")
