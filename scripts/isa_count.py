#!/usr/bin/env python3
"""Static instruction counts of a render kernel instance's hot loop (dev tool): compiles prv_kernels.hip to assembly and
prints, per basic block with more than a few VALU instructions, the VALU / MFMA / VMEM / LDS / SALU counts, plus the opcode
histogram of the two hot blocks (gather, MLP + compositing).
  python scripts/isa_count.py [F] [NDENSE] [ngp: 0|1]"""
import collections
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_prv_amd import build as b  # noqa: E402

F, ND, NGP = (sys.argv + ["4", "5", "0"])[1:4]
asm = "/tmp/prv_kernels.s"
flags = [f for f in b.HIP_FLAGS if f not in ("-shared", "-fPIC")]
subprocess.run([b.hipcc()] + flags + ["-S", "--cuda-device-only", "-o", asm, os.path.join(b.CSRC, "prv_kernels.hip")], check=True,
               stderr=subprocess.DEVNULL)
name = f"_ZN3prv21render_queue64_kernelILi{F}ELi{ND}ELb{NGP}EEEvNS_12RenderParamsE"
text = open(asm).read().split("\n")
start = next(i for i, l in enumerate(text) if l.startswith(name + ":"))
end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
lines = text[start:end]
NOT_VALU = ("salu", "vmem_ld", "vmem_st", "lds", "other", "mfma")


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global_load", "buffer_load")): return "vmem_ld"
    if op.startswith(("global_store", "global_atomic")): return "vmem_st"
    if op.startswith("ds_"): return "lds"
    return "other"


def ops(a, z):
    for l in lines[a:z]:
        l = l.strip()
        if l and not l.startswith((";", ".")) and not l.endswith(":"):
            yield l.split()[0]


labels = [i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)] + [len(lines)]
print(f"render_queue64_kernel<{F}, {ND}, {NGP}>: blocks with > 12 VALU instructions")
hot = []
for a, z in zip(labels, labels[1:]):
    c = collections.Counter(cls(o) for o in ops(a, z))
    if c["valu"] > 12 or c["mfma"]:
        print(f"  {lines[a].split(':')[0]:12s} VALU {c['valu']:4d}  MFMA {c['mfma']:3d}  VMEM {c['vmem_ld']:3d}  LDS {c['lds']:3d}  SALU {c['salu']:3d}")
        if c["valu"] > 200:
            hot.append((a, z))
for a, z in hot:
    h = collections.Counter(o for o in ops(a, z) if cls(o) in ("valu", "mfma"))
    print(f"  -- {lines[a].split(':')[0]}: " + ", ".join(f"{k} {v}" for k, v in h.most_common(30)))
