#!/usr/bin/env python3
"""Static instruction counts of a render kernel instance's hot loop (dev tool): compiles prv_kernels.hip to assembly and
prints, per basic block with more than a few VALU instructions, the VALU / MFMA / VMEM / LDS / SALU counts, plus the opcode
histogram of the two hot blocks (gather, MLP + compositing).
  python scripts/isa_count.py [F] [NDENSE] [ngp: 0|1] [--json]
--json: also record the issue-class histogram of the two hot blocks (c2 / c4 / c8 / mfma: the classes
scripts/valu_rate.hip measured, profiles/r04_valu_issue_rate.txt) in profiles/r06_isa_classes.json, keyed by the kernel
instance and stamped with the digest of the device code -- what bench.py prices the ISSUED instructions with."""
import collections
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_prv_amd import build as b  # noqa: E402

WANT_JSON = "--json" in sys.argv
F, ND, NGP = ([a for a in sys.argv if a not in ("--json", "--cache")] + ["4", "5", "0"])[1:4]
asm = "/tmp/prv_kernels.s"
flags = [f for f in b.HIP_FLAGS if f not in ("-shared", "-fPIC")]
subprocess.run([b.hipcc()] + flags + ["-S", "--cuda-device-only", "-o", asm, os.path.join(b.CSRC, "prv_kernels.hip")], check=True,
               stderr=subprocess.DEVNULL)
CACHE = "1" if "--cache" in sys.argv else "0"  # the CornerCache instance (small images under the engine's rule)
name = f"_ZN3prv21render_queue64_kernelILi{F}ELi{ND}ELb{NGP}ELb{CACHE}EEEvNS_12RenderParamsE"
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
print(f"render_queue64_kernel<{F}, {ND}, {NGP}, {CACHE}>: blocks with > 12 VALU instructions")
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


# issue classes as measured by scripts/valu_rate.hip (cycles per wave64 instruction per SIMD with >= 2 waves resident)
C2 = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32",
      "v_xor_b32", "v_mov_b32", "v_bitop3_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_max_f16", "v_accvgpr_write_b32", "v_accvgpr_read_b32",
      "v_fmac_f32", "v_mac_f32", "v_not_b32"}
C8 = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_permlane32_swap_b32", "v_fma_f16",
      "v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_rcp_iflag_f32"}


def issue_class(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.startswith("v_mfma"): return "mfma"
    if base in C2 and not op.endswith(("_dpp", "_sdwa")): return "c2"
    if base in C8: return "c8"
    return "c4"


if WANT_JSON and len(hot) >= 2:
    import json

    from nerf_prv_amd import _lib

    hist = collections.Counter()
    for a, z in hot[:2]:  # the gather block and the MLP + compositing block: once per 64-slot iteration
        hist.update(issue_class(o) for o in ops(a, z) if cls(o) in ("valu", "mfma"))
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_isa_classes.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[f"64<{F}, {ND}>" + (" ngp" if NGP == "1" else "") + (" cache" if CACHE == "1" else "")] = dict(hist, device_code_sha256=_lib.device_code_digest(),
                                                                    source="scripts/isa_count.py: static histogram of the gather block and the MLP + compositing block")
    json.dump(data, open(path, "w"), indent=1)
    print("  issue classes of the hot loop:", dict(hist))
