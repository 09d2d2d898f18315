"""In-tree build of the native pieces (explicit hipcc / g++ command lines, no JIT cache).

  libprv_hip.so   HIP kernels + C ABI                      (hipcc --offload-arch=gfx950)
  libprv_host.so  C++ planner shell, no GPU dependency     (g++)
  prv_planner     the planner executable                   (g++, dlopen-free: links both)
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
ROOT = os.path.dirname(HERE)

# -ffp-contract=off: op order is part of the arithmetic contract with the oracle.
# -fno-slp-vectorize: keeps the trilinear blend on v_fma_mix_f32 instead of cvt + v_pk_fma_f32.
# -amdgpu-mfma-vgpr-form: MFMA results land in arch VGPRs (no v_accvgpr_read copies before the VALU epilogues).
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-slp-vectorize",
             "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-Wall"]
if os.environ.get("PRV_ABLATE"):
    HIP_FLAGS.append("-DPRV_ABLATE=" + os.environ["PRV_ABLATE"])
if os.environ.get("PRV_TRAIN_ABLATE"):  # dev only: compile-time ablation of the training tile kernel
    HIP_FLAGS.append("-DPRV_TRAIN_ABLATE=" + os.environ["PRV_TRAIN_ABLATE"])
if os.environ.get("PRV_EXTRA_HIPFLAGS"):  # dev only: compiler-option experiments (scripts/gpu/ab_flags.sh)
    HIP_FLAGS += os.environ["PRV_EXTRA_HIPFLAGS"].split()
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-Wall", "-ffp-contract=off"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (and there is no fallback)")


def build_hip(force=False):
    out = os.path.join(HERE, "libprv_hip.so")
    srcs = [os.path.join(CSRC, f) for f in ("prv_kernels.hip", "prv_train.hip", "prv_api.cpp")]
    deps = srcs + [os.path.join(CSRC, f) for f in ("prv_device.hpp", "prv_kernels.hpp", "prv_json.hpp", "prv_train.hpp",
                                                   "prv_train_api.inc", "prv_comm_api.inc", "prv_star.hpp", "prv_levels.hpp", "prv_ingp.hpp")] + [
        os.path.join(ROOT, "include", "prv.h")]
    # the flags the library was built with (PRV_ABLATE / PRV_TRAIN_ABLATE / PRV_EXTRA_HIPFLAGS timing builds change them, not the
    # sources: a comparison of "two builds" that skipped the second compile measures one binary twice)
    stamp = out + ".flags"
    flags = " ".join(HIP_FLAGS)
    same_flags = os.path.exists(stamp) and open(stamp).read() == flags
    if force or not same_flags or _newer(out, deps):
        _run([hipcc()] + HIP_FLAGS + ["-o", out] + srcs + ["-ldl", "-lz"])  # zlib: .ingp snapshots; librccl itself is dlopen'ed (prv_comm_api.inc)
        with open(stamp, "w") as fh:
            fh.write(flags)
    return out


def build_host(force=False):
    if not os.path.isdir(HOST):
        return None
    out = os.path.join(HERE, "libprv_host.so")
    srcs = sorted(os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".cpp") and f != "main.cpp")
    deps = srcs + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp")] + [
        os.path.join(CSRC, "prv_json.hpp"), os.path.join(CSRC, "prv_star.hpp"), os.path.join(CSRC, "prv_ingp.hpp"),
        os.path.join(CSRC, "prv_levels.hpp"), os.path.join(ROOT, "include", "prv.h"),
        os.path.join(ROOT, "include", "prv_host.h")]
    deps = [d for d in deps if os.path.exists(d)]
    if srcs and (force or _newer(out, deps)):
        _run(["g++"] + HOST_FLAGS + ["-shared", "-o", out] + srcs + ["-lz"])
    exe = os.path.join(HERE, "prv_planner")
    main = os.path.join(HOST, "main.cpp")
    if os.path.exists(main) and (force or _newer(exe, deps + [main, os.path.join(HERE, "libprv_hip.so")])):
        _run(["g++"] + HOST_FLAGS + ["-o", exe, main] + srcs +
             ["-L" + HERE, "-lprv_hip", "-lz", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + HERE, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    return out


def build_all(force=False):
    build_hip(force)
    build_host(force)


if __name__ == "__main__":
    build_all("--force" in sys.argv)
