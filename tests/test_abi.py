"""The C-ABI libraries load without a GPU and export every symbol their headers declare."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(prvh?_[a-z0-9_]+)\s*\(", text)) - {"prvh_score_fn"})


def exported(lib):
    out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "nerf_prv_amd", lib)], text=True)
    return {l.split()[-1] for l in out.splitlines() if " T " in l}


def test_hip_library_exports_every_declared_symbol():
    names = declared("prv.h")
    assert len(names) >= 30
    missing = [n for n in names if n not in exported("libprv_hip.so")]
    assert not missing, missing


def test_host_library_exports_every_declared_symbol():
    names = declared("prv_host.h")
    missing = [n for n in names if n not in exported("libprv_host.so")]
    assert not missing, missing


def test_python_binding_lists_every_declared_symbol():
    from nerf_prv_amd import _lib, planner

    assert sorted(_lib.SIGNATURES) == declared("prv.h")
    assert sorted(planner.HOST_SIGNATURES) == declared("prv_host.h")
    lib = _lib.load()  # loads on a CPU-only box: no compute entry point is called here
    assert lib.prv_abi_version() == 5
    planner.host()


def test_no_device_fails_loudly_not_silently():
    """without a GPU the product refuses to run: there is no CPU fallback to fall into"""
    import ctypes as C

    from nerf_prv_amd import _lib

    lib = _lib.load()
    if lib.prv_device_count() > 0:
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    assert lib.prv_create(C.byref(h), 0) == _lib.PRV_E_NODEVICE
    assert b"no CPU path" in lib.prv_last_error(None)
    assert lib.prv_runtime_shutdown() == _lib.PRV_OK  # no context was ever created: nothing to shut down, nothing touched


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under nerf_prv_amd/, include/ may reference it"""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "nerf_prv_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"\boracle\b|prv_oracle|orc_", txt) and f != "build.py":
                    lines = [l for l in txt.splitlines() if re.search(r"import oracle|from oracle|prv_oracle\.h|orc_[a-z]+\(", l)]
                    if lines:
                        bad.append((f, lines[:2]))
    assert not bad, bad


def test_headers_are_plain_c(tmp_path):
    """the boundary is a C ABI: both headers must compile as C99 and a C caller must link"""
    src = tmp_path / "c_caller.c"
    src.write_text(
        '#include "prv.h"\n#include "prv_host.h"\n#include <stdio.h>\n'
        "int main(void) {\n"
        "  prv_ctx* ctx = NULL; prv_field_desc d = {8, 4, 14, 8, 96, 32, 3.0f, 4.0f};\n"
        "  uint64_t t = 0, m = 0, o = 0; double pose[16], tm[16]; double pos[3] = {0.1, 0.2, 0.25}, c[3] = {1e-10, 1e-10, 1e-10};\n"
        "  if (prv_abi_version() != PRV_ABI_VERSION) return 1;\n"
        "  if (prv_model_sizes(&d, &t, &m, &o) != PRV_OK || m != PRV_MLP_HALFS) return 2;\n"
        "  prvh_view_pose(pos, c, pose); prvh_transform_matrix(pose, tm);\n"
        "  if (prv_create(&ctx, 0) == PRV_OK) prv_destroy(ctx); else printf(\"%s\\n\", prv_last_error(NULL));\n"
        '  printf("table halfs %llu tm03 %.6f\\n", (unsigned long long)t, tm[3]);\n'
        '  printf("sizes %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(prv_field_desc), sizeof(prv_render_opts), sizeof(prv_score_record), sizeof(prv_stats),\n'
        "         sizeof(prv_intrinsics), sizeof(prv_rs2_intrinsics), sizeof(prv_train_opts));\n"
        "  if (prv_runtime_shutdown() != PRV_OK) return 3;\n  return 0;\n}\n")
    inc, libdir = os.path.join(ROOT, "include"), os.path.join(ROOT, "nerf_prv_amd")
    exe = tmp_path / "c_caller"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", inc, str(src), "-o", str(exe), "-L", libdir,
                           "-lprv_hip", "-lprv_host", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "table halfs" in out.stdout and "tm03 0.25" in out.stdout  # json translation = (z, x, y) of the position
    # the ctypes mirrors of the ABI's structs are the C compiler's size (a field added on one side only shows up here)
    import ctypes as C

    from nerf_prv_amd import _lib, api

    c_sizes = [int(x) for x in [l for l in out.stdout.splitlines() if l.startswith("sizes ")][0].split()[1:]]
    py_sizes = [C.sizeof(t) for t in (_lib.FieldDesc, _lib.RenderOpts, _lib.ScoreRecord, _lib.Stats, _lib.Intrinsics, _lib.Rs2Intrinsics,
                                      _lib.TrainOpts)]
    assert c_sizes == py_sizes, (c_sizes, py_sizes)
    assert api.render_opts(8, 8).step_mode == _lib.STEP_FIXED_S and api.engine_render_opts(8, 8, 0, 1, 0.01).step_mode == _lib.STEP_NGP


def test_device_code_keeps_the_rounding_contract(tmp_path):
    """the code objects inside libprv_hip.so: gfx950 only; the MLPs really are on the matrix cores; and no
    `v_fma_mix{lo,hi}_f16` -- that instruction is how clang folds `(half)(a * b)` into ONE rounding, where the
    arithmetic contract with the oracle is round-to-float-then-to-half (prv_device.hpp: to_half)"""
    import shutil

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not installed")
    so = os.path.join(ROOT, "nerf_prv_amd", "libprv_hip.so")
    assert os.path.exists(so), "libprv_hip.so missing: run __graft_entry__.build()"
    shutil.copy(so, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", "lib.so"], cwd=tmp_path, check=True, capture_output=True)
    objs = sorted(p for p in os.listdir(tmp_path) if "amdgcn" in p)
    assert objs and all(p.endswith("gfx950") for p in objs), objs
    text = "".join(subprocess.run([objdump, "-d", p], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
                   for p in objs)
    assert not re.search(r"v_fma_mix(lo|hi)_f16", text)
    assert text.count("v_mfma_f32_32x32x16_f16") >= 24 and "v_mfma_f32_32x32x2_f32" in text
