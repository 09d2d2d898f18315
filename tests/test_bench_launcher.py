"""`python bench.py --gpus N` must run by itself (the driver invokes exactly that): with no WORLD_SIZE in the
environment the process launches N ranks under torch.distributed.run, never touches a GPU, forwards rank 0's JSON
line and exits with the children's code.  Exercised here on CPU in the bench's dry-run mode (gloo, records that are a
fixed function of the view id, no performance figure)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, env=env,
                          timeout=300, cwd=ROOT)


@pytest.mark.parametrize("mode,extra,n_views", [("weak", ["--views-per-gpu", "5"], 10),
                                                 ("strong", ["--mode", "strong", "--views-total", "9"], 9)])
def test_plain_invocation_with_two_gpus_launches_its_own_ranks(mode, extra, n_views):
    out = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"] + extra, {"PRV_BENCH_DRY_RUN": "1"})
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # ONE JSON line, rank 0's
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["dry_run"] is True and two["value"] is None and two["scaling"] == mode
    assert two["views_total"] == n_views
    # the single-process run of the same candidate set ends with the same gathered records and ranking
    one_extra = ["--views-per-gpu", "10"] if mode == "weak" else extra
    one = run_bench(["--gpus", "1", "--steps", "1", "--warmup", "0"] + one_extra, {"PRV_BENCH_DRY_RUN": "1"})
    assert one.returncode == 0, one.stdout + one.stderr
    ref = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert ref["records_checksum"] == two["records_checksum"] and ref["ranking_head"] == two["ranking_head"]
    # N > 1 carries BASELINE configs[3] beside the headline: 1024 views sharded 1024 / N per rank, the same records and ranking
    # as one process scoring all of them
    c3 = two["config3"]
    assert c3["views_per_gpu"] == 512 and c3["comm_ranks"] == 2 and c3["scaling"] == "strong" and ref["config3"] is None
    if mode == "weak":
        all1024 = run_bench(["--gpus", "1", "--steps", "1", "--warmup", "0", "--mode", "strong", "--views-total", "1024"], {"PRV_BENCH_DRY_RUN": "1"})
        r3 = json.loads([l for l in all1024.stdout.splitlines() if l.startswith("{")][0])
        assert r3["records_checksum"] == c3["records_checksum"] and r3["ranking_head"] == c3["ranking_head"]


def test_launcher_passes_a_failing_rank_on():
    """a rank that dies (here: no GPU and no dry run -> the context refuses to exist) makes the launcher exit non-zero
    with no JSON line: a failed N-GPU run can not look like a result"""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the ranks would run")
    out = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras", "--no-training"], {})
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_mismatched_world_size_is_refused():
    out = run_bench(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "PRV_BENCH_DRY_RUN": "1"})
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stdout + out.stderr)


def test_roofline_peak_constants_are_the_committed_microbenchmarks():
    """bench.py prices the render kernel's instructions at issue costs MEASURED on the chip (scripts/valu_rate.hip ->
    profiles/r04_valu_issue_rate.txt): the constants in bench.py are the file's figures (grid 1, four waves per SIMD, the
    slowest wave), class by class, and the floor's instruction count is DESIGN.md's (647 / 913 per 64 samples)"""
    import importlib.util
    import re

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    text = open(os.path.join(ROOT, bench.ISSUE_RATE_FILE)).read()
    grid1 = text.split("## grid 1 block")[1].split("## grid 256")[0]

    def slowest_at_four_waves(name):
        line = next(l for l in grid1.splitlines() if l.startswith(name + " "))
        return float(re.findall(r"\[\s*([0-9.]+)\]", line)[3])

    c = bench.ISSUE_CYCLES
    for op in ("v_fma_f32", "v_mul_f32", "v_add_u32", "v_xor_b32", "v_bitop3_b32", "v_sub_f32"):
        assert abs(slowest_at_four_waves(op) - c["c2"]) <= 0.08, op
    for op in ("v_pk_fma_f16", "v_pk_mul_f16", "v_pk_max_f16", "v_cvt_pk_f16_f32", "v_fract_f32", "v_mul_u32_u24", "v_cvt_i32_f32"):
        assert abs(slowest_at_four_waves(op) - c["c4"]) <= 0.2, op
    for op in ("v_exp_f32", "v_rcp_f32", "v_permlane32_swap"):
        assert abs(slowest_at_four_waves(op) - c["c8"]) <= 0.2, op
    # an MFMA between 24 fillers: (time per MFMA - 24 fillers at the c4 cost) = what the MFMA itself holds the issue port for
    line = next(l for l in grid1.splitlines() if l.startswith("v_mfma + 24 VALU"))
    per_mfma = float(re.findall(r"\[\s*([0-9.]+)\]", line)[3])
    assert 6.0 <= per_mfma - 24 * c["c4"] <= 12.0 and c["mfma"] == 8.0
    assert bench.isa_floor(8, 5)["valu_per_64_samples"] == 647 and bench.isa_floor(16, 10)["valu_per_64_samples"] == 913
    f = bench.isa_floor(8, 5)
    assert abs(f["issue_cycles_per_64_samples"] - (f["c2"] * c["c2"] + f["c4"] * c["c4"] + f["c8"] * c["c8"] + 40 * 8.0)) < 1e-9
    assert abs(bench.ISSUE_PEAK_GCYC - 1024 * 2.4) < 1e-9


def test_trainer_atomic_bound_constants_are_the_committed_measurements():
    """the trainer's `atomic_bound` in the bench line: the request rate is scripts/atomic_rate.hip's (every shape and
    occupancy of profiles/r04_atomic_request_rate.txt within 8 % of it), the requests per sample the backward tile kernel's
    TCC_EA0_ATOMIC over the composited samples of the runs profiles/r06_train_rules.txt records (one per sampling rule)"""
    import importlib.util
    import re

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rates = [float(l.split()[-2]) for l in open(os.path.join(ROOT, bench.ATOMIC_RATE_FILE)) if re.match(r"^\d+ x \d+ B( \(16 lanes active\))?\s+\d+\s+[0-9.]+\s", l)]
    assert len(rates) >= 20 and all(abs(r - bench.ATOMIC_REQ_PEAK_G) <= 0.08 * bench.ATOMIC_REQ_PEAK_G for r in rates), rates
    text = open(os.path.join(ROOT, bench.ATOMIC_REQ_FILE)).read()
    for rule, want in bench.ATOMIC_REQ_PER_SAMPLE.items():  # round 6: re-measured at upstream's batch under both sampling rules
        m = re.search(rf"^rule {rule}: .*TCC_EA0_ATOMIC per launch (\d+), composited samples per step (\d+), requests per composited sample ([0-9.]+)", text, re.M)
        per_sample = float(m.group(1)) / float(m.group(2))
        assert abs(per_sample - want) <= 0.01 * want and abs(float(m.group(3)) - want) <= 0.01 * want, (rule, per_sample)


def test_roofline_record_is_flat_and_leads_with_the_contract_keys():
    """the driver's parser kept the first 22 keys of `roofline` in rounds 2-3 and lost `frac`: the object is flat (scalars
    and one string only), at most 22 entries, and starts with kernel / bound / frac / peak / achieved / unit / traffic; both
    kinds (issue-bound 256^3, HBM-bound 512^3) and a run without any profile file behind it"""
    import importlib.util
    import types

    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    st = types.SimpleNamespace(samples_evaluated=214798195, wave_rounds=7149040)
    m = {"prof": {"render_ms": 8.45 * 20, "render_launches": 20, "march_ms": 11.4, "march_launches": 20, "clock_ghz": 1.9}, "st": st, "steps": 20}
    for variant, hbm, layout in (("64<4, 5>", False, {"n_levels": 8, "n_dense_levels": 5}), ("64<2, 10>", True, {"n_levels": 16, "n_dense_levels": 10}),
                                 ("64<4, 3>", False, {"n_levels": 8, "n_dense_levels": 3})):  # the last: no profile entry
        ms = 26.3 if hbm else 8.45  # the 512^3 launch is three times as long
        m["prof"]["render_ms"] = ms * 20
        roof, detail = bench.kernel_figures(m, variant, hbm, "baseline", layout)
        keys = list(roof)
        assert keys[:7] == ["kernel", "bound", "frac", "peak", "achieved", "unit", "traffic"], keys
        assert len(keys) <= 22 and all(not isinstance(v, (dict, list)) for v in roof.values())
        assert roof["bound"] == ("fabric_request_rate" if hbm else "valu_issue") and 0.0 < roof["frac"] < 1.0
        assert (abs(roof["fabric_side_frac"] - roof["frac"]) < 1e-9) if hbm else roof["fabric_side_frac"] is None  # the 512^3 instance: what reaches the fabric over the calibration
        assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
        assert roof["units_per_launch"] == 214798195 and abs(roof["avg_launch_ms"] - ms) < 1e-9
        assert isinstance(detail, dict) and "floor" in detail
    # the issue-bound fraction is the floor's: 2,720.7 cycles per 64 samples over 1024 SIMDs x 2.4 GHz
    m["prof"]["render_ms"] = 8.45 * 20
    roof, _ = bench.kernel_figures(m, "64<4, 5>", False, "baseline", {"n_levels": 8, "n_dense_levels": 5})
    want = 214798195 / 64 * bench.isa_floor(8, 5)["issue_cycles_per_64_samples"] / 8.45e-3 / (1024 * 2.4e9)
    assert abs(roof["frac"] - want) < 1e-12
