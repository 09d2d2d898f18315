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
