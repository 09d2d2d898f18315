"""The C++ side of the N > 1 path on the GPU box (one MI355X):

  * transport "rccl" with ONE rank: librccl really loads, ncclCommInitRank / ncclAllGather / ncclBroadcast really run on
    the context's stream (what an 8-GPU job calls, at world size 1);
  * two PROCESSES sharing the one GPU over the "socket" transport (RCCL refuses two ranks on a device): the sharded
    scoring round and the device-side ensemble exchange through the C ABI, and prv_planner's `shard: views` mode -- records
    and chosen views byte-identical to the one-process loop."""
import os
import socket
import subprocess

import numpy as np
import pytest

from nerf_prv_amd import api, planner
from tests import util
from tests.test_gpu_planner import GOLD, ROOT, SEED, YAML, small_desc

pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def feed_all_then_wait(procs, text, timeout=300):
    """ranks that talk to each other must ALL have their console input before anyone is waited for"""
    for p in procs:
        p.stdin.write(text)
        p.stdin.close()
        p.stdin = None  # communicate() must not touch it again
    return [p.communicate(timeout=timeout) for p in procs]


def scene(c, n_views=11, w=40, h=24):
    d = small_desc()
    for e in range(3):
        c.synthetic_model(e, d, SEED + e)
    c.synthetic_model(7, d, SEED + 99)
    tms, scale, offset = planner.hemisphere_transforms(util.fibonacci_hemisphere(n_views), 0.3, 0.1, [1e-10] * 3)
    cams = c.cameras_from_matrices(tms, util.FOV_X, w, h, scale, offset)
    return d, cams, api.render_opts(w, h, 64, 2, 0.01, background=(0, 0, 0, 1))


def test_rccl_transport_with_one_rank_runs_the_real_collectives(ctx):
    d, cams, opts = scene(ctx)
    comm = api.Comm(ctx, 0, 1, transport="rccl", rendezvous=f"127.0.0.1:{free_port()}")
    try:
        assert comm.transport == "rccl"
        t = ctx.torch
        send = t.arange(4096, dtype=t.int32, device=ctx.device)
        out = comm.all_gather(send)
        assert bool((out.view(t.int32) == send).all())  # ncclAllGather, one rank: the identity
        comm.barrier()
        # the sharded round through RCCL == the plain round
        for method, slots in ((api.L.SCORE_ENSEMBLE_RGB_DENSITY, [0, 1, 2]), (api.L.SCORE_ENSEMBLE_RGB, [0, 1])):
            want, _ = ctx.score_views(method, slots, cams, None, opts)
            got, st = comm.score_views(method, slots, cams, len(cams), opts, want_stats=True)
            assert got.tobytes() == want.tobytes() and st.samples_evaluated > 0
        gt, _ = ctx.render(7, cams, None, api.render_opts(opts.width, opts.height, 64, 2, 0.01))
        o5 = api.render_opts(opts.width, opts.height, 64, 2, 0.01)
        want, _ = ctx.score_views(api.L.SCORE_PSNR_COVERAGE, [0], cams, None, o5, gt=gt)
        got, _ = comm.score_views(api.L.SCORE_PSNR_COVERAGE, [0], cams, len(cams), o5, gt_shard=gt)
        assert got.tobytes() == want.tobytes()
        # ncclBroadcast group over the three members' buffers: one rank owns them all, nothing may change
        before = [[a.copy() for a in ctx.export_model(e, d)] for e in range(3)]
        comm.exchange_models(3, d)
        for e in range(3):
            assert all(np.array_equal(a, b) for a, b in zip(before[e], ctx.export_model(e, d)))
    finally:
        comm.close()
        cams.close()


def _comm_worker(rank, world, port, outdir):
    """one of two processes on the one GPU: own context, socket transport"""
    c = api.Context(0)
    comm = api.Comm(c, rank, world, transport="socket", rendezvous=f"127.0.0.1:{port}")
    try:
        d, cams, opts = scene(c)
        t = c.torch
        # (1) raw all-gather of device buffers
        send = t.full((1000,), rank + 1, dtype=t.int16, device=c.device)
        out = comm.all_gather(send).view(t.int16).cpu().numpy().reshape(world, 1000)
        assert all((out[r] == r + 1).all() for r in range(world))
        # (2) the sharded round: ragged (11 views over 2 ranks), both shard orders
        rec_i, st = comm.score_views(api.L.SCORE_ENSEMBLE_RGB_DENSITY, [0, 1, 2], cams, len(cams), opts, interleaved=True, want_stats=True)
        rec_b, _ = comm.score_views(api.L.SCORE_ENSEMBLE_RGB_DENSITY, [0, 1, 2], cams, len(cams), opts, interleaved=False)
        ids, per = api.shard_views(len(cams), rank, world, True)
        gt, _ = c.render(7, cams, ids, api.render_opts(opts.width, opts.height, 64, 2, 0.01), want_stats=False)
        rec_5, _ = comm.score_views(api.L.SCORE_PSNR_COVERAGE, [0], cams, len(cams), api.render_opts(opts.width, opts.height, 64, 2, 0.01),
                                    gt_shard=gt)
        # (3) the ensemble exchange: every rank re-makes ITS members differently (seed by rank), then exchanges
        for e in range(3):
            if e % world == rank:
                c.synthetic_model(e, d, 1000 + e)
            else:
                c.synthetic_model(e, d, 5)  # something else, to be overwritten
        comm.exchange_models(3, d)
        members = [c.export_model(e, d) for e in range(3)]
        after, _ = c.score_views(api.L.SCORE_ENSEMBLE_RGB_DENSITY, [0, 1, 2], cams, None, opts)  # renders with the exchanged fields
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), rec_i=rec_i.view(np.uint8), rec_b=rec_b.view(np.uint8), rec_5=rec_5.view(np.uint8),
                 after=after.view(np.uint8), evaluated=st.samples_evaluated,
                 **{f"m{e}_{k}": members[e][k] for e in range(3) for k in range(3)})
        comm.barrier()
    finally:
        comm.close()
        c.close()


def test_two_processes_on_one_gpu_shard_score_gather_and_exchange(ctx, tmp_path):
    import torch.multiprocessing as mp

    port = free_port()
    mctx = mp.get_context("spawn")
    procs = [mctx.Process(target=_comm_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(2))
    # the one-process answers
    d, cams, opts = scene(ctx)
    want, st = ctx.score_views(api.L.SCORE_ENSEMBLE_RGB_DENSITY, [0, 1, 2], cams, None, opts, want_stats=True)
    gt, _ = ctx.render(7, cams, None, api.render_opts(opts.width, opts.height, 64, 2, 0.01), want_stats=False)
    want5, _ = ctx.score_views(api.L.SCORE_PSNR_COVERAGE, [0], cams, None, api.render_opts(opts.width, opts.height, 64, 2, 0.01), gt=gt)
    for r in (r0, r1):
        assert r["rec_i"].tobytes() == want.tobytes() and r["rec_b"].tobytes() == want.tobytes()  # both shard orders, every rank
        assert r["rec_5"].tobytes() == want5.tobytes()
    assert int(r0["evaluated"]) + int(r1["evaluated"]) == st.samples_evaluated  # the shards evaluate exactly the whole
    # exchanged members: bit-identical on both ranks and equal to what their owners made
    for e in range(3):
        ctx.synthetic_model(e, d, 1000 + e)
        mine = ctx.export_model(e, d)
        for k in range(3):
            assert np.array_equal(r0[f"m{e}_{k}"], r1[f"m{e}_{k}"]) and np.array_equal(r0[f"m{e}_{k}"], mine[k])
    after, _ = ctx.score_views(api.L.SCORE_ENSEMBLE_RGB_DENSITY, [0, 1, 2], cams, None, opts)
    assert r0["after"].tobytes() == r1["after"].tobytes() == after.tobytes()  # and they RENDER the same (derived state rebuilt)
    cams.close()


@pytest.mark.parametrize("method", [3, 5])
def test_planner_views_sharded_over_two_ranks_equals_the_one_process_loop(ctx, tmp_path, method):
    """prv_planner mode 21 with `shard: views` under RANK / WORLD_SIZE / LOCAL_RANK (two processes on the one GPU, socket
    transport): every iteration's gathered records and the chosen views are byte-identical to the one-process loop, on
    BOTH ranks (static members: in-process training is not bit-reproducible run to run -- float atomics)."""
    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    runs = {}
    for who in ("one", "two"):
        pre = tmp_path / f"{who}_{method}"
        pre.mkdir()
        cfg = pre / "cfg.yaml"
        cfg.write_text(YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=method,
                                   model_source=f"synthetic_seed: {SEED}\npretrained_members: 1\nshard: \"views\""))
        base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        base.update(PRV_PLANNER_DUMP_RECORDS="1")
        if who == "one":
            out = subprocess.run([exe, str(cfg)], input="21\nobjA\n-1\n", text=True, capture_output=True, timeout=300, env=base)
            assert out.returncode == 0, out.stdout + out.stderr
            outs = [out.stdout]
        else:
            port = free_port()
            procs = []
            for r in range(2):
                env = dict(base, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), PRV_COMM="socket", MASTER_ADDR="127.0.0.1",
                           PRV_COMM_PORT=str(port))
                procs.append(subprocess.Popen([exe, str(cfg)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                              text=True, env=env))
            res = feed_all_then_wait(procs, "21\nobjA\n-1\n")
            for p, (so, se) in zip(procs, res):
                assert p.returncode == 0, so + se
            outs = [so for so, _ in res]
            assert all("views sharded, transport socket" in so for so in outs)
        chosen = [[int(x) for x in [l for l in so.splitlines() if l.startswith("chosen_nbvs:")][-1].split(":")[1].split()] for so in outs]
        recs = []
        for r in range(len(outs)):
            root = pre if r == 0 else pre / f"rank{r}"
            save = root / "Compare" / "ShapeNet" / f"objA_m{method}_v1_t0"
            recs.append([(save / "records" / f"{it}.bin").read_bytes() for it in range(3)])
        runs[who] = (chosen, recs)
    (c1, r1), (c2, r2) = runs["one"], runs["two"]
    assert c2[0] == c2[1] == c1[0] and len(c1[0]) == 4
    assert r2[0] == r2[1] == r1[0] and all(len(b) == 16 * (4 - it) for it, b in enumerate(r1[0]))  # 5 views, 4..2 candidates left


def test_planner_views_sharded_trains_its_members_on_their_owners(ctx, tmp_path):
    """the training half in `shard: views` mode: member e is trained by rank e % 2, exchanged device-side, both ranks
    score with the whole ensemble -- both ranks end every iteration with byte-identical records and the same views"""
    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    pre = tmp_path / "train_views"
    pre.mkdir()
    cfg = pre / "cfg.yaml"
    text = YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=2,
                       model_source="n_steps: 40\ntrain_rays: 1024\ntrain_width: 64\ntrain_height: 36\nground_truth_seed: 4242\nshard: \"views\"")
    cfg.write_text(text)
    port = free_port()
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    procs = []
    for r in range(2):
        env = dict(base, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), PRV_COMM="socket", MASTER_ADDR="127.0.0.1",
                   PRV_COMM_PORT=str(port), PRV_PLANNER_DUMP_RECORDS="1", PRV_PLANNER_TIMING="1")
        procs.append(subprocess.Popen([exe, str(cfg)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    res = feed_all_then_wait(procs, "21\nobjA\n-1\n")
    for p, (so, se) in zip(procs, res):
        assert p.returncode == 0, so + se
    chosen = [[int(x) for x in [l for l in so.splitlines() if l.startswith("chosen_nbvs:")][-1].split(":")[1].split()] for so, _ in res]
    assert chosen[0] == chosen[1] and len(set(chosen[0])) == 4
    for it in range(3):
        a = (pre / "Compare" / "ShapeNet" / "objA_m2_v1_t0" / "records" / f"{it}.bin").read_bytes()
        b = (pre / "rank1" / "Compare" / "ShapeNet" / "objA_m2_v1_t0" / "records" / f"{it}.bin").read_bytes()
        assert a == b and len(a) == 16 * (4 - it)
        assert np.isfinite(np.frombuffer(a, api.RECORD_DTYPE)["score"]).all()
    assert all(se.count("train_members:") == 3 for _, se in res)  # both ranks went through the training step each iteration


def test_planner_members_sharded_deals_the_trainings_and_walks_the_objects_in_lockstep(ctx, tmp_path):
    """prv_planner `shard: members` (BASELINE configs[4] with more GPUs than objects): two objects, method 2 (two members
    each), two ranks on the one GPU over the socket transport.  The four (object, member) trainings of a lockstep round are
    dealt round-robin -- pair object * 2 + member to rank pair % 2: each rank trains ONE member of EACH object, side by side --,
    then object by object the members are exchanged and the candidates scored views-sharded: both ranks end every round of
    both objects with byte-identical records and the same chosen views.  The same yaml on ONE rank walks the same lockstep
    rounds with all four trainings on that rank."""
    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    names = ["objA", "objB"]

    def run(world, pre):
        pre.mkdir()
        cfg = pre / "cfg.yaml"
        cfg.write_text(YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=2,
                                   model_source="n_steps: 40\ntrain_rays: 1024\ntrain_width: 64\ntrain_height: 36\nground_truth_seed: 4242\nshard: \"members\""))
        base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        base.update(PRV_PLANNER_DUMP_RECORDS="1", PRV_PLANNER_TIMING="1")
        port = free_port()
        procs = []
        for r in range(world):
            env = dict(base, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), PRV_COMM="socket", MASTER_ADDR="127.0.0.1", PRV_COMM_PORT=str(port)) if world > 1 else base
            procs.append(subprocess.Popen([exe, str(cfg)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
        res = feed_all_then_wait(procs, "21\n" + "\n".join(names) + "\n-1\n")
        for p, (so, se) in zip(procs, res):
            assert p.returncode == 0, so + se
        return res

    two = run(2, tmp_path / "two")
    for r, (so, se) in enumerate(two):
        assert "member trainings dealt round-robin" in so and so.count("lockstep batch of 2") == 2
        # three lockstep rounds, in each: 2 trainings on this rank (one member of each object), side by side
        assert se.count("train_pairs: 2 (object, member) trainings of this round on rank %d" % r) == 3, se
        assert "train_members:" not in se  # nobody trained inside a scoring call
    chosen = [[[int(x) for x in l.split(":")[1].split()] for l in so.splitlines() if l.startswith("chosen_nbvs:")] for so, _ in two]
    assert chosen[0] == chosen[1] and len(chosen[0]) == 2 and all(len(set(c)) == 4 for c in chosen[0])
    for name in names:
        for it in range(3):
            a = (tmp_path / "two" / "Compare" / "ShapeNet" / f"{name}_m2_v1_t0" / "records" / f"{it}.bin").read_bytes()
            b = (tmp_path / "two" / "rank1" / "Compare" / "ShapeNet" / f"{name}_m2_v1_t0" / "records" / f"{it}.bin").read_bytes()
            assert a == b and len(a) == 16 * (4 - it) and np.isfinite(np.frombuffer(a, api.RECORD_DTYPE)["score"]).all()
    one = run(1, tmp_path / "one")
    so, se = one[0]
    assert se.count("train_pairs: 4 (object, member) trainings of this round on rank 0") == 3 and so.count("chosen_nbvs:") == 2


def test_communicator_outliving_its_context_is_inert():
    """destroying the context first (interpreter shutdown order) must not leave the communicator pointing at freed memory"""
    c2 = api.Context(0)
    comm = api.Comm(c2, 0, 1, transport="socket")
    lib, h = c2.lib, comm.handle
    c2.close()
    assert lib.prv_comm_barrier(h) == api.L.PRV_E_INVALID  # inert: an error code, no crash
    lib.prv_comm_destroy(h)
    comm.handle = None


def test_bench_two_ranks_time_the_product_collective_path(tmp_path):
    """`python bench.py --gpus 2`: the launcher starts two ranks; each brings up the C ABI's communicator and the TIMED
    region is prv_score_views_sharded (render + score the shard, ONE all-gather of the records inside libprv_hip.so), with
    torch.distributed's gather as the untimed cross-check.  On this one-GPU box the ranks share the device
    (PRV_BENCH_SHARED_GPU=1: gloo + the socket transport, RCCL refuses ranks that share a GPU); an 8-GPU node runs the
    same code with RCCL and needs no change."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PRV_BENCH_SHARED_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--views-per-gpu", "3",
           "--width", "96", "--height", "80", "--no-extras", "--no-training", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["config"]["views_total"] == 6 and two["value"] > 0
    col = two["collective"]
    assert col["timed_path"].startswith("prv_score_views_sharded") and "socket" in col["timed_path"]
    assert col["records_identical_to_torch_gather"] is True and col["error"] is None and two["comm_ranks"] == 2
    # the same six views on one rank: the same ranking
    one = subprocess.run(cmd[:3] + ["1"] + cmd[4:8] + ["--views-per-gpu", "6"] + cmd[10:], capture_output=True, text=True,
                         env={k: v for k, v in env.items() if k != "PRV_BENCH_SHARED_GPU"}, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stdout + one.stderr
    ref = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert ref["ranking_head"] == two["ranking_head"] and ref["collective"] is None


def test_bench_eight_ranks_shard_1024_views_of_the_512_field(tmp_path):
    """BASELINE configs[3]'s layout before the driver's 8-GPU run: `bench.py --gpus 8 --mode strong --views-total 1024
    --field 512` -- eight fresh rank processes, 128 views each (interleaved), the 512^3 F=2 field, ONE all-gather of the
    records per round through the C ABI's communicator.  The pool has one GPU, so the eight ranks share it (gloo + the socket
    transport; small images): the gathered records and the whole integer ranking are byte-identical to ONE process
    scoring all 1024 views.  The parent (this test) never touches the GPU for it."""
    import json
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--mode", "strong", "--views-total", "1024",
            "--field", "512", "--width", "40", "--height", "32", "--no-extras", "--no-training", "--no-cpu-baseline", "--no-full-loop"]
    out = subprocess.run(base + ["--gpus", "8"], capture_output=True, text=True, env=dict(env, PRV_BENCH_SHARED_GPU="1"), timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    eight = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert eight["n_gpus"] == 8 and eight["comm_ranks"] == 8 and eight["config"]["views_total"] == 1024
    col = eight["collective"]
    assert col["timed_path"].startswith("prv_score_views_sharded") and col["transport"] == "socket" and col["ranks"] == 8
    assert col["records_identical_to_torch_gather"] is True and col["error"] is None
    one = subprocess.run(base + ["--gpus", "1"], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    ref = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert ref["config"]["views_total"] == 1024 and ref["collective"] is None
    assert eight["records_sha256"] == ref["records_sha256"]  # 1024 records, byte for byte
    assert eight["ranking_sha256"] == ref["ranking_sha256"] and eight["ranking_head"] == ref["ranking_head"]
    assert eight["samples_evaluated_per_step_per_gpu"] > 0
    # ... and what the driver's `bench.py --gpus 8` (the WEAK headline) carries beside it: the `config3` sub-object -- the same
    # 1024 views of the 512^3 field, sharded 128 per rank on the same communicator, timed after the headline
    weak = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--views-per-gpu", "2", "--width", "40", "--height", "32",
            "--no-extras", "--no-training", "--no-cpu-baseline", "--no-full-loop", "--gpus", "8"]
    out = subprocess.run(weak, capture_output=True, text=True, env=dict(env, PRV_BENCH_SHARED_GPU="1"), timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert line["scaling"] == "weak" and line["config"]["views_total"] == 16 and line["comm_ranks"] == 8
    c3 = line["config3"]
    assert "error" not in c3, c3
    assert c3["views_per_gpu"] == 128 and c3["comm_ranks"] == 8 and c3["scaling"] == "strong" and c3["views_per_s"] > 0
    assert c3["records_sha256"] == ref["records_sha256"] and c3["ranking_sha256"] == ref["ranking_sha256"]
    assert line["config3_views_per_s"] == c3["views_per_s"]
    # which librccl: the socket transport loads none of its own, and the process maps at most the one torch links
    col = line["collective"]
    assert col["librccl_prv_comm"] == {"path": "", "version": 0, "found": ""} and col["one_rccl_per_process"] is True
    assert isinstance(col["librccl_mapped"], list) and len(col["librccl_mapped"]) <= 1


def test_bench_rccl_communicator_runs_on_the_librccl_the_process_already_has(tmp_path):
    """`PRV_FORCE_DIST=1 python bench.py`: torch.distributed comes up on "nccl" (torch's bundled librccl) and THEN the C ABI's
    communicator on transport "rccl": it must resolve to the SAME library file (dlopen RTLD_NOLOAD of what the process has
    mapped), not to a second copy found by soname -- one RCCL per process.  One rank: the calls are the real
    ncclCommInitRank / ncclAllGather."""
    import json
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PRV_RCCL_LIB")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--views-per-gpu", "3", "--width", "96", "--height", "80",
           "--no-extras", "--no-training", "--no-cpu-baseline", "--no-full-loop"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, PRV_FORCE_DIST="1", MASTER_PORT=str(free_port())), timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    col = line["collective"]
    assert col["transport"] == "rccl" and line["rccl_ranks"] == 1 and col["records_identical_to_torch_gather"] is True
    lib = col["librccl_prv_comm"]
    assert lib["path"] and lib["version"] > 0, lib
    assert len(col["librccl_mapped"]) == 1 and os.path.realpath(col["librccl_mapped"][0]) == os.path.realpath(lib["path"]), col
    assert lib["found"] == "already mapped by the process" and col["one_rccl_per_process"] is True


def _rccl_shared_device_worker(rank, port, q):
    c = api.Context(0)
    try:
        api.Comm(c, rank, 2, transport="rccl", rendezvous=f"127.0.0.1:{port}")
        q.put((rank, "created"))
    except api.PrvError as e:
        q.put((rank, f"{e.code}: {e}"))
    finally:
        c.close()


def test_rccl_transport_refuses_two_ranks_on_one_gpu_with_a_reason(tmp_path):
    """two ranks that sit on the same physical GPU ask for the rccl transport: BOTH get PRV_E_INVALID naming the ranks and
    the device, promptly -- not a hang inside ncclCommInitRank"""
    import time

    import torch.multiprocessing as mp

    port = free_port()
    mctx = mp.get_context("spawn")
    q = mctx.Queue()
    t0 = time.time()
    procs = [mctx.Process(target=_rccl_shared_device_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert time.time() - t0 < 170
    for r in range(2):
        assert got[r].startswith(str(api.L.PRV_E_INVALID)) and "ranks 0 and 1 share one GPU" in got[r] and "socket" in got[r], got
