"""The N>1 path on CPU: two gloo ranks shard the candidate views, score their shard, do ONE
all-gather of 16-byte records and must produce the single-process result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerf_prv_amd import planner
from nerf_prv_amd.api import RECORD_DTYPE
from tests import util


def _scores(n_views):
    """deterministic stand-in for the HIP scorer: the oracle's PSNR/coverage of tiny renders"""
    from oracle import oracle as orc

    fa = orc.OracleField(orc.desc(**util.SMALL), seed=util.SEED_A)
    fb = orc.OracleField(orc.desc(**util.SMALL), seed=util.SEED_B)
    tms, scale, offset = planner.hemisphere_transforms(planner.hemisphere_generate(n_views), 0.3, 0.1, [1e-10] * 3)
    cams = orc.cameras_from_transforms(tms, util.FOV_X, 12, 12, scale, offset)

    def score_shard(ids):
        rec = np.zeros(len(ids), RECORD_DTYPE)
        for k, v in enumerate(ids):
            a, _ = fa.render(cams[v], 12, 12, 32, 1, 1e-4, threads=1)
            b, _ = fb.render(cams[v], 12, 12, 32, 1, 1e-4, threads=1)
            rec[k] = orc.score_view(a, b)
        return rec

    return score_shard


def _worker(rank, world, port, n_views, q, interleaved=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        records, order = planner.scoring_round(n_views, _scores(n_views), interleaved=interleaved)
        q.put((rank, records.tobytes(), order.tolist()))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("interleaved", [False, True])
@pytest.mark.parametrize("n_views", [7, 8])  # ragged and even shards
def test_two_rank_scoring_round_equals_single_process(n_views, interleaved):
    ref_records, ref_order = planner.scoring_round(n_views, _scores(n_views))  # world = 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_views, q, interleaved)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, rec_bytes, order in got:
        assert rec_bytes == ref_records.tobytes()  # identical gathered array on every rank
        assert order == ref_order.tolist()  # identical integer ranking


def test_shard_layout():
    ids, per = planner.shard_views(1024, 3, 8)
    assert per == 128 and ids[0] == 384 and ids[-1] == 511  # BASELINE config 4: 1024 views, 128/GPU
    ids, per = planner.shard_views(7, 1, 2)
    assert per == 4 and ids.tolist() == [4, 5, 6]
    ids, per = planner.shard_views(3, 7, 8)
    assert per == 1 and len(ids) == 0  # more ranks than views: empty shard is legal
    cover = np.concatenate([planner.shard_views(1000, r, 8)[0] for r in range(8)])
    assert cover.tolist() == list(range(1000))
    ids, per = planner.shard_views(10, 1, 4, interleaved=True)
    assert per == 3 and ids.tolist() == [1, 5, 9]
    cover = np.concatenate([planner.shard_views(1001, r, 8, interleaved=True)[0] for r in range(8)])
    assert sorted(cover.tolist()) == list(range(1001))
    assert all(len(planner.shard_views(1001, r, 8, interleaved=True)[0]) <= 126 for r in range(8))


def _member(e, sizes):
    """a recognisable fake field for member e"""
    rng = np.random.default_rng(100 + e)
    return (rng.integers(0, 65536, sizes[0], dtype=np.uint16), rng.integers(0, 65536, sizes[1], dtype=np.uint16),
            rng.integers(0, 2 ** 32, sizes[2], dtype=np.uint32))


def _exchange_worker(rank, world, port, n_members, sizes, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        local = {e: _member(e, sizes) for e in range(n_members) if planner.member_owner(e, world) == rank}
        got = planner.exchange_members(local, n_members, sizes)
        q.put((rank, {e: [a.tobytes() for a in got[e]] for e in got}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_members", [5, 2, 1])  # uneven shares, one each, fewer members than ranks
def test_two_rank_ensemble_exchange(n_members):
    """the exchange step of a multi-GPU NBV iteration: every rank ends up with every member, bit for bit"""
    sizes = (4096 + 8, 10240, 129)
    want = {e: [a.tobytes() for a in _member(e, sizes)] for e in range(n_members)}
    single = planner.exchange_members({e: _member(e, sizes) for e in range(n_members)}, n_members, sizes)  # world = 1
    assert {e: [a.tobytes() for a in single[e]] for e in single} == want
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, n_members, sizes, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, fields in got:
        assert fields == want
    with pytest.raises(ValueError):
        planner.exchange_members({0: _member(0, sizes)}, 2, sizes)  # world 1 must bring both members


# ---- the C++ side of the N > 1 path: the TCP star behind prv_comm's rendezvous and its socket transport -------

def _star_worker(rank, world, port, q):
    import ctypes as C

    h = planner.host()
    star = h.prvh_star_open(rank, world, b"127.0.0.1", port, 60.0)
    assert star, "star rendezvous failed"
    try:
        # (1) all-gather of ragged-looking 16-byte records, as the scoring round gathers them
        per = 3
        send = np.zeros(per, RECORD_DTYPE)
        send["score"] = np.arange(per) + 100.0 * rank
        send["psnr"] = rank
        recv = np.zeros(per * world, RECORD_DTYPE)
        assert h.prvh_star_all_gather(star, send.ctypes.data_as(C.c_void_p), send.nbytes, recv.ctypes.data_as(C.c_void_p)) == 0
        # (2) a large block (the ensemble exchange's size class) broadcast from a non-zero root
        big = np.full(3_000_001, rank, np.uint8)
        if rank == world - 1:
            big[:] = np.arange(big.size) % 251
        assert h.prvh_star_broadcast(star, big.ctypes.data_as(C.c_void_p), big.nbytes, world - 1) == 0
        assert h.prvh_star_barrier(star) == 0
        q.put((rank, recv.tobytes(), int(big.astype(np.int64).sum())))
    finally:
        h.prvh_star_close(star)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_star_all_gather_and_broadcast_between_processes(world):
    """the rendezvous / socket transport of the C ABI's communicator, without a GPU: every rank ends with the same
    gathered records in rank order, a broadcast from the last rank reaches everyone"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_star_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.zeros(3 * world, RECORD_DTYPE)
    for r in range(world):
        want["score"][3 * r: 3 * r + 3] = np.arange(3) + 100.0 * r
        want["psnr"][3 * r: 3 * r + 3] = r
    want_sum = int((np.arange(3_000_001) % 251).sum())
    for rank, rec_bytes, s in got:
        assert rec_bytes == want.tobytes() and s == want_sum


def _star_odd_worker(rank, world, port, q, case):
    """rendezvous cases beyond the happy path; puts (rank, ok, message, seconds)"""
    import time

    h = planner.host()
    t0 = time.time()
    if case == "world_mismatch":  # rank 1 believes in 3 ranks, rank 0 in 2: refused with the reason, at once
        star = h.prvh_star_open(rank, 3 if rank == 1 else 2, b"127.0.0.1", port, 8.0 if rank == 0 else 30.0)
    elif case == "failed_first_open":  # rank 1's first communicator never got as far as the rendezvous; the second one must still work
        if rank == 1:
            assert not h.prvh_star_open(5, 2, b"127.0.0.1", port, 5.0)  # bad rank: fails before any connection
        star = h.prvh_star_open(rank, world, b"127.0.0.1", port, 30.0)
    elif case == "name_and_foreign_address":
        # rank 0 is given an address that is not one of this host's (a service / NAT address: bind -> EADDRNOTAVAIL):
        # it listens on every interface instead of spinning until the timeout; rank 1 reaches it through a NAME
        star = h.prvh_star_open(rank, world, b"192.0.2.1" if rank == 0 else b"localhost", port, 30.0)
    msg = "" if star else h.prvh_share_data_error().decode()
    ok = bool(star)
    if star and case != "world_mismatch":
        ok = h.prvh_star_barrier(star) == 0
    q.put((rank, ok, msg, time.time() - t0))
    if star:
        h.prvh_star_close(star)


@pytest.mark.parametrize("case", ["world_mismatch", "failed_first_open", "name_and_foreign_address"])
def test_star_rendezvous_corner_cases(case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_star_odd_worker, args=(r, 2, port, q, case)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (ok, msg, took)) for r, ok, msg, took in (q.get(timeout=90) for _ in range(2)))
    for p in procs:
        p.join(timeout=30)
    if case == "world_mismatch":
        ok, msg, took = got[1]
        assert not ok and "refused" in msg and "2 ranks" in msg and took < 5.0, got  # an explicit answer, not a timeout
        assert not got[0][0] and "did not arrive" in got[0][1]  # rank 0 keeps its seat free until ITS timeout
    else:
        assert got[0][0] and got[1][0], got


def test_star_reports_a_bind_error_at_once():
    """a bind failure that waiting cannot cure (here: a privileged port as an ordinary user, or an address family error)
    is reported immediately with its errno text -- round 3 retried every bind error for the whole timeout"""
    import time

    if os.geteuid() == 0:
        pytest.skip("root may bind any port")
    h = planner.host()
    t0 = time.time()
    star = h.prvh_star_open(0, 2, b"127.0.0.1", 1, 30.0)
    assert not star and time.time() - t0 < 3.0
    assert "bind" in h.prvh_share_data_error().decode()


def test_shard_views_of_the_c_abi_matches_the_python_sharding():
    from nerf_prv_amd import api

    for n, world in ((0, 1), (7, 2), (8, 2), (1024, 8), (5, 8), (13, 4)):
        for interleaved in (False, True):
            for r in range(world):
                ids_c, per_c = api.shard_views(n, r, world, interleaved)
                ids_p, per_p = planner.shard_views(n, r, world, interleaved)
                assert per_c == per_p and np.array_equal(ids_c, ids_p), (n, world, interleaved, r)


def _star_stuck_worker(rank, world, port, q):
    import ctypes as C
    import time

    os.environ["PRV_COMM_TIMEOUT_S"] = "2"
    h = planner.host()
    star = h.prvh_star_open(rank, world, b"127.0.0.1", port, 60.0)
    assert star
    buf = np.zeros(8, np.uint8)
    out = np.zeros(16, np.uint8)
    if rank == 0:
        t0 = time.time()
        rc = h.prvh_star_all_gather(star, buf.ctypes.data_as(C.c_void_p), 8, out.ctypes.data_as(C.c_void_p))  # rank 1 never sends
        q.put((rc, time.time() - t0))
    else:
        time.sleep(6)  # alive, connected, silent
    h.prvh_star_close(star)


def test_star_gives_up_on_a_silent_peer_instead_of_hanging():
    """a rank that is connected but never answers makes the collective FAIL after PRV_COMM_TIMEOUT_S, it does not hang
    the job for ever (the reference's hand-shake polls for ever, main.cpp:1695-1698)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_star_stuck_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    rc, took = q.get(timeout=60)
    for p in procs:
        p.join(timeout=30)
    assert rc != 0 and 1.5 < took < 10
