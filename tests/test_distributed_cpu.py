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
