"""instant-ngp snapshots on the GPU: a snapshot in the upstream layout (written by the independent Python writer of
tests/test_ingp.py) loads through prv_model_load_ingp, renders byte for byte like the same arrays loaded through
prv_model_load, matches the oracle on the level geometry tiny-cuda-nn's recipe gives (per_level_scale), and
prv_model_save_ingp writes it back.  BASELINE configs[2]'s input path; layout assumed from upstream, unpinned."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from nerf_prv_amd import api, planner
from tests import util
from tests.test_ingp import msgpack, upstream_snapshot

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

# instant-ngp's own nerf/base.json shape at aabb_scale 1: 16 levels x 2 features, 2^19 entries, base 16,
# per_level_scale = exp(ln(2048 / 16) / 15): the finest level has 2048 cells per axis
NGP = dict(n_levels=16, n_features=2, log2_hashmap=19, base_res=16, finest_res=2048, occ_res=128, density_bias=0.0, table_amp=4.0,
           per_level_scale=float(np.float32(np.exp(np.log(2048.0 / 16.0) / 15.0))))
SMALL_NGP = dict(n_levels=8, n_features=4, log2_hashmap=14, base_res=8, finest_res=96, occ_res=32, density_bias=0.0, table_amp=4.0,
                 per_level_scale=1.4262)


def snapshot_of(oracle, kw, seed, path):
    """a field with sensible weights (the oracle's synthetic generator on tiny-cuda-nn's level geometry) laid out as
    instant-ngp lays a snapshot out; density = exp(out) has no bias upstream, so the first density-output column of the
    second density layer is what carries the opacity here"""
    f = oracle.OracleField(oracle.desc(**kw), seed=seed)
    table, mlp, occ = f.params()
    R = kw["occ_res"]
    bits = np.unpackbits(occ.view(np.uint8), bitorder="little")[: R ** 3].reshape(R, R, R)  # [z][y][x]
    dens = np.where(bits > 0, np.float16(0.5), np.float16(0.0))
    fd = dict(n_levels=kw["n_levels"], n_features=kw["n_features"], log2_T=kw["log2_hashmap"], base=kw["base_res"], pls=kw["per_level_scale"],
              grid=R, table=table, mlp=mlp, density_xyz=dens)
    # vectorised Morton scatter for the 128^3 grid
    root = upstream_snapshot(dict(fd, grid=2, density_xyz=dens[:2, :2, :2]))  # cheap skeleton; the grid is replaced below
    z, y, x = np.meshgrid(np.arange(R), np.arange(R), np.arange(R), indexing="ij")

    def expand(v):
        v = v.astype(np.uint64)
        v = (v * 0x00010001) & 0xFF0000FF
        v = (v * 0x00000101) & 0x0F00F00F
        v = (v * 0x00000011) & 0xC30C30C3
        v = (v * 0x00000005) & 0x49249249
        return v

    m = (expand(x) | (expand(y) << 1) | (expand(z) << 2)).reshape(-1)
    grid = np.zeros(R ** 3, np.float16)
    grid[m] = dens.reshape(-1)
    root["snapshot"]["density_grid_size"] = R
    root["snapshot"]["density_grid_binary"] = grid.tobytes()
    raw = msgpack.packb(root, use_bin_type=True)
    with open(path, "wb") as fh:
        fh.write(gzip.compress(raw, compresslevel=1) if str(path).endswith(".ingp") else raw)
    return f, (table, mlp, occ)


@pytest.mark.parametrize("kw", [SMALL_NGP, NGP], ids=["small", "ngp_base_json"])
def test_snapshot_loads_and_renders_like_the_same_arrays(ctx, oracle, tmp_path, kw):
    path = tmp_path / "member.ingp"
    f, (table, mlp, occ) = snapshot_of(oracle, kw, util.SEED_A, path)
    d = ctx.load_ingp(0, path)
    assert (d.n_levels, d.n_features, d.log2_hashmap, d.base_res, d.occ_res) == tuple(kw[k] for k in ("n_levels", "n_features", "log2_hashmap", "base_res", "occ_res"))
    assert d.per_level_scale == np.float32(kw["per_level_scale"]) and d.density_bias == 0.0
    t2, m2, o2 = ctx.export_model(0, d)
    assert np.array_equal(t2, table) and np.array_equal(m2, mlp) and np.array_equal(o2, occ)
    # features on tiny-cuda-nn's level geometry: bit-exact against the oracle (finest level: 2048 cells per axis)
    rng = np.random.default_rng(2048)
    pos = rng.random((1200, 3), dtype=np.float32)
    pos[:4] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [0.999999, 0.000001, 0.5]]
    assert np.array_equal(ctx.debug_encode(0, pos), f.encode(pos))
    # same pixels as the same arrays through prv_model_load, and the oracle's rows
    ctx.load_model(1, api.L.FieldDesc(**kw), table, mlp, occ)
    w, h = (96, 64) if kw is SMALL_NGP else (200, 120)
    tms, scale, offset = planner.hemisphere_transforms(util.fibonacci_hemisphere(5), 0.3, 0.1, [1e-10] * 3)
    cams = ctx.cameras_from_matrices(tms, util.FOV_X, w, h, scale, offset)
    opts = api.render_opts(w, h, 128, 1, 1e-4)
    a, st = ctx.render(0, cams, None, opts)
    b, _ = ctx.render(1, cams, None, opts)
    assert bool((a == b).all()) and st.samples_evaluated > 1000
    ocam = oracle.cameras_from_transforms(tms, util.FOV_X, w, h, scale, offset)[2]
    rows = (h // 2 - 4, h // 2 + 4)
    want, _ = f.render(ocam, w, h, 128, 1, 1e-4, threads=8, rows=rows)
    util.assert_pixels_close(a[2].cpu().numpy()[rows[0]:rows[1]], want[rows[0]:rows[1]])
    assert want[rows[0]:rows[1], :, 3].max() > 0.2
    # ... and the way run.py:304 renders such a snapshot: the engine's own stepping rule (dt = sqrt(3)/1024, min_T 0.01),
    # march count exact, rows of a view against the oracle's restatement of that rule
    ngp = api.engine_render_opts(w, h, 0, 1, 1e-2)
    c, st_n = ctx.render(0, cams, [2], ngp)
    assert int(st_n.samples_live) == f.march_count(ocam, w, h, 0, step_mode=oracle.STEP_NGP)
    wants_n = [f.render(ocam, w, h, 0, 1, t, threads=8, rows=rows, step_mode=oracle.STEP_NGP)[0] for t in util.termination_variants(1e-2)]
    util.assert_pixels_close_any(c[0].cpu().numpy()[rows[0]:rows[1]], [x[rows[0]:rows[1]] for x in wants_n])
    assert wants_n[0][rows[0]:rows[1], :, 3].max() > 0.2
    # and out again: the writer's file is the reader's input
    out = tmp_path / "saved.msgpack"
    ctx.save_ingp(0, out)
    d3, t3, m3, o3 = planner.ingp_read(out)
    assert bytes(d3) == bytes(d) and np.array_equal(t3, table) and np.array_equal(m3, mlp) and np.array_equal(o3, occ)
    root = msgpack.unpackb(out.read_bytes(), raw=False)
    assert root["snapshot"]["n_params"] == 10240 + table.size and root["encoding"]["per_level_scale"] == float(np.float32(kw["per_level_scale"]))
    cams.close()


def test_testbed_load_snapshot_takes_ingp_files_and_refusals_reach_the_caller(ctx, oracle, tmp_path):
    """run.py:123-127: testbed.load_snapshot(args.load_snapshot) with the file instant-ngp wrote"""
    path = tmp_path / "base.ingp"
    f, _ = snapshot_of(oracle, SMALL_NGP, util.SEED_B, path)
    tb = api.Testbed(0)
    try:
        tb.load_snapshot(str(path))
        tb.fov_axis, tb.fov = 0, util.FOV_X * 180 / np.pi
        tb.scale, tb.offset = 5.0, [0.5, 0.5, 0.5]
        tms, _, _ = planner.hemisphere_transforms(util.fibonacci_hemisphere(3), 0.3, 0.1, [1e-10] * 3)
        tb.set_nerf_camera_matrix(np.asarray(tms[1])[:-1, :])
        tb.nerf.render_min_transmittance = 1e-4
        img = tb.render(48, 32, 1, True)
        assert img.shape == (32, 48, 4) and img[..., 3].max() > 0.2
        tb.save_snapshot(str(tmp_path / "again.ingp"))
        assert (tmp_path / "again.ingp").read_bytes()[:2] == b"\x1f\x8b"
    finally:
        tb.ctx.close()
    bad = tmp_path / "scaled.msgpack"
    root = msgpack.unpackb(gzip.decompress(path.read_bytes()), raw=False)
    root["snapshot"]["nerf"]["aabb_scale"] = 16
    bad.write_bytes(msgpack.packb(root, use_bin_type=True))
    with pytest.raises(api.PrvError) as e:
        ctx.load_ingp(2, bad)
    assert e.value.code == api.L.PRV_E_INVALID and "aabb_scale 16" in str(e.value)
    with pytest.raises(api.PrvError) as e:
        ctx.load_ingp(2, tmp_path / "nope.ingp")
    assert e.value.code == api.L.PRV_E_IO
    # a field without tiny-cuda-nn level geometry can not leave as a snapshot
    ctx.synthetic_model(3, api.field_desc(**util.SMALL), 1)
    with pytest.raises(api.PrvError, match="per_level_scale"):
        ctx.save_ingp(3, tmp_path / "x.ingp")


def test_planner_scores_with_members_given_as_ingp_snapshots(ctx, oracle, tmp_path):
    """configs[2]'s shape of input: the ensemble members are files instant-ngp left behind
    (<model_path>/<object>/member_<e>.ingp); the planner picks the views it picks with the same fields as .prvf"""
    from tests.test_gpu_planner import YAML

    exe = os.path.join(ROOT, "nerf_prv_amd", "prv_planner")
    chosen = {}
    for kind in ("ingp", "prvf"):
        pre = tmp_path / kind
        (pre / "models" / "objA").mkdir(parents=True)
        for e in range(2):
            p = pre / "models" / "objA" / f"member_{e}.ingp"
            snapshot_of(oracle, SMALL_NGP, 4000 + e, p)
            if kind == "prvf":
                ctx.load_ingp(0, p)
                ctx.save_model(0, pre / "models" / "objA" / f"member_{e}.prvf")
                p.unlink()
        cfg = pre / "cfg.yaml"
        cfg.write_text(YAML.format(pre=pre, vs=os.path.join(GOLD, "hemisphere"), method=2, model_source="pretrained_members: 1")
                       .replace("field_density_bias: 3.0", "field_density_bias: 0.0"))
        out = subprocess.run([exe, str(cfg)], input="21\nobjA\n-1\n", text=True, capture_output=True, timeout=300)
        assert out.returncode == 0, out.stdout + out.stderr
        chosen[kind] = [int(x) for x in [l for l in out.stdout.splitlines() if l.startswith("chosen_nbvs:")][-1].split(":")[1].split()]
    assert chosen["ingp"] == chosen["prvf"] and len(set(chosen["ingp"])) == 4
