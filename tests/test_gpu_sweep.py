"""Seeded random sweep of the render path against the CPU oracle: field shapes (levels x features, table sizes,
base / finest resolutions, occupancy resolutions incl. ones the coarse march test does not cover), image shapes
(odd, non-square, one pixel wide), sample counts, sub-sample counts, transmittance cut-offs, pinhole and lens
cameras, cameras inside and outside the unit cube.  Same bars as test_gpu_parity.py: grid features bit for bit,
pixels within 1e-3 (north_star), evaluated-sample counts equal up to the early-termination rounding."""
import numpy as np
import pytest

from nerf_prv_amd import api
from tests import util

pytestmark = pytest.mark.gpu

PIX_ATOL = 1e-3


def random_case(rng):
    F = int(rng.choice([2, 4]))
    L = 32 // F
    base = int(rng.integers(2, 17))
    kw = dict(n_levels=L, n_features=F, log2_hashmap=int(rng.integers(8, 17)), base_res=base,
              finest_res=int(rng.integers(base, 200)), occ_res=int(rng.choice([1, 5, 8, 16, 24, 30, 32, 64])),
              density_bias=float(rng.uniform(0.0, 4.0)), table_amp=float(rng.uniform(0.5, 4.0)))
    w, h = int(rng.integers(1, 49)), int(rng.integers(1, 33))
    return dict(field=kw, w=w, h=h, S=int(rng.integers(1, 129)), spp=int(rng.choice([1, 1, 2, 3, 4, 16])),
                min_T=float(rng.choice([0.0, 1e-4, 1e-2, 0.3])), lens=bool(rng.integers(0, 2)),
                n_views=int(rng.integers(1, 5)), predicted_size=float(rng.choice([0.1, 0.2, 0.45])),
                seed=int(rng.integers(1, 1 << 40)))


@pytest.mark.parametrize("case_id", range(40))
def test_random_render_case(ctx, oracle, case_id):
    rng = np.random.default_rng(0xC0FFEE + case_id)
    case = random_case(rng)
    d_o, d_p = oracle.desc(**case["field"]), api.field_desc(**case["field"])
    f = oracle.OracleField(d_o, seed=case["seed"])
    ctx.synthetic_model(2, d_p, case["seed"])
    t, m, o = ctx.export_model(2, d_p)
    to, mo, oo = f.params()
    assert np.array_equal(t, to) and np.array_equal(m, mo) and np.array_equal(o, oo), case
    # grid features at random positions (incl. the cube's faces and corners): fp16 bit patterns
    pos = rng.random((200, 3)).astype(np.float32)
    pos[:8] = np.array([[x, y, z] for x in (0.0, 1.0) for y in (0.0, 1.0) for z in (0.0, 1.0)], np.float32)
    pos[8:16, 0] = np.float32(1.0)
    assert np.array_equal(ctx.debug_encode(2, pos), f.encode(pos)), case
    # cameras: a few hemisphere views; a large predicted_size puts them inside the cube
    pts = util.fibonacci_hemisphere(case["n_views"] + 1)[1:] if case["n_views"] > 1 else util.fibonacci_hemisphere(1)
    tms, scale, offset = util.hemisphere_transforms(oracle, pts, predicted_size=case["predicted_size"])
    w, h = case["w"], case["h"]
    if case["lens"]:
        intr = {"fl_x": 915.6 * w / 1280, "fl_y": 913.3 * h / 720, "cx": 0.5055 * w, "cy": 0.5174 * h, "w": w, "h": h,
                "k1": float(rng.uniform(-0.2, 0.2)), "k2": float(rng.uniform(-0.2, 0.2)),
                "p1": float(rng.uniform(-0.01, 0.01)), "p2": float(rng.uniform(-0.01, 0.01))}
        cs = ctx.cameras_from_matrices_intr(tms, intr, scale, offset)
        ocams = oracle.cameras_from_dataset(tms, intr, scale, offset, w, h)
    else:
        cs = ctx.cameras_from_matrices(tms, util.FOV_X, w, h, scale, offset)
        ocams = oracle.cameras_from_transforms(tms, util.FOV_X, w, h, scale, offset)
    for v in range(len(ocams)):  # rays bit for bit, every sub-sample pattern used
        for k in {0, case["spp"] - 1}:
            o_d, d_d, t_d = ctx.debug_raygen(cs, v, w, h, k)
            o_o, d_o2, t_o = oracle.raygen(ocams[v], w, h, k)
            assert np.array_equal(o_d, o_o) and np.array_equal(d_d, d_o2) and np.array_equal(t_d, t_o), (case, v, k)
    img, st = ctx.render(2, cs, None, api.render_opts(w, h, case["S"], case["spp"], case["min_T"]))
    img = img.cpu().numpy()
    n_eval = 0
    for v, oc in enumerate(ocams):
        want, ne = f.render(oc, w, h, case["S"], case["spp"], case["min_T"])
        n_eval += ne
        assert np.abs(img[v] - want).max() <= PIX_ATOL, (case, v, float(np.abs(img[v] - want).max()))
        # north_star's RELATIVE 1e-3 (floor 1/255, tests/util.py) wherever no ray stops on a different sample than the
        # oracle's: with min_T up to 0.3 in this sweep one extra / missing sample legitimately moves a pixel by up to
        # min_T, so the relative bar applies to the cases that march to 1e-4
        if case["min_T"] <= 1e-4:
            util.assert_pixels_close(img[v], want)
    assert abs(int(st.samples_evaluated) - n_eval) <= max(2, n_eval // 10000), (case, int(st.samples_evaluated), n_eval)
    assert st.rays == len(ocams) * w * h * case["spp"]
    cs.close()
    f.close()


def rel_l2(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.mark.parametrize("case_id", range(10))
def test_random_training_case(ctx, oracle, case_id):
    """one training batch on a random field / image set / option set: the same rays and live samples bit for bit
    (sample counts equal), loss and gradients within 1e-3 (north_star) of the double-accumulating oracle"""
    rng = np.random.default_rng(0x7EA + case_id)
    F = int(rng.choice([2, 4]))
    base = int(rng.integers(2, 9))
    kw = dict(n_levels=32 // F, n_features=F, log2_hashmap=int(rng.integers(8, 13)), base_res=base,
              finest_res=int(rng.integers(base + 1, 64)), occ_res=int(rng.choice([4, 8, 16, 20])),
              density_bias=float(rng.uniform(0.0, 2.0)), table_amp=float(rng.uniform(0.05, 1.0)))
    w, h, n_views = int(rng.integers(3, 33)), int(rng.integers(3, 25)), int(rng.integers(1, 7))
    intr = {"fl_x": 0.8 * w, "fl_y": 0.78 * w, "cx": 0.51 * w, "cy": 0.48 * h, "w": w, "h": h,
            "k1": float(rng.uniform(-0.1, 0.1)), "k2": float(rng.uniform(-0.1, 0.1)),
            "p1": float(rng.uniform(-0.005, 0.005)), "p2": float(rng.uniform(-0.005, 0.005))}
    pts = util.fibonacci_hemisphere(n_views)
    tms, scale, offset = util.hemisphere_transforms(oracle, pts, predicted_size=float(rng.choice([0.1, 0.3])))
    ocams = oracle.cameras_from_dataset(tms, intr, scale, offset)
    cams = ctx.cameras_from_matrices_intr(tms, intr, scale, offset)
    imgs = rng.integers(0, 256, (n_views, h, w, 4), dtype=np.uint8)
    imgs[..., 3] = np.where(rng.random((n_views, h, w)) < 0.3, 0, np.where(rng.random((n_views, h, w)) < 0.5, 255, imgs[..., 3]))
    f = oracle.OracleField(oracle.desc(**kw), seed=int(rng.integers(1, 1 << 40)))
    t, m, o = f.params()
    if rng.random() < 0.5:
        o = np.full_like(o, 0xFFFFFFFF)  # a fresh network: every cell occupied
    f = oracle.OracleField(f.desc, params=(t, m, o))
    ctx.load_model(3, api.field_desc(**kw), t, m, o)
    opts = dict(n_rays=int(rng.integers(1, 700)), n_samples=int(rng.integers(1, 129)), occ_every=0,
                random_bg=int(rng.integers(0, 2)), min_T=float(rng.choice([0.0, 1e-4, 1e-2])),
                seed=int(rng.integers(1, 1 << 40)), target_samples=int(rng.choice([0, 1 << 18])))
    otr = oracle.OracleTrainer(f, oracle.train_opts(**opts), ocams, imgs)
    gtr = api.Trainer(ctx, 3, cams, ctx.torch.from_numpy(imgs), api.train_opts(**opts))
    want_loss, want_tg, want_mg = otr.gradients()
    loss, tg, mg = gtr.gradients()
    assert gtr.info()["samples_last"] == otr.samples_last, (kw, opts)
    assert gtr.info()["active_rays"] == otr.active_rays
    assert loss == pytest.approx(want_loss, rel=1e-3, abs=1e-7), (kw, opts)
    if np.abs(want_mg).max() > 1e-9:
        assert rel_l2(mg, want_mg) < 1e-3 and rel_l2(tg, want_tg) < 1e-3, (kw, opts, rel_l2(mg, want_mg), rel_l2(tg, want_tg))
    gtr.close()
    cams.close()


@pytest.mark.parametrize("case_id", range(12))
def test_random_score_case(ctx, oracle, case_id):
    """ensemble scores from random bytes: any ensemble size, image shape and view count.  The device adds the
    per-pixel addends in the reference loop's own order (one sequential double sum per view), so EnsembleRGBDensity
    scores are BIT-identical to the oracle's; EnsembleRGB adds logarithms, where the device's log and libm's may
    differ in the last bit of an addend -> 1e-13.  Rankings and arg-max identical, mathematically tied views
    (ensembles that agree on almost every pixel) included."""
    rng = np.random.default_rng(0x5C0 + case_id)
    E, n_views = int(rng.integers(2, 9)), int(rng.integers(1, 20))
    h, w = int(rng.integers(1, 70)), int(rng.integers(1, 90))
    method = int(rng.choice([2, 3]))
    imgs = [rng.integers(0, 256, (n_views, h, w, 4), dtype=np.uint8) for _ in range(E)]
    if rng.random() < 0.5:  # mostly agreeing members: variances near the 1e-10 cut-off and exact zeros
        for e in range(1, E):
            imgs[e] = imgs[0].copy()
            flip = rng.random(imgs[e].shape) < 0.02
            imgs[e][flip] ^= 1
    dev = [ctx.torch.from_numpy(im).to(ctx.device) for im in imgs]
    rec = ctx.score_ensemble_images(method, dev)
    fn = oracle.score_ensemble_rgb if method == 2 else oracle.score_ensemble_rgbdensity
    want = np.array([fn([im[v] for im in imgs]) for v in range(n_views)])
    if method == 3:
        assert np.array_equal(rec["score"], want)
    else:
        np.testing.assert_allclose(rec["score"], want, rtol=1e-13, atol=1e-13)
        print("bit-identical EnsembleRGB scores:", int((rec["score"] == want).sum()), "of", n_views)
    ids = np.arange(n_views, dtype=np.int32)
    assert list(ctx.rank(rec, ids)) == list(oracle.rank(want, ids))
    assert ctx.argmax(rec, ids) == oracle.argmax(want, ids)


def _random_cameras(ctx, oracle, rng, w, h, n_views, lens):
    pts = util.fibonacci_hemisphere(n_views + 1)[1:] if n_views > 1 else util.fibonacci_hemisphere(1)
    tms, scale, offset = util.hemisphere_transforms(oracle, pts, predicted_size=float(rng.choice([0.1, 0.2, 0.45])))
    if lens:
        intr = {"fl_x": 0.75 * w, "fl_y": 0.72 * w, "cx": 0.49 * w, "cy": 0.53 * h, "w": w, "h": h,
                "k1": float(rng.uniform(-0.2, 0.2)), "k2": float(rng.uniform(-0.2, 0.2)),
                "p1": float(rng.uniform(-0.01, 0.01)), "p2": float(rng.uniform(-0.01, 0.01))}
        return ctx.cameras_from_matrices_intr(tms, intr, scale, offset), oracle.cameras_from_dataset(tms, intr, scale, offset, w, h), (tms, scale, offset)
    return (ctx.cameras_from_matrices(tms, util.FOV_X, w, h, scale, offset),
            oracle.cameras_from_transforms(tms, util.FOV_X, w, h, scale, offset), (tms, scale, offset))


@pytest.mark.parametrize("case_id", range(12))
def test_random_first_hit_case(ctx, oracle, case_id):
    """the occupancy ray-caster (a13) on random grids, cameras and ranges: integer cell ids, exact"""
    rng = np.random.default_rng(0xF1 + case_id)
    F = int(rng.choice([2, 4]))
    kw = dict(n_levels=32 // F, n_features=F, log2_hashmap=8, base_res=2, finest_res=8,
              occ_res=int(rng.choice([1, 3, 8, 17, 32, 64, 100])))
    f = oracle.OracleField(oracle.desc(**kw), seed=5)
    t, m, o = f.params()
    o = rng.integers(0, 1 << 32, o.shape, dtype=np.uint32) & rng.integers(0, 1 << 32, o.shape, dtype=np.uint32)
    if rng.random() < 0.5:
        o &= rng.integers(0, 1 << 32, o.shape, dtype=np.uint32)  # sparser
    f = oracle.OracleField(f.desc, params=(t, m, o))
    ctx.load_model(2, api.field_desc(**kw), t, m, o)
    w, h = int(rng.integers(1, 40)), int(rng.integers(1, 30))
    cs, ocams, _ = _random_cameras(ctx, oracle, rng, w, h, int(rng.integers(1, 4)), bool(rng.integers(0, 2)))
    max_range = float(rng.choice([1e30, 1.0, 1.4, 0.3]))
    got = ctx.first_hit(2, cs, None, w, h, max_range=max_range).cpu().numpy()
    for v, oc in enumerate(ocams):
        assert np.array_equal(got[v], oracle.first_hit_image(f, oc, w, h, max_range=max_range)), (kw, v, max_range)
    cs.close()


@pytest.mark.parametrize("case_id", range(8))
def test_random_splat_case(ctx, oracle, case_id):
    """ground-truth splats (8f-4) of random clouds: bytes identical, whatever the order of the depth atomics"""
    rng = np.random.default_rng(0x5B1A7 + case_id)
    n = int(rng.integers(0, 30000))
    xyz = (rng.normal(size=(n, 3)) * float(rng.choice([0.01, 0.03, 0.08]))).astype(np.float32)
    if n > 10:
        xyz[: n // 4] = xyz[n // 4: 2 * (n // 4)]  # coincident points: equal depths, the colour breaks the tie
    rgb = rng.integers(0, 256, size=(n, 3), dtype=np.uint8)
    w, h = int(rng.integers(1, 120)), int(rng.integers(1, 70))
    cs, ocams, (tms, scale, offset) = _random_cameras(ctx, oracle, rng, w, h, int(rng.integers(1, 4)), True)
    size, flip = int(rng.integers(1, 9)), bool(rng.integers(0, 2))
    got = ctx.splat_points(xyz, rgb, scale, offset, cs, None, w, h, point_size=size, flip180=flip).cpu().numpy()
    for v, oc in enumerate(ocams):
        want = oracle.splat_points(xyz, rgb, scale, offset, oc, w, h, point_size=size, flip180=flip)
        assert np.array_equal(got[v], want), (n, w, h, size, flip, v)
    cs.close()


@pytest.mark.parametrize("case_id", range(8))
def test_random_evaluation_case(ctx, oracle, case_id):
    """PSNR / coverage / SSIM of random image pairs (a10) against the oracle's double-precision restatement"""
    rng = np.random.default_rng(0xE7A1 + case_id)
    n, h, w = int(rng.integers(1, 5)), int(rng.integers(5, 60)), int(rng.integers(5, 80))
    a = rng.random((n, h, w, 4)).astype(np.float32)
    a[..., :3] *= a[..., 3:4]  # premultiplied
    b = np.clip(a + rng.normal(scale=float(rng.choice([0.002, 0.05, 0.3])), size=a.shape), 0, 1).astype(np.float32)
    bg = tuple(float(x) for x in rng.choice([0.0, 1.0, 0.5], 3)) + (1.0,)
    ta, tb = ctx.torch.from_numpy(a).to(ctx.device), ctx.torch.from_numpy(b).to(ctx.device)
    rec = ctx.score_psnr_images(ta, tb, bg)
    ps, ss = ctx.evaluate_images(ta, tb, bg)
    for v in range(n):
        want_s, want_p, want_c = oracle.score_view(a[v], b[v], bg)
        assert rec["score"][v] == pytest.approx(want_s, rel=1e-6)
        assert rec["psnr"][v] == pytest.approx(want_p, rel=1e-5) and ps[v] == pytest.approx(want_p, rel=1e-5)
        assert rec["coverage"][v] == pytest.approx(want_c, rel=1e-5)
        assert ss[v] == pytest.approx(oracle.ssim(a[v], b[v], bg), rel=1e-3, abs=1e-5)


def test_view_batches_do_not_change_anything(ctx, oracle, monkeypatch):
    """a render call deals its views to the ray queue in batches when the queue would outgrow its budget (the
    128-views-per-GPU configuration does); a context with a 1 MiB budget renders 17 views in several batches and
    must give the same bytes, floats, scores and counts as the one-batch context -- with view subsets, repeats
    and sub-samples"""
    monkeypatch.setenv("PRV_QUEUE_MB", "1")
    small = api.Context(0)
    monkeypatch.delenv("PRV_QUEUE_MB")
    try:
        d_p = api.field_desc(**util.SMALL)
        for c in (ctx, small):
            c.synthetic_model(0, d_p, util.SEED_A)
            c.synthetic_model(1, d_p, util.SEED_B)
        pts = util.fibonacci_hemisphere(17)
        tms, scale, offset = util.hemisphere_transforms(oracle, pts)
        w, h = 48, 40  # 48*40*96 B = 184 KB per view and sub-sample: 5 views per batch at spp 1, 1 at spp 4
        ids = np.array([3, 16, 0, 0, 7, 8, 9, 1, 2, 15, 14, 13, 3, 5, 6, 4, 10, 11, 12], np.int32)
        for spp in (1, 4):
            opts = api.render_opts(w, h, 64, spp, 1e-4, background=(0.2, 0.4, 0.6, 1.0))
            outs = []
            for c in (ctx, small):
                cs = c.cameras_from_matrices(tms, util.FOV_X, w, h, scale, offset)
                img, st = c.render(0, cs, ids, opts)
                u8, _ = c.render_rgba8(0, cs, ids, opts)
                gt, _ = c.render(1, cs, ids, opts, want_stats=False)
                rec, _ = c.score_views(api.L.SCORE_PSNR_COVERAGE, [0], cs, ids, opts, gt=gt)
                ens, _ = c.score_views(api.L.SCORE_ENSEMBLE_RGB_DENSITY, [0, 1], cs, ids, opts)
                outs.append((img.cpu().numpy(), u8.cpu().numpy(), int(st.samples_evaluated), rec.copy(), ens.copy()))
                cs.close()
            a, b = outs
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] > 0
            assert a[3].tobytes() == b[3].tobytes() and a[4].tobytes() == b[4].tobytes()
            assert np.array_equal(a[0][2], a[0][3]) and np.array_equal(a[0][0], a[0][12])  # repeated ids
    finally:
        small.close()


@pytest.mark.parametrize("env", [{"PRV_QUEUE_SEGMENTS": "1"}, {"PRV_QUEUE_SEGMENTS": "3"}, {"PRV_NO_PAIR": "1"},
                                 {"PRV_SPATIAL_REGIONS": "0"}, {"PRV_SPATIAL_REGIONS": "2"}, {"PRV_SPATIAL_REGIONS": "0", "PRV_QUEUE_SEGMENTS": "2"},
                                 {"PRV_BLOCKS_PER_CU": "1"}, {"PRV_BLOCKS_PER_CU": "6"},
                                 {"PRV_MERGE_MAX": "0"}, {"PRV_MERGE_MAX": "6", "PRV_POOL": "0"}, {"PRV_MERGE_MAX": "31", "PRV_BLOCKS_PER_CU": "2"},
                                 {"PRV_MERGE_MAX": "12", "PRV_POOL": "1"}, {"PRV_MERGE_MAX": "31", "PRV_POOL": "1", "PRV_BLOCKS_PER_CU": "1"}])
@pytest.mark.parametrize("which", ["F4", "F2"])
@pytest.mark.parametrize("step_mode", ["fixed", "ngp"])
def test_placement_and_layout_switches_change_speed_only(ctx, oracle, monkeypatch, env, which, step_mode):
    """the tuning switches of the render path (queue segments per XCD, which region a wave's rays go to -- the octant of the
    middle of their live span, round 6's default, or the block index --, the generic gather for every level, resident
    blocks per CU, tail merge and tail pool: rays change lanes mid-flight) decide where and in which order rays are
    rendered -- never the arithmetic: images and counts are bit-identical to the defaults, in both stepping modes"""
    kw = util.SMALL if which == "F4" else util.SMALL_F2
    d_p = api.field_desc(**kw)
    pts = util.fibonacci_hemisphere(5)
    tms, scale, offset = util.hemisphere_transforms(oracle, pts)
    w, h = 56, 44
    opts = api.render_opts(w, h, 96, 1, 1e-4, step_mode=api.L.STEP_NGP if step_mode == "ngp" else api.L.STEP_FIXED_S)
    ctx.synthetic_model(2, d_p, util.SEED_A)
    cs = ctx.cameras_from_matrices(tms, util.FOV_X, w, h, scale, offset)
    want, st0 = ctx.render(2, cs, None, opts)
    feat0 = ctx.debug_encode(2, np.random.default_rng(1).random((300, 3)).astype(np.float32))
    cs.close()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    other = api.Context(0)
    for k in env:
        monkeypatch.delenv(k)
    try:
        other.synthetic_model(2, d_p, util.SEED_A)
        cs = other.cameras_from_matrices(tms, util.FOV_X, w, h, scale, offset)
        got, st1 = other.render(2, cs, None, opts)
        assert np.array_equal(got.cpu().numpy(), want.cpu().numpy()), env
        assert st1.samples_evaluated == st0.samples_evaluated > 0
        assert np.array_equal(other.debug_encode(2, np.random.default_rng(1).random((300, 3)).astype(np.float32)), feat0)
        cs.close()
    finally:
        other.close()


@pytest.mark.parametrize("env", [{"PRV_TRAIN_GRAPH": "0"}, {"PRV_TRAIN_FAST_FWD": "0"},
                                 {"PRV_TRAIN_GRAPH": "0", "PRV_TRAIN_FAST_FWD": "0"}, {"PRV_TRAIN_KEEP_ACT": "0"},
                                 {"PRV_TRAIN_ACT_CAP": "256"}, {"PRV_TRAIN_ACT_CAP": "256", "PRV_TRAIN_GRAPH": "0"},
                                 {"PRV_TRAIN_REG_CHAIN": "0"}, {"PRV_TRAIN_REG_CHAIN": "0", "PRV_TRAIN_ACT_CAP": "256"},
                                 {"PRV_TRAIN_OWN_QUEUE": "0"}, {"PRV_TRAIN_OWN_QUEUE": "0", "PRV_TRAIN_GRAPH": "0"},
                                 {"patch": "4x2"}, {"patch": "4x2", "PRV_TRAIN_FAST_FWD": "0"}, {"patch": "2x2", "PRV_TRAIN_KEEP_ACT": "0"},
                                 {"patch": "4x4", "PRV_TRAIN_ACT_CAP": "256", "PRV_TRAIN_GRAPH": "0"}, {"patch": "2x2", "PRV_TRAIN_REG_CHAIN": "0"}])
def test_trainer_switches_hold_the_same_bars(ctx, oracle, monkeypatch, env):
    """the trainer's switches -- plain launches instead of the captured step graph, f32 tile forward instead of the
    f16-MFMA forward, a backward pass that recomputes the forward activations instead of reading the ones the forward
    pass kept, and a kept-activation buffer smaller than the batch (256 of > 500 samples: the backward pass is then two
    launches, kept tiles + recomputed tiles), the LDS / f32-MFMA gradient chain instead of the register-resident bf16-split
    one -- against the oracle on the same batch and over a few optimiser steps: same
    ray batch, loss and gradients within 1e-3, steps tracking the oracle like the default configuration does"""
    kw = dict(n_levels=8, n_features=4, log2_hashmap=10, base_res=4, finest_res=24, occ_res=16, density_bias=1.0, table_amp=0.5)
    intr = {"fl_x": 20.0, "fl_y": 19.5, "cx": 12.3, "cy": 7.8, "w": 24, "h": 16, "k1": 0.05, "k2": -0.02, "p1": 0.001, "p2": -0.002}
    gt = oracle.OracleField(oracle.desc(**dict(kw, density_bias=3.0, table_amp=2.0)), seed=util.SEED_B)
    tms, scale, offset = util.hemisphere_transforms(oracle, util.fibonacci_hemisphere(6))
    ocams = oracle.cameras_from_dataset(tms, intr, scale, offset)
    imgs = np.stack([oracle.quantize_rgba8(gt.render(c, 24, 16, 32, 1, 1e-4)[0], (0, 0, 0, 0)) for c in ocams])
    cams = ctx.cameras_from_matrices_intr(tms, intr, scale, offset)
    f = oracle.OracleField(oracle.desc(**kw), seed=util.SEED_A)
    t, m, o = f.params()
    o = np.full_like(o, 0xFFFFFFFF)
    f = oracle.OracleField(f.desc, params=(t, m, o))
    ctx.load_model(3, api.field_desc(**kw), t, m, o)
    opts = dict(n_rays=160, n_samples=24, occ_every=4, occ_sigma_thresh=0.3)
    env = dict(env)
    if "patch" in env:  # the same switches with the step's rays drawn as patches (the list order and slot_of change under every path)
        pw, ph = (int(x) for x in env.pop("patch").split("x"))
        opts.update(patch_w=pw, patch_h=ph)
    otr = oracle.OracleTrainer(f, oracle.train_opts(**opts), ocams, imgs)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    gtr = api.Trainer(ctx, 3, cams, ctx.torch.from_numpy(imgs), api.train_opts(**opts))  # reads the switches
    for k in env:
        monkeypatch.delenv(k)
    want_loss, want_tg, want_mg = otr.gradients()
    loss, tg, mg = gtr.gradients()
    assert gtr.info()["samples_last"] == otr.samples_last > 500
    assert loss == pytest.approx(want_loss, rel=1e-3)
    assert rel_l2(mg, want_mg) < 1e-3 and rel_l2(tg, want_tg) < 1e-3
    want = np.array([otr.step() for _ in range(6)])
    np.testing.assert_allclose(gtr.steps(6), want, rtol=2e-3)
    assert gtr.info()["active_rays"] == otr.active_rays
    gtr.close()
    cams.close()


def test_queue_regions_by_octant_leave_the_ensemble_round_unchanged(ctx, oracle, monkeypatch):
    """round 6: a wave's records go to the queue region of the OCTANT its first live ray crosses (an XCD drains a region: its L2
    then holds one part of the table), in sub-regions with a counter each, moving on to the next when one is full.  The
    reference-shaped scoring round -- five members marched in one launch, 16 sub-samples per pixel, the engine's stepping --
    must give the same records and the same member images as the block-index regions of rounds 1-5, byte for byte."""
    d_p = api.field_desc(**util.SMALL)
    pts = util.fibonacci_hemisphere(12)
    tms, scale, offset = util.hemisphere_transforms(oracle, pts)
    opts = api.engine_render_opts(80, 45, 0, 16, 0.01, background=(0, 0, 0, 1))

    def round_of(c):
        for e in range(5):
            c.synthetic_model(e, d_p, util.SEED_A + e)
        cs = c.cameras_from_matrices(tms, util.FOV_X, 80, 45, scale, offset)
        rec, st = c.score_views(api.L.SCORE_ENSEMBLE_RGB_DENSITY, [0, 1, 2, 3, 4], cs, None, opts, want_stats=True)
        u8 = [c.render_rgba8(e, cs, None, opts)[0].cpu().numpy() for e in (0, 4)]
        cs.close()
        return rec.tobytes(), u8, int(st.samples_evaluated), int(st.samples_live)

    want = round_of(ctx)
    for v in ("0", "2"):
        monkeypatch.setenv("PRV_SPATIAL_REGIONS", v)
        other = api.Context(0)
        monkeypatch.delenv("PRV_SPATIAL_REGIONS")
        try:
            got = round_of(other)
        finally:
            other.close()
        assert got[0] == want[0] and got[2:] == want[2:] and all(np.array_equal(a, b) for a, b in zip(got[1], want[1])), v
