"""Full-size (800x800, 128 samples/ray, 256^3 field) checks on the GPU through size-independent
properties, plus bounded direct comparisons against the oracle (a few rows of one view)."""
import os

import numpy as np
import pytest

from nerf_prv_amd import api, planner
from tests import util

pytestmark = pytest.mark.gpu

W = H = 800
S = 128


@pytest.fixture(scope="module")
def scene(ctx):
    desc = api.L.FieldDesc(**api.FIELD_256)
    ctx.synthetic_model(0, desc, util.SEED_A)
    ctx.synthetic_model(1, desc, util.SEED_B)
    pts = planner.hemisphere_generate(8)
    tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
    cams = ctx.cameras_from_matrices(tms, util.FOV_X, W, H, scale, offset)
    return desc, cams, (tms, scale, offset)


def test_rows_of_a_full_size_view_match_the_oracle(ctx, oracle, scene):
    desc, cams, (tms, scale, offset) = scene
    opts = api.render_opts(W, H, S, 1, 1e-4)
    img, st = ctx.render(0, cams, [3], opts)
    img = img[0].cpu().numpy()
    f = oracle.OracleField(oracle.desc(), seed=util.SEED_A)
    ocam = oracle.cameras_from_transforms(tms, util.FOV_X, W, H, scale, offset)[3]
    rows = (396, 404)
    want, _ = f.render(ocam, W, H, S, 1, 1e-4, threads=8, rows=rows)
    util.assert_pixels_close(img[rows[0]:rows[1]], want[rows[0]:rows[1]])  # north_star: 1e-3 relative (floor: tests/util.py)
    assert want[rows[0]:rows[1], :, 3].max() > 0.5  # the rows do cross the object


def test_full_size_determinism_view_independence_and_bounds(ctx, scene):
    desc, cams, _ = scene
    opts = api.render_opts(W, H, S, 1, 1e-4)
    a, st = ctx.render(0, cams, None, opts)
    b, _ = ctx.render(0, cams, [5, 2], opts)
    c, _ = ctx.render(0, cams, None, opts)
    assert bool((a == c).all())  # bit-reproducible although the ray queue order is not
    assert bool((b[0] == a[5]).all()) and bool((b[1] == a[2]).all())  # a view does not depend on its batch
    x = a.cpu().numpy()
    assert np.isfinite(x).all() and x.min() >= 0.0
    assert x[..., 3].max() <= 1.0 and (x[..., :3] <= x[..., 3:4] + 1e-6).all()  # premultiplied colour <= alpha
    assert st.rays == 8 * W * H and st.samples_nominal == st.rays * S
    assert 0 < st.samples_evaluated < st.samples_nominal // 10  # occupancy skip + early termination
    # corner pixels look past the object: exactly transparent
    assert float(a[:, 0, 0].abs().max()) == 0.0


def test_render_launch_reports_the_shader_clock_it_ran_at(ctx, scene):
    """prv_debug_render_clock: the launch stamps the shader cycle counter against the constant-rate reference counter;
    bench.py prices the per-clock roofline peaks with it.  An MI355X runs between its idle floor and 2.4 GHz."""
    desc, cams, _ = scene
    opts = api.render_opts(W, H, S, 1, 1e-4)
    for view_ids in (None, [1]):
        ctx.render(0, cams, view_ids, opts)
        ghz = ctx.render_clock_ghz()
        assert 0.3 < ghz <= 2.5, ghz


def test_full_size_scores_are_consistent_and_ranking_matches_host_recompute(ctx, oracle, scene):
    desc, cams, _ = scene
    opts = api.render_opts(W, H, S, 1, 1e-4)
    gt, _ = ctx.render(1, cams, None, opts)
    img, _ = ctx.render(0, cams, None, opts)
    rec, _ = ctx.score_views(api.L.SCORE_PSNR_COVERAGE, [0], cams, None, opts, gt=gt)
    rec2 = ctx.score_psnr_images(img, gt)
    assert rec.tobytes() == rec2.tobytes()  # fused round == render then score
    # checker: the oracle's PSNR recipe on the GPU's own images (bytes identical in, 1e-6 out)
    x, g = img.cpu().numpy(), gt.cpu().numpy()
    want = np.array([oracle.score_view(x[v], g[v]) for v in range(len(x))])  # (score, psnr, coverage)
    np.testing.assert_allclose(rec["score"], want[:, 0], rtol=1e-6)
    np.testing.assert_allclose(rec["psnr"], want[:, 1], rtol=1e-5)
    np.testing.assert_allclose(rec["coverage"], want[:, 2], rtol=1e-6)
    ids = np.arange(len(x))
    assert np.array_equal(ctx.rank(rec, ids), oracle.rank(want[:, 0], ids))
    # identical images -> mse 0 -> psnr +inf, score -inf, ranked last (run.py:263 has no guard either)
    same = ctx.score_psnr_images(img, img)
    assert np.isinf(same["psnr"]).all() and (same["score"] == -np.inf).all()


def test_ensemble_round_at_reference_candidate_size(ctx, oracle, scene):
    """80x45 candidates, spp 16, E = 2 / 5 (main.cpp:1805-1806, run.py:48, Share_Data.hpp:505-510)"""
    desc, cams, (tms, scale, offset) = scene
    small = ctx.cameras_from_matrices(tms, util.FOV_X, 80, 45, scale, offset)
    for e in range(5):
        ctx.synthetic_model(2 + e, desc, 4000 + e)
    opts = api.render_opts(80, 45, S, 16, 0.01, background=(0, 0, 0, 1))
    for method, E in ((2, 2), (3, 5)):
        rec, st = ctx.score_views(method, list(range(2, 2 + E)), small, None, opts, want_stats=True)
        assert st.rays == 8 * 80 * 45 * 16 * E
        imgs = [ctx.render_rgba8(2 + e, small, None, opts)[0].cpu().numpy() for e in range(E)]
        fn = oracle.score_ensemble_rgb if method == 2 else oracle.score_ensemble_rgbdensity
        want = np.array([fn([im[v] for im in imgs]) for v in range(8)])
        if method == 3:  # same bytes in, the reference loop's own summation order on both sides -> same bits out
            assert np.array_equal(rec["score"], want)
        else:  # sums of logs: the last bit of an addend may differ between the two log implementations
            np.testing.assert_allclose(rec["score"], want, rtol=1e-13)
        assert (imgs[0][..., 3] == 255).all()  # opaque background: alpha carries nothing


def test_reference_round_with_the_march_in_one_launch_scores_what_the_member_renders_score(ctx, oracle, scene):
    """the reference's round at ITS sizes under the engine's rule -- 80x45, 16 sub-samples, five members of the full-size
    256^3 field, 144 candidate views -- goes through ONE march launch for the five members (march_multi_kernel).  The
    records must be the scores of the images each member renders ON ITS OWN (prv_render_rgba8: the one-member march and
    render), bit for bit (method 3: same bytes in, the reference loop's summation order on both sides), and the round's
    march count the sum of the members' own"""
    desc, cams, _ = scene
    pts = planner.hemisphere_read(os.path.join(os.path.dirname(__file__), "golden", "hemisphere", "144.txt"), 144)
    tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
    small = ctx.cameras_from_matrices(tms, util.FOV_X, 80, 45, scale, offset)
    for e in range(5):
        ctx.synthetic_model(2 + e, desc, 4100 + e)
    opts = api.engine_render_opts(80, 45, 0, 16, 0.01, background=(0, 0, 0, 1))
    rec, st = ctx.score_views(3, list(range(2, 7)), small, None, opts, want_stats=True)
    imgs, live = [], 0
    for e in range(5):
        im, se = ctx.render_rgba8(2 + e, small, None, opts, want_stats=True)
        imgs.append(im.cpu().numpy())
        live += int(se.samples_live)
    want = np.array([oracle.score_ensemble_rgbdensity([im[v] for im in imgs]) for v in range(144)])
    assert np.array_equal(rec["score"], want)
    assert int(st.samples_live) == live > 0 and st.rays == 144 * 80 * 45 * 16 * 5
    small.close()


@pytest.mark.parametrize("rule", ["fixed", "ngp"])
def test_reference_size_candidates_match_the_oracle(ctx, oracle, scene, rule):
    """what the ensemble scores are MADE of: the 80x45, 16-sub-sample candidate renders of run.py:304 (main.cpp:1796-1806)
    on the full-size field against the oracle's own render, whole images, under both stepping rules -- the float image to
    1e-3 (termination variants: the engine's min_T = 0.01) and the bytes the scores read to one code"""
    desc, cams, (tms, scale, offset) = scene
    w, h, spp, min_T = 80, 45, 16, 1e-2
    small = ctx.cameras_from_matrices(tms, util.FOV_X, w, h, scale, offset)
    ocams = oracle.cameras_from_transforms(tms, util.FOV_X, w, h, scale, offset)
    f = oracle.OracleField(oracle.desc(), seed=util.SEED_A)
    bg = (0, 0, 0, 1)
    opts = api.engine_render_opts(w, h, 0 if rule == "ngp" else S, spp, min_T, background=bg)
    mode = oracle.STEP_NGP if rule == "ngp" else oracle.STEP_FIXED_S
    img, st = ctx.render(0, small, None, opts)
    u8, _ = ctx.render_rgba8(0, small, None, opts)
    assert st.rays == 8 * w * h * spp
    n_codes_off = 0
    for v in range(8):
        wants = [f.render(ocams[v], w, h, 0 if rule == "ngp" else S, spp, t, threads=8, step_mode=mode)[0] for t in util.termination_variants(min_T)]
        util.assert_pixels_close_any(img[v].cpu().numpy(), wants)
        want8 = oracle.quantize_rgba8(wants[0], bg)
        d = np.abs(u8[v].cpu().numpy().astype(np.int32) - want8.astype(np.int32))
        assert d.max() <= 1  # a value on a rounding boundary may land on either code
        n_codes_off += int((d != 0).sum())
    assert n_codes_off <= 8 * w * h * 4 // 2000  # and hardly any does (measured: 1 and 0 of 115,200 bytes)
    assert wants[0][..., 3].max() > 0.9
    f.close()
    small.close()


def test_ensemble_round_end_to_end_against_the_oracle_alone(ctx, oracle, scene):
    """the reference's scoring round (main.cpp:2045-2160) with nothing of the GPU's on the checking side: five members,
    eight candidates at 80x45 with 16 sub-samples; the oracle renders every member's candidates, quantises them as
    write_image does and scores them with the reference's loops -- against prv_score_views of the same members.  A byte
    on a rounding boundary may differ by one code between the two renders, so the scores agree closely, not bitwise;
    the ranking must be the same."""
    desc, cams, (tms, scale, offset) = scene
    w, h, spp, min_T, E = 80, 45, 16, 1e-2, 5
    small = ctx.cameras_from_matrices(tms, util.FOV_X, w, h, scale, offset)
    ocams = oracle.cameras_from_transforms(tms, util.FOV_X, w, h, scale, offset)
    bg = (0, 0, 0, 1)
    opts = api.render_opts(w, h, S, spp, min_T, background=bg)
    imgs = []
    for e in range(E):
        ctx.synthetic_model(2 + e, desc, 4000 + e)
        f = oracle.OracleField(oracle.desc(), seed=4000 + e)
        imgs.append([oracle.quantize_rgba8(f.render(oc, w, h, S, spp, min_T, threads=8)[0], bg) for oc in ocams])
        f.close()
    ids = np.arange(8, dtype=np.int32)
    for method, n, fn in ((2, 2, oracle.score_ensemble_rgb), (3, 5, oracle.score_ensemble_rgbdensity)):
        want = np.array([fn([imgs[e][v] for e in range(n)]) for v in range(8)])
        rec, _ = ctx.score_views(method, list(range(2, 2 + n)), small, None, opts)
        np.testing.assert_allclose(rec["score"], want, rtol=1e-3)  # measured: 1.5e-4 (sums of logs), 1e-5 (variance + density)
        got_order, want_order = ctx.rank(rec, ids), oracle.rank(want, ids)
        for a, b in zip(got_order, want_order):
            assert a == b or abs(want[a] - want[b]) <= 4e-3 * abs(want[a])
        assert np.ptp(want) > 0.05 * abs(want).max()  # the candidates do differ
    small.close()


def test_evaluation_path_psnr_ssim_and_metrics_file(ctx, oracle, scene, tmp_path):
    """run.py:226-277 on the device: spp 8 snapped to pixel centres == spp 1, min_T 1e-4, black opaque
    background; mean PSNR / SSIM over the test views; the metrics file other tools read"""
    desc, cams, (tms, scale, offset) = scene
    w, h = 200, 112  # a 16:9 test size; the oracle's SSIM is a scalar loop
    small = ctx.cameras_from_matrices(tms, util.FOV_X, w, h, scale, offset)
    bg = (0.0, 0.0, 0.0, 1.0)  # run.py:226
    opts = api.render_opts(w, h, S, 1, 1e-4, background=bg)
    gt, _ = ctx.render(1, small, None, opts)
    img, _ = ctx.render(0, small, None, opts)
    ps, ss = ctx.evaluate_images(img, gt, bg)
    x, g = img.cpu().numpy(), gt.cpu().numpy()
    for v in range(len(x)):
        want_p, _ = oracle.score_psnr_coverage(x[v], g[v], bg)
        np.testing.assert_allclose(ps[v], want_p, rtol=1e-6)
        np.testing.assert_allclose(ss[v], oracle.ssim(x[v], g[v], bg), rtol=1e-5)
    same_p, same_s = ctx.evaluate_images(img, img, bg)
    np.testing.assert_allclose(same_s, 1.0, atol=1e-6)
    mp, ms = ctx.evaluate(0, small, None, opts, gt)
    assert abs(mp - ps.mean()) < 1e-9 and abs(ms - ss.mean()) < 1e-12
    path = tmp_path / "64.txt"
    planner.write_metrics(path, mp, ms)
    assert planner.read_metrics(path) == (mp, ms)


def test_evaluation_block_at_the_reference_size_against_the_oracle_alone(ctx, oracle, scene):
    """run.py:226-277 as the reference runs it, at ITS size: a 1280x720 test view through the dataset's own intrinsics and
    OpenCV lens (render_with_lens_distortion, run.py:145), the engine's stepping rule, min_T 1e-4, black opaque
    background -- prv_evaluate's PSNR and SSIM of the full-size field against a reference image rendered by the ORACLE,
    compared with the oracle's metrics of the oracle's own render.  Nothing of the GPU's on the checking side."""
    from tests.test_gpu_parity import REF_INTR

    pts = util.fibonacci_hemisphere(4)
    tms, scale, offset = util.hemisphere_transforms(oracle, pts)
    cs = ctx.cameras_from_matrices_intr(tms, REF_INTR, scale, offset)
    w, h, v = 1280, 720, 2
    oc = oracle.cameras_from_dataset(tms, REF_INTR, scale, offset)[v]
    bg = (0.0, 0.0, 0.0, 1.0)
    fa, fb = oracle.OracleField(oracle.desc(), seed=util.SEED_A), oracle.OracleField(oracle.desc(), seed=util.SEED_B)
    mine, _ = fa.render(oc, w, h, 0, 1, 1e-4, threads=16, step_mode=oracle.STEP_NGP)
    ref, _ = fb.render(oc, w, h, 0, 1, 1e-4, threads=16, step_mode=oracle.STEP_NGP)
    fa.close()
    fb.close()
    want_psnr, want_ssim = oracle.score_view(mine, ref, bg)[1], oracle.ssim(mine, ref, bg)
    gt = ctx.torch.from_numpy(ref[None]).cuda()
    opts = api.engine_render_opts(w, h, 0, 1, 1e-4, background=bg)  # run.py:231-245: spp 8 SNAPPED to pixel centres = one sub-sample
    psnr, ssim = ctx.evaluate(0, cs, [v], opts, gt)
    # measured: PSNR 32.6792700 against 32.6792695, SSIM equal to 1.5e-8
    assert psnr == pytest.approx(want_psnr, abs=1e-3) and ssim == pytest.approx(want_ssim, abs=1e-5)
    assert 5.0 < want_psnr < 80.0 and mine[..., 3].max() > 0.9
    cs.close()


@pytest.mark.parametrize("which,patch", [("256", (0, 0)), ("512", (0, 0)), ("256", (4, 4)), ("512", (2, 2))])
def test_training_gradients_on_the_full_size_fields(ctx, oracle, which, patch):
    """one training batch on the BASELINE fields (256^3: L=8 F=4 T=2^19; 512^3: L=16 F=2 T=2^21) at 128 samples
    per ray: same ray batch, loss and gradients within 1e-3 of the oracle -- covers the full-size level tables
    (dense up to 79^3 / 215^3 vertices, 2^19 / 2^21-entry hashed levels) in the encoder's backward scatter"""
    fd = dict(api.FIELD_256 if which == "256" else api.FIELD_512)
    fd.update(density_bias=1.0, table_amp=0.5)  # moderate density: rays go several samples deep
    d_p = api.L.FieldDesc(**fd)
    d_o = oracle.desc(**fd)
    f = oracle.OracleField(d_o, seed=util.SEED_A)
    t, m, o = f.params()
    ctx.load_model(3, d_p, t, m, o)
    ctx.synthetic_model(1, api.L.FieldDesc(**dict(fd, density_bias=3.0, table_amp=4.0)), util.SEED_B)
    pts = planner.hemisphere_generate(6)
    tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
    intr = {"fl_x": 70.0, "fl_y": 69.0, "cx": 47.5, "cy": 36.2, "w": 96, "h": 72, "k1": 0.05, "k2": -0.02, "p1": 0.001, "p2": -0.002}
    cams = ctx.cameras_from_matrices_intr(tms, intr, scale, offset)
    u8, _ = ctx.render_rgba8(1, cams, None, api.render_opts(96, 72, S, 1, 1e-4, background=(0, 0, 0, 0)))
    imgs = u8.cpu().numpy()
    ocams = oracle.cameras_from_dataset(tms, intr, scale, offset)
    base = dict(n_rays=192, n_samples=S, occ_every=0, patch_w=patch[0], patch_h=patch[1])  # patches: the merging scatter on full-size tables
    otr = oracle.OracleTrainer(f, oracle.train_opts(**base), ocams, imgs)
    gtr = api.Trainer(ctx, 3, cams, u8, api.train_opts(**base))
    want_loss, want_tg, want_mg = otr.gradients()
    loss, tg, mg = gtr.gradients()
    assert gtr.info()["samples_last"] == otr.samples_last > 300
    assert loss == pytest.approx(want_loss, rel=1e-3)
    for got, want in ((mg, want_mg), (tg, want_tg)):
        assert np.linalg.norm(got - want) <= 1e-3 * np.linalg.norm(want)
        big = np.abs(want) > 1e-3 * np.abs(want).max()
        assert big.sum() > 50
        np.testing.assert_allclose(got[big], want[big], rtol=5e-3)
    assert np.array_equal(np.flatnonzero(tg), np.flatnonzero(want_tg.astype(np.float32))) or \
        np.linalg.norm(tg - want_tg) <= 1e-3 * np.linalg.norm(want_tg)


def test_tail_merge_and_pool_do_not_change_a_full_size_image(ctx, scene, monkeypatch):
    """at 800x800 the 64-slot kernel relocates thinned-out cohorts' rays (in-wave merge, the block's LDS pool: on by default
    for this size): which lane composites a ray changes, not one bit of the image or the sample count"""
    desc, cams, _ = scene
    opts = api.render_opts(W, H, S, 1, 1e-4)
    want, st = ctx.render(0, cams, [1, 6], opts)
    monkeypatch.setenv("PRV_MERGE_MAX", "0")
    monkeypatch.setenv("PRV_POOL", "0")
    plain = api.Context(0)
    monkeypatch.delenv("PRV_MERGE_MAX")
    monkeypatch.delenv("PRV_POOL")
    try:
        plain.synthetic_model(0, desc, util.SEED_A)
        pc = plain.cameras_from_matrices(*[scene[2][0], util.FOV_X, W, H, scene[2][1], scene[2][2]])
        got, st2 = plain.render(0, pc, [1, 6], opts)
        assert bool((got.cpu() == want.cpu()).all()) and st2.samples_evaluated == st.samples_evaluated
        pc.close()
    finally:
        plain.close()
