"""The bench workload itself against the oracle, at full size: the 64-view hemisphere round bench.py times
(800x800, 128 samples/ray; BASELINE configs[1]) rendered in ONE call with everything the product path switches on at
that size -- tail merge, the block's tail pool, queue regions drained per XCD, default blocks per CU -- and compared
WHOLE VIEW by whole view (eight of the set, every ninth from the pole view to the lowest one) with the CPU oracle, for
  * the literal BASELINE.md section 6 scene (table U(-0.1,0.1), no density bias): bench.py's headline,
  * the denser scene the other full-size tests use (table U(-4,4), bias 3: early termination exercised),
  * the 512^3 field of configs[3] (L=16, F=2, log2T=21: the HBM-bound instance).
The march count of the WHOLE 64-view round (samples in occupied cells: an integer per round, 200+ M) must equal the
oracle's; on the section 6 scene no ray terminates early, so the evaluated-sample count of the round equals it too.
Pixels: 1e-3 relative (tests/util.py).  Oracle time: ~0.5 s per whole view on 16 cores, a few seconds for the round's march count."""
import os

import numpy as np
import pytest

from nerf_prv_amd import api, planner
from tests import util

pytestmark = pytest.mark.gpu

W = H = 800
S = 128
N_VIEWS = 64
THREADS = min(64, os.cpu_count() or 8)
WHOLE_VIEWS = tuple(range(0, N_VIEWS, 9))  # every ninth view, from the pole view (0) to the lowest one (63)
SCENES = {
    "baseline256": dict(api.FIELD_256, table_amp=0.1, density_bias=0.0),
    "default256": dict(api.FIELD_256),
    "default512": dict(api.FIELD_512),
}


@pytest.fixture(scope="module")
def round_cams(ctx, oracle):
    pts = planner.hemisphere_generate(N_VIEWS)  # bench.py's candidate set
    tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
    cams = ctx.cameras_from_matrices(tms, util.FOV_X, W, H, scale, offset)
    ocams = oracle.cameras_from_transforms(tms, util.FOV_X, W, H, scale, offset)
    yield cams, ocams
    cams.close()


@pytest.fixture(scope="module")
def round_march_count(oracle, round_cams):
    """the three scenes share the analytic occupancy grid, hence the march count: computed once"""
    _, ocams = round_cams
    f = oracle.OracleField(oracle.desc(**SCENES["baseline256"]), seed=util.SEED_A)
    per_view = [f.march_count(oc, W, H, S, threads=THREADS) for oc in ocams]
    f.close()
    return per_view


@pytest.mark.parametrize("scene", list(SCENES))
def test_whole_views_and_the_rounds_march_count(ctx, oracle, round_cams, round_march_count, scene):
    cams, ocams = round_cams
    kw = SCENES[scene]
    ctx.synthetic_model(6, api.L.FieldDesc(**kw), util.SEED_A)
    f = oracle.OracleField(oracle.desc(**kw), seed=util.SEED_A)
    assert np.array_equal(f.params()[2], oracle.OracleField(oracle.desc(**SCENES["baseline256"]), seed=util.SEED_A).params()[2])
    opts = api.render_opts(W, H, S, 1, 1e-4)
    img, st = ctx.render(6, cams, None, opts)  # all 64 views, one call: one march + one render launch, as bench.py's step
    assert st.rays == N_VIEWS * W * H
    assert int(st.samples_live) == sum(round_march_count)
    if scene == "baseline256":
        assert int(st.samples_evaluated) == int(st.samples_live)  # nothing terminates early in the section 6 scene
    else:
        assert 0 < int(st.samples_evaluated) < int(st.samples_live)
    n_eval_views = 0
    for v in WHOLE_VIEWS:
        want, ne = f.render(ocams[v], W, H, S, 1, 1e-4, threads=THREADS)
        util.assert_pixels_close(img[v].cpu().numpy(), want)
        assert want[..., 3].max() > (0.1 if scene == "baseline256" else 0.9)
        n_eval_views += ne
        if scene == "baseline256":
            assert ne == round_march_count[v]
    # the same views alone: identical images (a view does not depend on its batch), and their evaluated count
    alone, st2 = ctx.render(6, cams, list(WHOLE_VIEWS), opts)
    assert all(bool((alone[k] == img[v]).all()) for k, v in enumerate(WHOLE_VIEWS))
    assert abs(int(st2.samples_evaluated) - n_eval_views) <= max(2, n_eval_views // 100000)
    f.close()


def test_first_hit_of_the_whole_round_is_bit_exact(ctx, oracle, round_cams):
    """configs[0]'s path at the bench size (a13, main.cpp:238-284 as a ray cast per pixel): the first occupied voxel of
    every pixel of all 64 views at 800x800 -- 41 M integer voxel ids -- equal to the oracle's, bit for bit"""
    cams, ocams = round_cams
    kw = SCENES["baseline256"]
    ctx.synthetic_model(6, api.L.FieldDesc(**kw), util.SEED_A)
    f = oracle.OracleField(oracle.desc(**kw), seed=util.SEED_A)
    got = ctx.first_hit(6, cams, None, W, H).cpu().numpy()
    hits = 0
    for v, oc in enumerate(ocams):
        want = oracle.first_hit_image(f, oc, W, H)
        assert np.array_equal(got[v], want), v
        hits += int((want >= 0).sum())
    assert 0.02 * N_VIEWS * W * H < hits < 0.5 * N_VIEWS * W * H  # the object is seen, and so is the empty space around it
    f.close()


@pytest.mark.parametrize("scene,stride", [("baseline256", 6), ("baseline512", 12)])
def test_the_scoring_round_end_to_end_against_the_oracle_alone(ctx, oracle, round_cams, scene, stride):
    """bench.py's step with NOTHING of the GPU's on the checking side: eleven of the round's 64 views of the section 6
    scene (and six of the same scene on configs[3]'s 512^3 field, the HBM-bound instance), reference images = the oracle's renders of the second field (seed B; uploaded as the round's gt), scored by
    the fused GPU round (march + render + PSNR / coverage reduce + rank) -- against the oracle's renders of the first field
    scored by the oracle's recipe (run.py:257-263, main.cpp:2148).  PSNR to 1e-3 dB, coverage to 1e-5, the integer ranking
    identical (no tolerance)."""
    cams, ocams = round_cams
    kw = SCENES[scene] if scene in SCENES else dict(api.FIELD_512, table_amp=0.1, density_bias=0.0)  # bench.py's `field512`
    ids = list(range(1, N_VIEWS, stride))
    ctx.synthetic_model(6, api.L.FieldDesc(**kw), util.SEED_A)
    fa = oracle.OracleField(oracle.desc(**kw), seed=util.SEED_A)
    fb = oracle.OracleField(oracle.desc(**kw), seed=util.SEED_B)
    mine = [fa.render(ocams[v], W, H, S, 1, 1e-4, threads=THREADS)[0] for v in ids]
    refs = [fb.render(ocams[v], W, H, S, 1, 1e-4, threads=THREADS)[0] for v in ids]
    fa.close()
    fb.close()
    want = np.array([oracle.score_view(a, b) for a, b in zip(mine, refs)])  # (score, psnr, coverage)
    gt = ctx.torch.from_numpy(np.stack(refs)).cuda()
    opts = api.render_opts(W, H, S, 1, 1e-4)
    rec, st = ctx.score_views(api.L.SCORE_PSNR_COVERAGE, [6], cams, ids, opts, gt=gt, want_stats=True)
    assert st.rays == len(ids) * W * H and np.all(np.isfinite(rec["psnr"]))
    np.testing.assert_allclose(rec["psnr"], want[:, 1], atol=1e-3)  # measured: 4e-5 dB on PSNRs of 46..58 dB
    np.testing.assert_allclose(rec["coverage"], want[:, 2], rtol=1e-5)  # measured: 3e-7
    np.testing.assert_allclose(rec["score"], want[:, 0], atol=1e-3)
    got_order, want_order = ctx.rank(rec, np.asarray(ids, np.int32)), oracle.rank(want[:, 0], np.asarray(ids, np.int32))
    score_of = dict(zip(ids, want[:, 0]))
    # north_star: integer view rankings bit-exact.  No tolerance: if this ever fails, the message says how close the oracle's
    # own scores of the swapped views are (a near-tie below the PSNR agreement above is the one excusable cause)
    swapped = [(int(a), int(b), float(abs(score_of[int(a)] - score_of[int(b)]))) for a, b in zip(got_order, want_order) if a != b]
    assert np.array_equal(got_order, want_order), f"ranking differs from the oracle's at (got, want, |oracle score gap|): {swapped}"
    assert want[:, 1].max() - want[:, 1].min() > 0.5  # the views do differ: the ranking is not a coin toss


@pytest.mark.parametrize("scene", ["default256", "default512"])
def test_ngp_step_at_full_size(ctx, oracle, round_cams, scene):
    """the engine's own stepping rule (PRV_STEP_NGP, what run.py:304 renders with) on the full-size fields: the march
    count of two whole 800x800 views exactly (every step's occupancy decision, ~700 steps per ray), the middle 240 rows of
    each against the oracle, the engine's default min_T 0.01 (hence the termination variants of tests/util.py)"""
    cams, ocams = round_cams
    kw = SCENES[scene]
    ctx.synthetic_model(6, api.L.FieldDesc(**kw), util.SEED_A)
    f = oracle.OracleField(oracle.desc(**kw), seed=util.SEED_A)
    views = [0, N_VIEWS - 1]
    opts = api.engine_render_opts(W, H, 0, 1, 1e-2)
    img, st = ctx.render(6, cams, views, opts)
    assert int(st.samples_live) == sum(f.march_count(ocams[v], W, H, 0, threads=THREADS, step_mode=oracle.STEP_NGP) for v in views)
    assert st.samples_nominal == 2 * W * H * api.L.NGP_MAX_STEPS and 0 < st.samples_evaluated < st.samples_live
    rows = (H // 2 - 120, H // 2 + 120)
    n_eval = 0
    for k, v in enumerate(views):
        wants = [f.render(ocams[v], W, H, 0, 1, t, threads=THREADS, rows=rows, step_mode=oracle.STEP_NGP) for t in util.termination_variants(1e-2)]
        util.assert_pixels_close_any(img[k].cpu().numpy()[rows[0]:rows[1]], [x[0][rows[0]:rows[1]] for x in wants])
        assert wants[0][0][rows[0]:rows[1], :, 3].max() > 0.9
        n_eval += wants[0][1]
    assert n_eval > 0
    f.close()


@pytest.mark.parametrize("scene", ["baseline256", "baseline512"])
def test_ngp_step_whole_views_of_the_headline_scene(ctx, oracle, round_cams, scene):
    """the engine's stepping rule on the HEADLINE scene (BASELINE.md section 6 literally; and the same scene on configs[3]'s
    512^3 field): the 64-view 800x800 round in one call -- its march count (every occupancy decision of every step of 41 M
    rays: 1.1 G samples in occupied cells) equal to the oracle's as an integer, and four WHOLE views (pole, two in
    between, the lowest) against the oracle pixel by pixel.  Nothing terminates early in this scene (alpha stays small),
    so evaluated == live and min_T needs no termination variants."""
    cams, ocams = round_cams
    kw = SCENES[scene] if scene in SCENES else dict(api.FIELD_512, table_amp=0.1, density_bias=0.0)
    ctx.synthetic_model(6, api.L.FieldDesc(**kw), util.SEED_A)
    f = oracle.OracleField(oracle.desc(**kw), seed=util.SEED_A)
    opts = api.engine_render_opts(W, H, 0, 1, 1e-2)
    img, st = ctx.render(6, cams, None, opts)
    want_march = sum(f.march_count(oc, W, H, 0, threads=THREADS, step_mode=oracle.STEP_NGP) for oc in ocams)
    assert int(st.samples_live) == want_march and want_march > 1_000_000_000
    assert int(st.samples_evaluated) == int(st.samples_live)
    assert st.samples_nominal == N_VIEWS * W * H * api.L.NGP_MAX_STEPS
    n_eval = 0
    for v in (0, 21, 42, 63):
        want, ne = f.render(ocams[v], W, H, 0, 1, 1e-2, threads=THREADS, step_mode=oracle.STEP_NGP)
        util.assert_pixels_close(img[v].cpu().numpy(), want)
        assert want[..., 3].max() > 0.1
        n_eval += ne
    assert n_eval > 0
    f.close()


def test_whole_views_of_a_table_beyond_the_caches(ctx, oracle, round_cams, round_march_count):
    """round 6: a field whose hashed levels are 64 MiB each (log2T = 24, F = 2, finest 2048: seven hashed levels = 448 MiB of
    random gathers, beyond the 256 MiB Infinity Cache -- bench.py's `field_hbm`).  Levels larger than 16 MiB take the generic
    gather with 32-bit offsets (FieldDev::wide_offsets); the limit of rounds 1-5 refused them.  Four views of the 64-view
    round in one call: the march count is the analytic grid's, four WHOLE 800x800 views against the oracle pixel by pixel."""
    cams, ocams = round_cams
    kw = dict(api.FIELD_HBM, table_amp=0.1, density_bias=0.0)
    ctx.synthetic_model(6, api.L.FieldDesc(**kw), util.SEED_A)
    lay = ctx.model_layout(6)
    assert lay["n_hashed_levels"] == 7 and lay["n_dense_levels"] == 9
    f = oracle.OracleField(oracle.desc(**kw), seed=util.SEED_A)
    views = [0, 21, 42, 63]
    opts = api.render_opts(W, H, S, 1, 1e-4)
    img, st = ctx.render(6, cams, views, opts)
    assert int(st.samples_live) == sum(round_march_count[v] for v in views) == int(st.samples_evaluated)
    for k, v in enumerate(views):
        want, ne = f.render(ocams[v], W, H, S, 1, 1e-4, threads=THREADS)
        util.assert_pixels_close(img[k].cpu().numpy(), want)
        assert ne == round_march_count[v] and want[..., 3].max() > 0.05
    f.close()
    ctx.synthetic_model(6, api.L.FieldDesc(**SCENES["baseline256"]), util.SEED_A)  # (give the gigabyte back)
