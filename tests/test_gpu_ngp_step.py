"""PRV_STEP_NGP on the GPU: instant-ngp's stepping rule (what run.py:245-247, 304 renders with -- fixed step
dt = sqrt(3)/1024 from the AABB entry, every step tested against the occupancy grid, no per-ray sample cap)
through the C ABI, against the CPU oracle's restatement of the same rule (oracle/prv_oracle.c: march_ray) and the
independent numpy golden (tests/golden/golden_render_ngp.json).  Bars as everywhere: the march count (samples in occupied
cells) is an integer and must be EXACT; evaluated samples may differ by a ray stopping one sample early or late (hardware
exp vs expf at T ~ min_T); pixels 1e-3 relative (tests/util.py)."""
import json
import os

import numpy as np
import pytest

from nerf_prv_amd import api
from tests import util

pytestmark = pytest.mark.gpu
NGP = api.L.STEP_NGP


@pytest.fixture(scope="module", params=["F4", "F2"])
def fields(request, ctx, oracle):
    kw = util.SMALL if request.param == "F4" else util.SMALL_F2
    d_o, d_p = oracle.desc(**kw), api.field_desc(**kw)
    f = oracle.OracleField(d_o, seed=util.SEED_A)
    ctx.synthetic_model(0, d_p, util.SEED_A)
    return d_o, d_p, f


@pytest.fixture(scope="module")
def cams(ctx, oracle):
    pts = util.fibonacci_hemisphere(5)
    tms, scale, offset = util.hemisphere_transforms(oracle, pts)
    w, h = 44, 36
    cs = ctx.cameras_from_matrices(tms, util.FOV_X, w, h, scale, offset)
    ocams = oracle.cameras_from_transforms(tms, util.FOV_X, w, h, scale, offset)
    return cs, ocams, w, h


@pytest.mark.parametrize("min_T,spp", [(1e-2, 1), (1e-4, 1), (1e-2, 4)])
def test_ngp_step_pixels_and_counts(ctx, oracle, fields, cams, min_T, spp):
    d_o, d_p, f = fields
    cs, ocams, w, h = cams
    opts = api.render_opts(w, h, 0, spp, min_T, step_mode=NGP)
    img, st = ctx.render(0, cs, None, opts)
    img = img.cpu().numpy()
    n_eval = n_live = 0
    for v, oc in enumerate(ocams):
        wants = [f.render(oc, w, h, 0, spp, t, step_mode=oracle.STEP_NGP) for t in util.termination_variants(min_T)]
        n_eval += wants[0][1]
        n_live += f.march_count(oc, w, h, 0, spp, step_mode=oracle.STEP_NGP)
        util.assert_pixels_close_any(img[v], [x[0] for x in wants])  # tests/util.py: a ray may take either side of the threshold
    assert int(st.samples_live) == n_live > 0  # the march pass's masks: every step's occupancy decision, exactly
    assert abs(int(st.samples_evaluated) - n_eval) <= max(4, n_eval // 20000)
    assert st.rays == len(ocams) * w * h * spp and st.samples_nominal == st.rays * api.L.NGP_MAX_STEPS
    assert n_live > 128 * 10  # rays with more live samples than one mask chunk exist in this scene (checked per ray below)


def test_fixed_mode_march_count_is_exact_too(ctx, oracle, fields, cams):
    d_o, d_p, f = fields
    cs, ocams, w, h = cams
    for S in (128, 37):
        _, st = ctx.render(0, cs, None, api.render_opts(w, h, S, 1, 1e-4))
        assert int(st.samples_live) == sum(f.march_count(oc, w, h, S) for oc in ocams) > 0


def test_ngp_step_against_the_numpy_golden(ctx, oracle):
    """the fixture's camera sees rays with up to 430 live steps (4 mask chunks); counts per image, pixels per pixel"""
    with open(os.path.join(os.path.dirname(__file__), "golden", "golden_render_ngp.json")) as fh:
        g = json.load(fh)
    ctx.synthetic_model(3, api.field_desc(**g["desc"]), g["seed"])
    w, h = g["w"], g["h"]
    c2w = np.array(g["c2w"], np.float64).reshape(3, 4)
    # the fixture states its camera in the ENGINE frame; prv_cameras_from_matrices takes transform_matrix + scale/offset
    # (nerf_matrix_to_ngp: columns 1,2 negated, axes cycled) -- invert that here
    tm = np.eye(4)
    eng = c2w.copy()
    eng[:, 1] *= -1
    eng[:, 2] *= -1
    tm[:3, :] = eng[[2, 0, 1], :]  # engine (x,y,z) = nerf (y,z,x)
    fov = 2.0 * np.arctan(0.5 * w / g["fx"])
    cs = ctx.cameras_from_matrices(tm[None], fov, w, h, 1.0, np.zeros(3))
    got_c2w, got_intr = cs.get(0)
    np.testing.assert_allclose(got_c2w, c2w, atol=1e-7)
    assert abs(got_intr[0] - g["fx"]) <= 1e-5 * g["fx"]
    img, st = ctx.render(3, cs, None, api.render_opts(w, h, 0, 1, g["min_T"], step_mode=NGP))
    assert int(st.samples_live) == g["n_live"]
    assert abs(int(st.samples_evaluated) - g["n_evaluated"]) <= 2
    util.assert_pixels_close(img[0].cpu().numpy(), np.array(g["image"]))
    per_ray = np.array(g["per_ray"])
    assert per_ray[:, 1].max() > 384  # four chunks
    cs.close()


def test_ngp_step_awkward_cameras(ctx, oracle, fields):
    """camera inside the object, axis-aligned rays, far away, looking away, grazing: the clipped step range and the
    word-level coarse skip may only ever err towards testing more"""
    d_o, d_p, f = fields
    w = h = 33

    def tm(rot, t):
        m = np.eye(4)
        m[:3, :3] = rot
        m[:3, 3] = t
        return m

    eye, back = np.eye(3), np.diag([-1.0, 1.0, -1.0])
    ry = lambda a: np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    cases = [("inside", tm(eye, [0, 0, 0]), util.FOV_X), ("axis", tm(eye, [0, 0, 0.3]), util.FOV_X), ("away", tm(back, [0, 0, 0.3]), util.FOV_X),
             ("far", tm(eye, [0, 0, 60.0]), 0.02), ("corner", tm(ry(0.6), [0.25, 0.07, 0.28]), 0.5),
             ("diag", tm(ry(0.7853981), [0.2, 0.0, 0.2]), 0.3)]
    cases += [(f"graze{i}", tm(eye, [0.04 * i, 0.02 * i, 0.3]), 0.6) for i in range(1, 6)]
    scale, offset = 5.0, np.array([0.5, 0.5, 0.5])
    for name, m, fov in cases:
        cs = ctx.cameras_from_matrices(m[None], fov, w, h, scale, offset)
        oc = oracle.cameras_from_transforms(m[None], fov, w, h, scale, offset)[0]
        img, st = ctx.render(0, cs, None, api.render_opts(w, h, 0, 1, 1e-2, step_mode=NGP))
        wants = [f.render(oc, w, h, 0, 1, t, step_mode=oracle.STEP_NGP) for t in util.termination_variants(1e-2)]
        util.assert_pixels_close_any(img[0].cpu().numpy(), [x[0] for x in wants])
        assert int(st.samples_live) == f.march_count(oc, w, h, 0, step_mode=oracle.STEP_NGP), name
        assert abs(int(st.samples_evaluated) - wants[0][1]) <= 4, (name, int(st.samples_evaluated), wants[0][1])
        cs.close()


def test_ngp_step_transparent_field_walks_every_chunk(ctx, oracle):
    """a nearly transparent field (BASELINE.md section 6's table amplitude and density bias) never terminates early:
    every live step of every ray is evaluated, so evaluated == live == the oracle's march count, chunk after chunk"""
    kw = dict(util.SMALL, table_amp=0.1, density_bias=0.0)
    f = oracle.OracleField(oracle.desc(**kw), seed=util.SEED_A)
    ctx.synthetic_model(3, api.field_desc(**kw), util.SEED_A)
    pts = util.fibonacci_hemisphere(3)
    tms, scale, offset = util.hemisphere_transforms(oracle, pts)
    w, h = 40, 30
    cs = ctx.cameras_from_matrices(tms, util.FOV_X, w, h, scale, offset)
    ocams = oracle.cameras_from_transforms(tms, util.FOV_X, w, h, scale, offset)
    img, st = ctx.render(3, cs, None, api.render_opts(w, h, 0, 1, 1e-2, step_mode=NGP))
    live = sum(f.march_count(oc, w, h, 0, step_mode=oracle.STEP_NGP) for oc in ocams)
    assert int(st.samples_live) == int(st.samples_evaluated) == live
    for v, oc in enumerate(ocams):
        want, _ = f.render(oc, w, h, 0, 1, 1e-2, step_mode=oracle.STEP_NGP)
        util.assert_pixels_close(img[v].cpu().numpy(), want)
    cs.close()


def test_ngp_step_scores_and_bytes_follow(ctx, oracle, fields, cams):
    """the scoring round and the PNG bytes in the engine's own stepping mode: same records as scoring the same renders"""
    d_o, d_p, f = fields
    cs, ocams, w, h = cams
    ctx.synthetic_model(1, d_p, util.SEED_B)
    opts = api.render_opts(w, h, 0, 4, 1e-2, background=(0.0, 0.0, 0.0, 1.0), step_mode=NGP)
    gt, _ = ctx.render(1, cs, None, opts, want_stats=False)
    img, _ = ctx.render(0, cs, None, opts, want_stats=False)
    rec, _ = ctx.score_views(api.L.SCORE_PSNR_COVERAGE, [0], cs, None, opts, gt=gt)
    assert rec.tobytes() == ctx.score_psnr_images(img, gt, background=(0.0, 0.0, 0.0, 1.0)).tobytes()
    u8, _ = ctx.render_rgba8(0, cs, None, opts)
    want8 = oracle.quantize_rgba8(img.cpu().numpy(), (0.0, 0.0, 0.0, 1.0))
    assert np.array_equal(u8.cpu().numpy(), want8)
    ens, _ = ctx.score_views(api.L.SCORE_ENSEMBLE_RGB_DENSITY, [0, 1], cs, None, opts)
    u8b, _ = ctx.render_rgba8(1, cs, None, opts)
    for v in range(len(ocams)):
        want = oracle.score_ensemble_rgbdensity([u8[v].cpu().numpy(), u8b[v].cpu().numpy()])
        assert ens["score"][v] == pytest.approx(want, rel=1e-12)


def test_step_mode_is_validated(ctx, fields, cams):
    cs, ocams, w, h = cams
    with pytest.raises(api.PrvError) as e:
        ctx.render(0, cs, None, api.render_opts(w, h, 64, 1, 1e-2, step_mode=7))
    assert e.value.code == api.L.PRV_E_INVALID
    with pytest.raises(api.PrvError):  # the fixed mode still needs a sample count
        ctx.render(0, cs, None, api.render_opts(w, h, 0, 1, 1e-2))
    ctx.render(0, cs, None, api.render_opts(w, h, 0, 1, 1e-2, step_mode=NGP))  # NGP ignores it


def test_ngp_step_through_the_lens_cameras_of_the_evaluation_block(ctx, oracle, fields):
    """run.py:238-247 renders the test views with their own intrinsics and lens (render_with_lens_distortion) and
    `render_min_transmittance = 1e-4`, through the same engine rule: dataset cameras here, march count exact, pixels vs
    the oracle; and prv_evaluate's PSNR over them equals the oracle's recipe on the oracle's renders to 1e-3"""
    from tests.test_gpu_parity import REF_INTR

    d_o, d_p, f = fields
    ctx.synthetic_model(1, d_p, util.SEED_B)
    fb = oracle.OracleField(d_o, seed=util.SEED_B)
    pts = util.fibonacci_hemisphere(4)
    tms, scale, offset = util.hemisphere_transforms(oracle, pts)
    cs = ctx.cameras_from_matrices_intr(tms, REF_INTR, scale, offset)
    w, h = 64, 36
    ocams = oracle.cameras_from_dataset(tms, REF_INTR, scale, offset, w, h)
    opts = api.engine_render_opts(w, h, 0, 1, 1e-4)
    img, st = ctx.render(0, cs, None, opts)
    gt, _ = ctx.render(1, cs, None, opts, want_stats=False)
    live, psnrs = 0, []
    for v, oc in enumerate(ocams):
        want, _ = f.render(oc, w, h, 0, 1, 1e-4, step_mode=oracle.STEP_NGP)
        ref, _ = fb.render(oc, w, h, 0, 1, 1e-4, step_mode=oracle.STEP_NGP)
        live += f.march_count(oc, w, h, 0, step_mode=oracle.STEP_NGP)
        util.assert_pixels_close(img[v].cpu().numpy(), want)
        psnrs.append(oracle.score_view(want, ref, (0.0, 0.0, 0.0, 1.0))[1])
    assert int(st.samples_live) == live > 0
    bg_opts = api.engine_render_opts(w, h, 0, 1, 1e-4, background=(0.0, 0.0, 0.0, 1.0))
    psnr, ssim = ctx.evaluate(0, cs, None, bg_opts, gt)
    assert psnr == pytest.approx(np.mean(psnrs), rel=1e-3) and 0.0 < ssim <= 1.0
    cs.close()


@pytest.mark.parametrize("mode", ["fixed", "ngp"])
def test_tile_rejection_never_drops_a_live_ray_random_cameras(ctx, oracle, fields, mode):
    """the march pass skips whole 16x16 tiles outside the pixel rectangle of the occupied box's projected corners
    (prv_api.cpp: set_cull_rect) and the fused PSNR round does not even launch them: 40 random cameras -- near and far,
    looking at, past and away from the object, rolled, some inside the cube -- must give the oracle's march count exactly
    (one dropped live ray would show), through prv_render AND through the fused scoring round's private image"""
    d_o, d_p, f = fields
    rng = np.random.default_rng(20260)
    w, h = 96, 72
    step = oracle.STEP_NGP if mode == "ngp" else oracle.STEP_FIXED_S
    opts = api.engine_render_opts(w, h, 0 if mode == "ngp" else 96, 1, 1e-4)
    mats = []
    for k in range(40):
        eye = rng.normal(size=3)
        eye = eye / np.linalg.norm(eye) * rng.choice([0.02, 0.08, 0.15, 0.3, 0.6, 2.5])
        target = rng.normal(size=3) * rng.choice([0.0, 0.02, 0.08, 0.3])
        fwd = target - eye
        fwd /= np.linalg.norm(fwd)
        up = rng.normal(size=3)
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        m = np.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up, -fwd, eye  # NeRF convention: the camera looks down -z
        mats.append(m)
    mats = np.array(mats)
    scale, offset = 5.0, np.array([0.5, 0.5, 0.5])
    fov = 0.9
    cs = ctx.cameras_from_matrices(mats, fov, w, h, scale, offset)
    ocams = oracle.cameras_from_transforms(mats, fov, w, h, scale, offset)
    want = [f.march_count(oc, w, h, 96, step_mode=step) for oc in ocams]
    assert sum(1 for x in want if x > 0) >= 15 and sum(1 for x in want if x == 0) >= 3  # both kinds of view occur
    for v in range(len(ocams)):
        _, st = ctx.render(0, cs, [v], opts)
        assert int(st.samples_live) == want[v], (v, int(st.samples_live), want[v])
    # the fused round (only the tiles inside the rectangles are launched, the score reads through the rectangles):
    # same records as scoring fully written images
    ctx.synthetic_model(1, d_p, util.SEED_B)
    gt, _ = ctx.render(1, cs, None, opts, want_stats=False)
    img, _ = ctx.render(0, cs, None, opts, want_stats=False)
    rec, st = ctx.score_views(api.L.SCORE_PSNR_COVERAGE, [0], cs, None, opts, gt=gt, want_stats=True)
    assert int(st.samples_live) == sum(want)
    assert rec.tobytes() == ctx.score_psnr_images(img, gt).tobytes()
    cs.close()


def test_corner_cache_instance_renders_the_same_pixels(oracle, monkeypatch):
    """render_queue64_kernel<4, 5, NGP, CACHE>: lanes keep the corner entries of their last cell on the three hashed
    levels and skip the loads while the ray stays in the cell (the default for small images under the engine's rule; forced
    on here for both rules and a large image too).  The cache maps cells to table entries, nothing else: images and counts
    are bit-identical to the plain instance, relocation on or off, and equal to the oracle."""
    kw = dict(api.FIELD_256)
    pts = util.fibonacci_hemisphere(6)
    tms, scale, offset = util.hemisphere_transforms(oracle, pts)
    runs = {}
    for cache, merge in (("0", "0"), ("1", "0"), ("1", "24")):
        monkeypatch.setenv("PRV_CELL_CACHE", cache)
        monkeypatch.setenv("PRV_MERGE_MAX", merge)
        monkeypatch.setenv("PRV_POOL", "1" if merge != "0" else "0")
        c = api.Context(0)
        try:
            c.synthetic_model(0, api.L.FieldDesc(**kw), util.SEED_A)
            out = []
            for (w, h, spp, spr, min_T) in ((80, 45, 16, 0, 1e-2), (80, 45, 4, 128, 1e-2), (400, 300, 1, 0, 1e-4)):
                cams = c.cameras_from_matrices(tms, util.FOV_X, w, h, scale, offset)
                img, st = c.render(0, cams, None, api.engine_render_opts(w, h, spr, spp, min_T))
                out.append((img.cpu().numpy(), int(st.samples_evaluated), int(st.samples_live)))
                cams.close()
            runs[(cache, merge)] = out
        finally:
            c.close()
    base = runs[("0", "0")]
    for key in (("1", "0"), ("1", "24")):
        for (a, ea, la), (b, eb, lb) in zip(base, runs[key]):
            assert np.array_equal(a, b) and ea == eb and la == lb, key
    # and the oracle on one whole 80x45 x 16 spp view under the engine's rule
    f = oracle.OracleField(oracle.desc(**kw), seed=util.SEED_A)
    ocams = oracle.cameras_from_transforms(tms, util.FOV_X, 80, 45, scale, offset)
    wants = [f.render(ocams[2], 80, 45, 0, 16, t, step_mode=oracle.STEP_NGP)[0] for t in util.termination_variants(1e-2)]
    util.assert_pixels_close_any(runs[("1", "24")][0][0][2], wants)
    f.close()


def _ensemble_with_own_occupancies(ctx_, oracle, kw, n, seed0):
    """n members of the same field shape whose occupancy grids DIFFER: the synthetic scene's grid, a solid ball, an
    all-occupied grid (rays with 600+ consecutive live steps: longer than the march's LDS window), an empty one, a slab"""
    d_o, d_p = oracle.desc(**kw), api.field_desc(**kw)
    R = kw["occ_res"]
    z, y, x = np.meshgrid(np.arange(R), np.arange(R), np.arange(R), indexing="ij")
    cx = (x + 0.5) / R - 0.5
    cy = (y + 0.5) / R - 0.5
    cz = (z + 0.5) / R - 0.5
    grids = [None, (cx ** 2 + cy ** 2 + cz ** 2) < 0.3 ** 2, np.ones((R, R, R), bool), np.zeros((R, R, R), bool), np.abs(cz + 0.1) < 0.12,
             (np.abs(cx) < 0.2) & (np.abs(cy) < 0.35), ((x + y + z) % 3 == 0), (cx > 0.1)]
    for e in range(n):
        f = oracle.OracleField(d_o, seed=seed0 + e)
        t, m, o = f.params()
        if grids[e] is not None:
            bits = np.packbits(grids[e].reshape(-1).astype(np.uint8), bitorder="little")
            o = np.frombuffer(bits.tobytes().ljust(o.nbytes, b"\0"), np.uint32).copy()
        ctx_.load_model(e, d_p, t, m, o)
    return d_p


@pytest.mark.parametrize("n_members,method,spp,size", [(5, 3, 16, (80, 45)), (2, 2, 16, (80, 45)), (5, 3, 1, (44, 36)), (5, 3, 3, (44, 36)), (2, 2, 4, (20, 16))])
def test_one_march_launch_for_the_ensemble_gives_every_member_its_own_result(oracle, n_members, method, spp, size):
    """The ensemble's candidates under the engine's rule go through ONE march launch (march_multi_kernel: a ray's steps are
    walked once, a byte of the members' interleaved occupancy answers a step for all of them) and one render launch per
    member.  Against the member-by-member path (PRV_MARCH_MULTI=0: one march launch per member) on members whose occupancy
    grids differ: the same 16-byte records bit for bit (every byte of every member's image enters the score), the same march
    count (every step's occupancy decision of every member), the same evaluated count."""
    kw = util.SMALL
    pts = util.fibonacci_hemisphere(7)
    tms, scale, offset = util.hemisphere_transforms(oracle, pts)
    w, h = size
    got = {}
    for multi in ("0", "1"):
        os.environ["PRV_MARCH_MULTI"] = multi
        try:
            c = api.Context(0)
        finally:
            del os.environ["PRV_MARCH_MULTI"]
        _ensemble_with_own_occupancies(c, oracle, kw, n_members, 4242)
        cs = c.cameras_from_matrices(tms, util.FOV_X, w, h, scale, offset)
        opts = api.engine_render_opts(w, h, 0, spp, 0.01, background=(0, 0, 0, 1))
        rec, st = c.score_views(method, list(range(n_members)), cs, None, opts, want_stats=True)
        rec2, st2 = c.score_views(method, list(range(n_members)), cs, [5, 2, 3], opts, want_stats=True)  # a subset, in another order
        got[multi] = (rec.tobytes(), int(st.samples_live), int(st.samples_evaluated), rec2.tobytes(), int(st2.samples_live), rec)
        cs.close()
        c.close()
    assert got["0"][1] == got["1"][1] > 0 and got["0"][4] == got["1"][4] > 0
    assert got["0"][2] == got["1"][2]
    assert got["0"][0] == got["1"][0] and got["0"][3] == got["1"][3]
    assert np.all(np.isfinite(got["1"][5]["score"])) and len(set(got["1"][5]["score"].tolist())) > 1
