import pytest
"""Hand-derivable known answers for the reference's in-tree formulas (SURVEY 8c(4)).  CPU only."""
import math

import numpy as np


def test_identical_ensemble_scores_zero_and_argmax_lowest_id(oracle):
    img = np.random.default_rng(0).integers(0, 256, (30, 4), dtype=np.uint8)
    assert oracle.score_ensemble_rgb([img, img]) == 0.0  # every variance is 0 <= 1e-10 (main.cpp:2082-2084)
    assert oracle.argmax([0.0, 0.0, 0.0], [3, 5, 9]) == 3  # strict '>' keeps the first (main.cpp:2088)
    assert oracle.argmax([], []) == -1


def test_single_channel_difference_gives_log_delta2_over_4(oracle):
    a = np.full((10, 4), 100, np.uint8)
    b = a.copy()
    b[4, 1] = 106  # delta = 6 in one channel of one pixel: population variance = (delta/2)^2
    assert math.isclose(oracle.score_ensemble_rgb([a, b]), math.log(6 * 6 / 4.0), rel_tol=1e-15)
    b[4, 3] = 0  # alpha is not an RGB term of method 2
    assert math.isclose(oracle.score_ensemble_rgb([a, b]), math.log(9.0), rel_tol=1e-15)


def test_variance_gate_is_strict_at_1e_10(oracle):
    # E=2, values differing by 1: var = 0.25 > 1e-10 contributes log(0.25) < 0 (negative terms are kept)
    a = np.zeros((1, 4), np.uint8)
    b = a.copy()
    b[0, 0] = 1
    assert math.isclose(oracle.score_ensemble_rgb([a, b]), math.log(0.25), rel_tol=1e-15)


def test_rgbdensity_terms(oracle):
    n = 12
    opaque = np.full((n, 4), 255, np.uint8)
    assert oracle.score_ensemble_rgbdensity([opaque] * 5) == 0.0  # no variance, mean alpha = 1
    clear = opaque.copy()
    clear[:, 3] = 0
    assert math.isclose(oracle.score_ensemble_rgbdensity([clear] * 5), n * 1.0, rel_tol=1e-15)  # (1-0)^2 per pixel
    half = opaque.copy()
    half[:, 3] = 51  # 51/255 = 0.2 -> (0.8)^2
    assert math.isclose(oracle.score_ensemble_rgbdensity([half] * 2), n * 0.64, rel_tol=1e-12)
    a, b = opaque.copy(), opaque.copy()
    a[:, 0], b[:, 0] = 10, 20  # var_r = 25, others 0 -> 25/3 per pixel
    assert math.isclose(oracle.score_ensemble_rgbdensity([a, b]), n * 25.0 / 3.0, rel_tol=1e-12)


def test_psnr_known_values(oracle):
    h = w = 8
    img = np.zeros((h, w, 4), np.float32)
    gt = np.zeros((h, w, 4), np.float32)
    img[..., :3], img[..., 3] = 0.5, 1.0
    gt[..., :3], gt[..., 3] = 0.25, 1.0
    s = lambda x: 1.055 * x ** (1 / 2.4) - 0.055
    want = -10 * math.log10((s(0.5) - s(0.25)) ** 2)
    psnr, cov = oracle.score_psnr_coverage(img, gt)
    assert math.isclose(psnr, want, rel_tol=1e-5) and cov == 1.0
    psnr2, cov2 = oracle.score_psnr_coverage(img * np.float32(0.5), gt)  # premultiplied half-opacity
    assert math.isclose(cov2, 0.5, rel_tol=1e-7)
    assert psnr2 == float("inf")  # 0.25 vs 0.25 -> mse = 0: -10*log10(0) (run.py:263 has no guard either)


def test_quantize_rule(oracle):
    rgba = np.array([[0, 0, 0, 0], [1, 1, 1, 1], [0.0031308 * 0.5, 0, 0, 0.5], [0.5, 0.25, 0.125, 1.0]], np.float32)
    q = oracle.quantize_rgba8(rgba, (0, 0, 0, 1))
    assert q[0].tolist() == [0, 0, 0, 255] and q[1].tolist() == [255, 255, 255, 255]
    # opaque black background: alpha becomes 1, colours stay premultiplied -> srgb(0.0015654) = 12.92*x
    assert q[2, 0] == int(12.92 * 0.0031308 * 0.5 * 255 + 0.5) and q[2, 3] == 255
    q2 = oracle.quantize_rgba8(rgba, (0, 0, 0, 0))
    assert q2[0].tolist() == [0, 0, 0, 0]
    assert q2[2, 3] == 128 and q2[2, 0] == int(12.92 * 0.0031308 * 255 + 0.5)  # un-premultiplied by alpha 0.5
    s = lambda x: 1.055 * x ** (1 / 2.4) - 0.055
    assert q2[3].tolist() == [int(s(0.5) * 255 + 0.5), int(s(0.25) * 255 + 0.5), int(s(0.125) * 255 + 0.5), 255]


def test_sh_basis_values(oracle):
    import ctypes as C

    out = np.zeros(16, np.float32)
    d = np.array([0, 0, 1], np.float32)
    oracle.lib().orc_sh4(d.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    want = np.zeros(16)
    want[0] = 0.5 * math.sqrt(1 / math.pi)  # Y00
    want[2] = math.sqrt(3 / (4 * math.pi))  # Y10 at z=1
    want[6] = 0.25 * math.sqrt(5 / math.pi) * 2  # Y20 = (3z^2-1) * sqrt(5/pi)/4
    want[12] = 0.25 * math.sqrt(7 / math.pi) * 2  # Y30 = (5z^3-3z) * sqrt(7/pi)/4
    np.testing.assert_allclose(out, want, atol=1e-6)


def test_camera_model_roundtrip(oracle):
    """rs2 project/deproject (Share_Data.hpp:92-196): pinhole round trip; inverse Brown-Conrady is
    applied on deprojection only (model 2), so project(deproject(px)) != px there -- as in the reference"""
    import ctypes as C

    intr = np.array([647.145, 372.515, 915.607, 913.327, 0.12042, -0.21373, 0.005386, -0.002121, 0.0], np.float32)
    px = np.array([100.0, 600.0], np.float32)
    pt = np.zeros(3, np.float32)
    back = np.zeros(2, np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    oracle.lib().orc_rs2_deproject(p(pt), p(intr), 0, p(px), C.c_float(2.0))
    assert pt[2] == 2.0
    oracle.lib().orc_rs2_project(p(back), p(intr), 0, p(pt))
    np.testing.assert_allclose(back, px, atol=1e-3)
    oracle.lib().orc_rs2_deproject(p(pt), p(intr), 2, p(px), C.c_float(1.0))
    x, y = (100.0 - 647.145) / 915.607, (600.0 - 372.515) / 913.327
    r2 = x * x + y * y
    f = 1 + 0.12042 * r2 - 0.21373 * r2 * r2 + 0.0 * r2 ** 3  # coeffs[4] multiplies r^6 (Share_Data.hpp:150)
    ux = x * f + 2 * 0.005386 * x * y - 0.002121 * (r2 + 2 * x * x)
    np.testing.assert_allclose(pt[0], ux, rtol=1e-5)


def test_first_hit_dda(oracle):
    import ctypes as C

    f = oracle.OracleField(oracle.desc(n_levels=8, n_features=4, log2_hashmap=9, base_res=4, finest_res=32, occ_res=32),
                           seed=1)
    o = np.array([0.5, 0.5, -1.0], np.float32)
    d = np.array([0, 0, 1], np.float32)
    cell = np.zeros(3, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert oracle.lib().orc_first_hit(f.ptr, p(o), p(d), C.c_float(10.0), p(cell)) == 1
    # main sphere r=0.35 about 0.5: first occupied cell centre along +z through the middle is at z ~ 0.15
    assert cell[0] == 16 and cell[1] == 16 and 4 <= cell[2] <= 5
    o2 = np.array([0.02, 0.02, -1.0], np.float32)  # corner column misses every sphere
    assert oracle.lib().orc_first_hit(f.ptr, p(o2), p(d), C.c_float(10.0), p(cell)) == 0


def test_lens_known_answers(oracle):
    """hand-derived: no lens terms -> identity; k1 only on the x axis -> x (1 + k1 x^2); p1 only at (0, y) ->
    (0, y + 3 p1 y^2) (OpenCV: yd = y*radial + p1 (r^2 + 2 y^2) + 2 p2 x y)"""
    z = (0.0, 0.0, 0.0, 0.0)
    assert oracle.lens_distort(z, 0.3, -0.2) == (np.float32(0.3), np.float32(-0.2))
    assert oracle.lens_undistort(z, 0.3, -0.2) == (np.float32(0.3), np.float32(-0.2))
    xd, yd = oracle.lens_distort((0.5, 0, 0, 0), 0.5, 0.0)
    assert xd == pytest.approx(0.5 * (1 + 0.5 * 0.25), rel=1e-7) and yd == 0.0
    xd, yd = oracle.lens_distort((0, 0, 0.01, 0), 0.0, 0.4)
    assert xd == 0.0 and yd == pytest.approx(0.4 + 0.01 * 3 * 0.16, rel=1e-6)
    # p2 mirrors p1 on the other axis
    xd, yd = oracle.lens_distort((0, 0, 0, 0.01), 0.4, 0.0)
    assert yd == 0.0 and xd == pytest.approx(0.4 + 0.01 * 3 * 0.16, rel=1e-6)
    # the inverse undoes the forward model well inside the field of view of the reference camera
    lens = (0.12042199820280075, -0.21373499929904938, -0.0021210000850260258, 0.0)
    for x, y in ((0.69, 0.39), (-0.5, 0.1), (0.0, 0.0), (0.2, -0.4)):
        a, b = oracle.lens_distort(lens, x, y)
        ux, uy = oracle.lens_undistort(lens, a, b)
        assert abs(ux - x) < 3e-7 and abs(uy - y) < 3e-7


def test_ground_truth_splat_known_answers(oracle):
    """hand-placed points in front of an axis-aligned camera: projection, square splat footprint, nearest
    wins, white -> transparent (convertToAlpha, Share_Data.hpp:771-784), 180-degree flip (main.cpp:1616)"""
    # engine-frame camera at (0.5, 0.5, 2) looking down -z with x right, y down in the image: c2w columns
    c2w = [1, 0, 0, 0.5, 0, -1, 0, 0.5, 0, 0, -1, 2.0]
    cam = oracle.camera(c2w, 10.0, 10.0, 8.0, 6.0)
    w, h = 16, 12

    def world(e):  # engine (x,y,z) -> world point for scale 1, offset 0 (engine = (wy, wz, wx))
        return [e[2], e[0], e[1]]

    red, green, white = (200, 10, 20), (5, 220, 30), (255, 255, 255)
    pts = [world((0.5, 0.5, 1.0)),          # on the axis, depth 1 -> pixel centre (8, 6)
           world((0.5, 0.5, 0.5)),          # behind it on the same ray, depth 1.5: hidden
           world((0.5 + 0.3, 0.5, 1.0)),    # depth 1, x/z = 0.3 -> u = 11
           world((0.5, 0.5 - 0.2, 1.0))]    # y axis of the camera points to -engine y: v = 6 + 2 = 8
    cols = [red, green, green, white]
    img = oracle.splat_points(pts, cols, 1.0, (0, 0, 0), cam, w, h, point_size=1, flip180=False)
    assert tuple(img[6, 8]) == red + (255,)
    assert tuple(img[6, 11]) == green + (255,)
    assert tuple(img[8, 8]) == white + (0,)  # a white point is as transparent as the background
    assert tuple(img[0, 0]) == (255, 255, 255, 0)
    assert (img[..., 3] == 255).sum() == 2
    # point size 3: a 3x3 footprint centred on the pixel
    img3 = oracle.splat_points(pts[:1], cols[:1], 1.0, (0, 0, 0), cam, w, h, point_size=3, flip180=False)
    ys, xs = np.nonzero(img3[..., 3])
    assert sorted(set(ys)) == [5, 6, 7] and sorted(set(xs)) == [7, 8, 9] and len(ys) == 9
    # flip: rotate by 180 degrees
    f = oracle.splat_points(pts, cols, 1.0, (0, 0, 0), cam, w, h, point_size=1, flip180=True)
    assert np.array_equal(f, img[::-1, ::-1])
    # behind the camera: nothing
    assert oracle.splat_points([world((0.5, 0.5, 3.0))], [red], 1.0, (0, 0, 0), cam, w, h, 1, False)[..., 3].sum() == 0


def test_binary16_rounding_fast_path_equals_the_integer_definition(oracle):
    """orc_d2h (one rounding from double to binary16 -- how the oracle models fp16 mul / fma) has a hardware fast
    path; it must equal the integer-arithmetic definition on every half, every midpoint between adjacent halfs
    (the ties) and the doubles next to them, the range ends, and a seeded random set -- and numpy's own
    float64 -> float16 conversion (also a single rounding) on all of them"""
    L = oracle.lib()
    rng = np.random.default_rng(0)
    h = np.arange(0, 0x7C00, dtype=np.uint16).view(np.float16).astype(np.float64)
    mids = (h[:-1] + h[1:]) / 2
    vals = np.concatenate([h, mids, np.nextafter(mids, np.inf), np.nextafter(mids, -np.inf),
                           [65504, 65519.99, 65520, 65520.01, 1e5, np.inf, 6.103515625e-05, 5.96e-8, 2.98e-8, 0.0],
                           rng.uniform(-70000, 70000, 20000), rng.normal(size=20000) * 10.0 ** rng.integers(-9, 5, 20000)])
    vals = np.concatenate([vals, -vals])
    with np.errstate(over="ignore"):
        want = vals.astype(np.float16).view(np.uint16)
    for v, w in zip(vals, want):
        a = L.orc_d2h(float(v))
        assert a == L.orc_d2h_soft(float(v)) == int(w), v


def test_rank_orders_nan_scores_last_in_oracle_and_library(oracle):
    """a diverged ensemble member yields NaN scores: the ranking stays a strict weak order (NaN after every
    number, NaNs by ascending id), identically in the oracle and in the library's host-only prv_rank, and the
    arg-max rule (strict '>', main.cpp:2088-2091) never selects a NaN"""
    from nerf_prv_amd import api

    orc = oracle
    rng = np.random.default_rng(5)
    for trial in range(20):
        n = int(rng.integers(1, 40))
        scores = np.round(rng.normal(size=n), 1)  # ties on purpose
        scores[rng.random(n) < 0.3] = np.nan
        if trial == 0:
            scores[:] = np.nan
        scores[rng.random(n) < 0.1] = -np.inf
        ids = rng.permutation(1000)[:n].astype(np.int32)
        want = orc.rank(scores, ids)
        rec = np.zeros(n, api.RECORD_DTYPE)
        rec["score"] = scores
        got = api.rank_host(rec, ids)
        assert np.array_equal(got, want)
        # the definition, spelled out: numbers by (score desc, id asc), then NaNs by id asc
        num = [i for i in range(n) if not np.isnan(scores[i])]
        nan = [i for i in range(n) if np.isnan(scores[i])]
        ref = sorted(num, key=lambda i: (-scores[i], ids[i])) + sorted(nan, key=lambda i: ids[i])
        assert [int(ids[i]) for i in ref] == want.tolist()
        am = orc.argmax(scores, ids)
        finite_best = [i for i in num if scores[i] > -1e100]
        assert am == (min((ids[i] for i in finite_best if scores[i] == max(scores[j] for j in finite_best)), default=-1))
