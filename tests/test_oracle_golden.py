"""The C oracle against the committed golden vectors (independent numpy restatement,
tests/golden/gen_golden.py) and against the reference's view-set data.  CPU only."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


@pytest.mark.parametrize("n", ["5", "64"])
def test_pose_and_transform_matrix(oracle, n):
    g = load("golden_cameras.json")[n]
    pts = np.loadtxt(os.path.join(GOLD, "hemisphere", f"{n}.txt"))
    pos = oracle.view_space(pts, g["radius"], g["center"])
    np.testing.assert_allclose(pos, g["positions"], rtol=0, atol=1e-15)
    for p, want in zip(pos, g["transform_matrix"]):
        tm = oracle.transform_matrix(oracle.view_pose(p, g["center"]))
        # both restatements pick the same 5-degree roll.  The look-at X axis is (object - view) x view
        # with |object| = 1e-10: conditioning ~1e10, so fp64 round-off (1e-16) of two
        # different inverse / rotation formulas shows up at ~1e-7 in the rotation block.
        np.testing.assert_allclose(tm, want, rtol=0, atol=1e-6)
        # properties from the reference code: +Z of the camera looks at the object, and the json
        # translation column is (z, x, y) of the camera position
        np.testing.assert_allclose(tm[:3, 3], [p[2], p[0], p[1]], atol=1e-12)
        fwd = -np.array([tm[1, 2], tm[2, 2], tm[0, 2]])  # undo P and the z flip
        to_obj = (np.array(g["center"]) - p) / np.linalg.norm(np.array(g["center"]) - p)
        np.testing.assert_allclose(fwd, to_obj, atol=1e-9)  # the view direction itself is well conditioned


def test_survey_known_answer_5txt_row0(oracle):
    """5.txt row 0 @ r=0.3, centre 1e-10 -> JSON row 0 = [-0.9719, 0.1057, 0.2101, 0.0630]"""
    pts = np.loadtxt(os.path.join(GOLD, "hemisphere", "5.txt"))
    c = [1e-10] * 3
    pos = oracle.view_space(pts, 0.3, c)
    tm = oracle.transform_matrix(oracle.view_pose(pos[0], c))
    np.testing.assert_allclose(tm[0], [-0.9719, 0.1057, 0.2101, 0.0630], atol=5e-5)


def test_nerf_to_ngp_and_bbx(oracle):
    g = load("golden_cameras.json")
    n = g["nerf_to_ngp"]
    got = oracle.nerf_to_ngp(np.array(n["tm"]), n["scale"], n["offset"])
    np.testing.assert_array_equal(got.astype(np.float64), np.array(n["c2w"]))
    c, s = oracle.bbx(np.array(g["bbx"]["cloud"]))
    np.testing.assert_allclose(c, g["bbx"]["center"], rtol=1e-13)
    np.testing.assert_allclose(s, g["bbx"]["predicted_size"], rtol=1e-13)


def test_scores_against_numpy(oracle):
    g = load("golden_scores.json")
    for case in g["ensemble"]:
        imgs = [np.array(i, np.uint8) for i in case["images"]]
        np.testing.assert_allclose(oracle.score_ensemble_rgb(imgs), case["rgb"], rtol=1e-12)
        np.testing.assert_allclose(oracle.score_ensemble_rgbdensity(imgs), case["rgbdensity"], rtol=1e-12)
    p = g["psnr"]
    psnr, cov = oracle.score_psnr_coverage(np.array(p["img"], np.float32), np.array(p["gt"], np.float32))
    np.testing.assert_allclose(psnr, p["psnr"], rtol=1e-6)  # powf: libm vs numpy
    np.testing.assert_allclose(cov, p["coverage"], rtol=1e-12)
    a = np.array(p["img"], np.float32)
    b = np.array(p["gt"], np.float32)
    a[..., 3] = 1
    b[..., 3] = 1
    np.testing.assert_allclose(oracle.ssim(a, b), g["ssim"]["value"], rtol=1e-5)  # fma vs mul+add, powf
    assert abs(oracle.ssim(a, a) - 1.0) < 1e-6 and abs(g["ssim"]["self"] - 1.0) < 1e-6
    q = g["quantize"]
    rgba = np.array(q["rgba"], np.float32)
    for key, bg in (("bg_opaque", (0, 0, 0, 1)), ("bg_clear", (0, 0, 0, 0))):
        got = oracle.quantize_rgba8(rgba, bg).astype(int)
        want = np.array(q[key], int)
        assert np.abs(got - want).max() <= 1 and (got != want).mean() < 0.02  # powf ulp at a .5 boundary
    r = g["rank"]
    assert oracle.rank(r["scores"], r["ids"]).tolist() == r["order"]


@pytest.mark.parametrize("idx", [0, 1])
def test_field_against_numpy(oracle, idx):
    g = load("golden_field.json")[idx]
    d = oracle.desc(**g["desc"])
    lv, total = oracle.levels(d)
    assert [[float(L.scale), L.res, L.offset, L.size, L.hashed] for L in lv] == g["levels"]
    f = oracle.OracleField(d, seed=g["seed"])
    table, mlp, occ = f.params()
    assert table[:64].tolist() == g["table_head_bits"]  # counter RNG + fp16 rounding
    offs = [0, 2048, 3072, 5120, 9216]
    for l in range(5):
        assert mlp[offs[l]:offs[l] + 16].tolist() == g["mlp_head_bits"][l]
    assert int(np.unpackbits(occ.view(np.uint8)).sum()) == g["occ_count"]
    pos, dirs = np.array(g["pos"], np.float32), np.array(g["dir"], np.float32)
    assert f.encode(pos).tolist() == g["feat_bits"]  # fp16 bit patterns, bit-exact
    out, occd = f.eval(pos, dirs)
    assert occd.tolist() == g["occupied"]
    np.testing.assert_allclose(out[:, 4:20], g["dens_out"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out[:, 20:36], g["rgb_out"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out[:, 0], g["sigma"], rtol=1e-5)
    np.testing.assert_allclose(out[:, 1:4], g["rgb"], rtol=1e-5)


def test_render_against_numpy(oracle):
    g = load("golden_render.json")
    f = oracle.OracleField(oracle.desc(**g["desc"]), seed=g["seed"])
    w, h = g["w"], g["h"]
    cam = oracle.camera(g["c2w"], g["fx"], g["fx"], w / 2, h / 2)
    img, n_eval = f.render(cam, w, h, g["samples"], 1, g["min_T"], threads=2)
    assert n_eval == g["n_evaluated"] and n_eval > 0
    np.testing.assert_allclose(img, np.array(g["image"]), rtol=1e-5, atol=1e-6)
    o, d, t = oracle.raygen(cam, w, h)
    for r in g["rays"]:
        i = r["py"] * w + r["px"]
        np.testing.assert_array_equal(o[i].astype(np.float64), r["o"])
        np.testing.assert_array_equal(d[i].astype(np.float64), r["d"])
        assert float(t[i, 0]) == r["t0"] and float(t[i, 1]) == r["t1"]


def test_ngp_step_render_against_numpy(oracle):
    """instant-ngp's stepping rule (dt = sqrt(3)/1024 from the AABB entry, every step tested against the occupancy grid,
    no sample cap -- what run.py:304 renders with): the C oracle against the independent numpy
    restatement, per-ray evaluated and live (march) counts exactly, pixels to float rounding"""
    g = load("golden_render_ngp.json")
    f = oracle.OracleField(oracle.desc(**g["desc"]), seed=g["seed"])
    w, h = g["w"], g["h"]
    cam = oracle.camera(g["c2w"], g["fx"], g["fx"], w / 2, h / 2)
    img, n_eval = f.render(cam, w, h, 0, 1, g["min_T"], threads=2, step_mode=oracle.STEP_NGP)
    assert n_eval == g["n_evaluated"] and f.march_count(cam, w, h, 0, step_mode=oracle.STEP_NGP) == g["n_live"]
    np.testing.assert_allclose(img, np.array(g["image"]), rtol=1e-5, atol=1e-6)
    per_ray = np.array(g["per_ray"]).reshape(h, w, 2)
    assert per_ray[..., 1].max() > 128  # rays with more live samples than the fixed-S mode's mask holds
    for y in (0, 4, 9):  # row by row: the counts of single rows add up to the golden's
        _, ne = f.render(cam, w, h, 0, 1, g["min_T"], threads=1, rows=(y, y + 1), step_mode=oracle.STEP_NGP)
        assert ne == per_ray[y, :, 0].sum()
        assert f.march_count(cam, w, h, 0, rows=(y, y + 1), step_mode=oracle.STEP_NGP) == per_ray[y, :, 1].sum()
    # the fixed-S march count of the older fixture
    g0 = load("golden_render.json")
    f0 = oracle.OracleField(oracle.desc(**g0["desc"]), seed=g0["seed"])
    cam0 = oracle.camera(g0["c2w"], g0["fx"], g0["fx"], g0["w"] / 2, g0["h"] / 2)
    assert f0.march_count(cam0, g0["w"], g0["h"], g0["samples"]) == g0["n_live"]


def test_lens_model_against_numpy(oracle):
    """inverse OpenCV lens (dataset cameras of run.py:238-247): the C oracle against the independent numpy
    restatement of tests/golden/gen_golden.py -- bit-exact, and the Newton solve really inverts the model"""
    import ctypes as C

    g = load("golden_lens.json")
    for e in g["grid"]:
        assert oracle.lens_undistort(e["lens"], e["xd"], e["yd"]) == (e["x"], e["y"])
        assert oracle.lens_distort(e["lens"], e["x"], e["y"]) == (e["back_x"], e["back_y"])
        assert abs(e["back_x"] - e["xd"]) < 2e-7 and abs(e["back_y"] - e["yd"]) < 2e-7
    k = g["intr"]
    cam = oracle.camera(g["c2w"], k["fl_x"], k["fl_y"], k["cx"], k["cy"], g["lens_rays"])
    for r in g["rays"]:
        o, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
        oracle.lib().orc_raygen(C.byref(cam), r["px"], r["py"], C.c_float(0.5), C.c_float(0.5), oracle._p(o), oracle._p(d))
        np.testing.assert_array_equal(o.astype(np.float64), r["o"])
        np.testing.assert_array_equal(d.astype(np.float64), r["d"])


@pytest.mark.parametrize("n,top", [(5, 1), (64, 48), (144, 62)])
def test_reference_view_sets_properties(n, top):
    """N rows, unit norm, z >= 0, exactly one row at (0,0,1) -- at the surveyed index"""
    pts = np.loadtxt(os.path.join(GOLD, "hemisphere", f"{n}.txt"))
    assert pts.shape == (n, 3)
    np.testing.assert_allclose(np.linalg.norm(pts, axis=1), 1.0, atol=2e-6)
    assert (pts[:, 2] >= 0).all()
    tops = np.where(np.linalg.norm(pts - [0, 0, 1], axis=1) < 1e-6)[0]
    assert tops.tolist() == [top]
