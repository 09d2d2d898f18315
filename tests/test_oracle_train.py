"""The training-step oracle (oracle/prv_train.c) checked against itself: the analytic backward pass against
central finite differences of the (exact-mode, rounding-free) forward pass, optimiser bookkeeping, and that
a short run actually fits images of a known field.  CPU only."""
import numpy as np
import pytest

from tests import util

TINY = dict(n_levels=8, n_features=4, log2_hashmap=10, base_res=4, finest_res=24, occ_res=16, density_bias=1.0, table_amp=0.5)


@pytest.fixture(scope="module")
def scene(oracle):
    """8 views of a 'ground truth' field rendered by the oracle marcher, as straight-alpha sRGB bytes"""
    d = oracle.desc(**TINY)
    gt = oracle.OracleField(oracle.desc(**dict(TINY, density_bias=3.0, table_amp=2.0)), seed=util.SEED_B)
    pts = util.fibonacci_hemisphere(8)
    tms, scale, offset = util.hemisphere_transforms(oracle, pts)
    w, h = 24, 16
    intr = {"fl_x": 20.0, "fl_y": 19.5, "cx": 12.3, "cy": 7.8, "w": w, "h": h, "k1": 0.05, "k2": -0.02, "p1": 0.001, "p2": -0.002}
    cams = oracle.cameras_from_dataset(tms, intr, scale, offset)
    imgs = []
    for c in cams:
        rgba, _ = gt.render(c, w, h, 32, 1, 1e-4)
        imgs.append(oracle.quantize_rgba8(rgba, (0, 0, 0, 0)))
    return d, cams, np.stack(imgs), (w, h)


def all_occupied(oracle, field):
    t, m, o = field.params()
    return oracle.OracleField(field.desc, params=(t, m, np.full_like(o, 0xFFFFFFFF)))


@pytest.mark.parametrize("patch,step_mode", [((0, 0), 0), ((4, 2), 0), ((0, 0), 1)])
def test_backward_matches_finite_differences(oracle, scene, patch, step_mode):
    """step_mode 1 = the engine's marcher (dt = sqrt(3)/1024, per-ray random start, here capped at 200 steps per ray)"""
    d, cams, imgs, _ = scene
    init = all_occupied(oracle, oracle.OracleField(d, seed=util.SEED_A))
    opts = oracle.train_opts(n_rays=96, n_samples=24, occ_every=0, patch_w=patch[0], patch_h=patch[1])
    if step_mode == oracle.STEP_NGP:
        opts = oracle.train_opts(n_rays=24, n_samples=200, occ_every=0, step_mode=oracle.STEP_NGP)
    tr = oracle.OracleTrainer(init, opts, cams, imgs, exact=True)
    loss, tg, mg = tr.gradients()
    assert loss > 1e-3 and tr.samples_last > 500
    assert loss == pytest.approx(tr.loss_only(), rel=1e-12)
    table, mlp = tr.master()
    rng = np.random.default_rng(5)

    def fd(arr, i, h):
        keep = arr[i]
        arr[i] = keep + h
        up = tr.loss_only()
        arr[i] = keep - h
        dn = tr.loss_only()
        arr[i] = keep
        return (up - dn) / (2 * h)

    # MLP weights: a few of every layer, the largest-gradient ones included
    for lo, hi in ((0, 2048), (2048, 3072), (3072, 5120), (5120, 9216), (9216, 10240)):
        idx = list(lo + np.argsort(-np.abs(mg[lo:hi]))[:3]) + list(rng.integers(lo, hi, 3))
        for i in idx:
            if abs(mg[i]) < 2e-5:
                continue
            assert fd(mlp, i, 4e-3) == pytest.approx(mg[i], rel=3e-2, abs=1e-6), i
    # hash-table entries with the largest gradients and some random touched ones
    touched = np.flatnonzero(tg)
    assert len(touched) > 100
    idx = list(np.argsort(-np.abs(tg))[:12]) + list(rng.choice(touched, 8))
    n_checked = 0
    for i in idx:
        if abs(tg[i]) < 2e-5:
            continue
        n_checked += 1
        assert fd(table, i, 1e-2) == pytest.approx(tg[i], rel=3e-2, abs=1e-6), i
    assert n_checked >= 8


def test_adam_step_bookkeeping(oracle, scene):
    d, cams, imgs, _ = scene
    init = all_occupied(oracle, oracle.OracleField(d, seed=util.SEED_A))
    opts = oracle.train_opts(n_rays=64, n_samples=24, occ_every=0, l2_reg=0.0)
    tr = oracle.OracleTrainer(init, opts, cams, imgs)
    t0, m0 = (a.copy() for a in tr.master())
    loss, tg, mg = tr.gradients()
    assert tr.step() == pytest.approx(loss, rel=1e-12)
    t1, m1 = tr.master()
    # first Adam step moves every touched weight by lr against the sign of its gradient (|m|/sqrt(v) = 1
    # after bias correction), untouched table entries do not move at all
    g32 = tg.astype(np.float32)
    moved = t1 != t0
    assert np.array_equal(moved, g32 != 0)
    big = np.abs(g32) > 1e-10  # below that sqrt(v) is no longer >> eps = 1e-15 and the step shrinks
    np.testing.assert_allclose((t1 - t0)[big], -opts.lr * np.sign(g32[big]), rtol=2e-3)
    nz = np.abs(mg.astype(np.float32)) > 1e-10
    np.testing.assert_allclose((m1 - m0)[nz], -opts.lr * np.sign(mg[nz]), rtol=2e-3)
    # the fp16 working copy is the rounded master
    tab16, mlp16, _ = tr.params()
    assert np.array_equal(tab16, t1.astype(np.float16).view(np.uint16))
    assert np.array_equal(mlp16, m1.astype(np.float16).view(np.uint16))


def test_rng_and_batch_are_reproducible(oracle, scene):
    d, cams, imgs, _ = scene
    init = all_occupied(oracle, oracle.OracleField(d, seed=util.SEED_A))
    opts = oracle.train_opts(n_rays=64, n_samples=24, occ_every=4)
    a = oracle.OracleTrainer(init, opts, cams, imgs)
    b = oracle.OracleTrainer(init, opts, cams, imgs)
    la = [a.step() for _ in range(5)]
    lb = [b.step() for _ in range(5)]
    assert la == lb and all(np.array_equal(x, y) for x, y in zip(a.params(), b.params()))
    assert oracle.lib().orc_rng_u24(1, 2, 3) == oracle.lib().orc_rng_u24(1, 2, 3) < (1 << 24)
    assert len({oracle.lib().orc_rng_u24(1, 2, i) for i in range(64)}) > 60


def test_patch_batch_rule(oracle, scene):
    """patch_w x patch_h: ray j is pixel j % P of patch j / P; a patch's rays are adjacent pixels of ONE image (rows in
    snake order) under ONE jitter -- observed through the loss: a dataset whose images are constant colours that differ
    per image and an empty density grid make a ray's loss a function of its image (and its own background) alone"""
    d, cams, imgs, (w, h) = scene
    f = oracle.OracleField(d, seed=util.SEED_A)
    t, m, o = f.params()
    empty = oracle.OracleField(f.desc, params=(t, m, np.zeros_like(o)))
    flat = np.zeros_like(imgs)
    for i in range(len(flat)):
        flat[i, ..., :3] = 20 + 25 * i
        flat[i, ..., 3] = 255
    lin = lambda c: np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    seed = 0x7EA10001

    def expected(n_rays, pw, ph):
        P = max(pw, 1) * max(ph, 1)
        tot = 0.0
        for j in range(n_rays):
            img = (oracle.lib().orc_rng_u24(seed, 0, j // P) * len(flat)) >> 24
            tgt = np.float32(lin(np.float32((20 + 25 * img) / 255.0)))
            tot += 3 * float(tgt) ** 2  # black prediction (no samples, random_bg off)
        return tot / (3 * n_rays)

    for pw, ph, n in ((0, 0, 40), (2, 2, 40), (4, 4, 40), (4, 2, 37)):
        tr = oracle.OracleTrainer(empty, oracle.train_opts(n_rays=n, n_samples=8, occ_every=0, random_bg=0, patch_w=pw, patch_h=ph, seed=seed), cams, flat)
        assert tr.loss_only() == pytest.approx(expected(n, pw, ph), rel=1e-5)
    # ... and WHICH pixels: images whose colour is the pixel's own coordinates (r = x, g = y, b = image) make a ray's loss term
    # lin(x)^2 + lin(y)^2 + lin(image)^2; the term of ray n - 1 is 3 n loss(n) - 3 (n - 1) loss(n - 1).  Expected from the rule
    # restated here: origin drawn per patch in [0, W - pw] x [0, H - ph], ray r of a patch at row r // pw, rows walked in snake order
    coord = np.zeros_like(imgs)
    coord[..., 0] = np.arange(w, dtype=np.uint8)[None, None, :]
    coord[..., 1] = np.arange(h, dtype=np.uint8)[None, :, None]
    coord[..., 2] = np.arange(len(coord), dtype=np.uint8)[:, None, None]
    coord[..., 3] = 255
    u24 = oracle.lib().orc_rng_u24
    for pw, ph in ((1, 1), (2, 2), (4, 2), (3, 3), (4, 4), (1, 4)):
        P, n, prev = pw * ph, 2 * pw * ph + 3, 0.0
        for k in range(1, n + 1):
            tr = oracle.OracleTrainer(empty, oracle.train_opts(n_rays=k, n_samples=8, occ_every=0, random_bg=0, target_samples=0, patch_w=pw, patch_h=ph,
                                                               seed=seed), cams, coord)
            tot = tr.loss_only() * 3 * k
            j, q, r = k - 1, (k - 1) // P, (k - 1) % P
            img = (u24(seed, 0, q) * len(coord)) >> 24
            ry = r // pw
            rx = pw - 1 - r % pw if ry & 1 else r % pw
            x = ((u24(seed, 1, q) * (w - pw + 1)) >> 24) + rx
            y = ((u24(seed, 2, q) * (h - ph + 1)) >> 24) + ry
            assert 0 <= x < w and 0 <= y < h
            want = sum(float(lin(np.float32(c / 255.0))) ** 2 for c in (x, y, img))
            assert tot - prev == pytest.approx(want, rel=2e-4, abs=1e-7), (pw, ph, j, x, y, img)
            prev = tot
    # patches never leave the image: a patch as large as the image has one possible origin
    tr = oracle.OracleTrainer(empty, oracle.train_opts(n_rays=w * h, n_samples=8, occ_every=0, random_bg=0, patch_w=w, patch_h=h), cams, flat)
    assert tr.loss_only() > 0
    with pytest.raises(Exception):
        oracle.OracleTrainer(empty, oracle.train_opts(n_rays=4, n_samples=8, patch_w=w + 1, patch_h=1), cams, flat)


def test_training_fits_the_images(oracle, scene):
    """a few hundred steps on 8 views of a known field: loss falls by an order of magnitude and a training
    view renders closer to its image than the initial model did"""
    d, cams, imgs, (w, h) = scene
    init = all_occupied(oracle, oracle.OracleField(oracle.desc(**dict(TINY, table_amp=1e-4)), seed=util.SEED_A))
    opts = oracle.train_opts(n_rays=256, n_samples=24, occ_every=16, occ_sigma_thresh=0.01 * 24 / 3 ** 0.5)
    tr = oracle.OracleTrainer(init, opts, cams, imgs)
    losses = [tr.step() for _ in range(160)]
    assert np.mean(losses[-10:]) < 0.2 * np.mean(losses[:5])

    def view_mse(field):
        rgba, _ = field.render(cams[2], w, h, 24, 1, 1e-4)
        got = oracle.quantize_rgba8(rgba, (0, 0, 0, 1)).astype(np.float64)[..., :3]
        a = imgs[2].astype(np.float64)
        want = a[..., :3] * a[..., 3:4] / 255.0  # the image over black
        return np.mean((got - want) ** 2)

    assert view_mse(tr.field()) < 0.35 * view_mse(init)
    _, _, occ = tr.params()
    bits = np.unpackbits(occ.view(np.uint8)).sum()
    assert 0 < bits < d.occ_res ** 3  # the density grid has carved some empty space and kept some
