"""GPU parity of the training step (nerf_prv_amd/csrc/prv_train.hip through the C ABI) against the CPU oracle
(oracle/prv_train.c) on the same seeded batch.  Bars: identical ray batch (same RNG, same live samples);
loss and gradients within 1e-3 relative (north_star; f32 atomics reorder the sums, the oracle accumulates in
double); after optimiser steps the fp16 weights agree except for rare one-ulp roundings; a short run fits."""
import numpy as np
import pytest

from nerf_prv_amd import api
from tests import util

pytestmark = pytest.mark.gpu

TINY = dict(n_levels=8, n_features=4, log2_hashmap=10, base_res=4, finest_res=24, occ_res=16, density_bias=1.0, table_amp=0.5)
TINY_F2 = dict(n_levels=16, n_features=2, log2_hashmap=10, base_res=4, finest_res=24, occ_res=16, density_bias=1.0, table_amp=0.5)
INTR = {"fl_x": 20.0, "fl_y": 19.5, "cx": 12.3, "cy": 7.8, "w": 24, "h": 16, "k1": 0.05, "k2": -0.02, "p1": 0.001, "p2": -0.002}


@pytest.fixture(scope="module", params=["F4", "F2"])
def scene(request, ctx, oracle):
    kw = TINY if request.param == "F4" else TINY_F2
    gt = oracle.OracleField(oracle.desc(**dict(kw, density_bias=3.0, table_amp=2.0)), seed=util.SEED_B)
    pts = util.fibonacci_hemisphere(8)
    tms, scale, offset = util.hemisphere_transforms(oracle, pts)
    ocams = oracle.cameras_from_dataset(tms, INTR, scale, offset)
    imgs = np.stack([oracle.quantize_rgba8(gt.render(c, 24, 16, 32, 1, 1e-4)[0], (0, 0, 0, 0)) for c in ocams])
    cams = ctx.cameras_from_matrices_intr(tms, INTR, scale, offset)
    return kw, ocams, cams, imgs


def start(ctx, oracle, scene, seed=util.SEED_A, table_amp=None, **opts):
    """the same initial field (all cells occupied) and the same options on both sides"""
    kw, ocams, cams, imgs = scene
    if table_amp is not None:
        kw = dict(kw, table_amp=table_amp)
    f = oracle.OracleField(oracle.desc(**kw), seed=seed)
    t, m, o = f.params()
    o = np.full_like(o, 0xFFFFFFFF)
    f = oracle.OracleField(f.desc, params=(t, m, o))
    ctx.load_model(3, api.field_desc(**kw), t, m, o)
    base = dict(n_rays=192, n_samples=24, occ_every=0)
    base.update(opts)
    otr = oracle.OracleTrainer(f, oracle.train_opts(**base), ocams, imgs)
    gtr = api.Trainer(ctx, 3, cams, ctx.torch.from_numpy(imgs), api.train_opts(**base))
    return f, otr, gtr


def rel_l2(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


# the two sampling rules of a training ray (prv_train_opts.step_mode): 24 jittered uniform samples between the AABB hits, and the
# engine's own marcher (dt = sqrt(3)/1024, per-ray random start, every step tested, up to 1024 steps: ~600 live per ray in the
# all-occupied start field) -- what upstream trains with behind run.py:188
RULES = {"fixed_s": dict(), "ngp": dict(step_mode=api.L.STEP_NGP, n_samples=1024, n_rays=40)}


@pytest.mark.parametrize("rule", list(RULES))
def test_batch_loss_and_gradients(ctx, oracle, scene, rule):
    f, otr, gtr = start(ctx, oracle, scene, **RULES[rule])
    want_loss, want_tg, want_mg = otr.gradients()
    loss, tg, mg = gtr.gradients()
    assert gtr.info()["samples_last"] == otr.samples_last > 1000  # same rays, same live + used samples
    assert loss == pytest.approx(want_loss, rel=1e-3)
    assert np.array_equal(tg != 0, want_tg.astype(np.float32) != 0) or rel_l2(tg, want_tg) < 1e-3  # same entries touched
    assert rel_l2(mg, want_mg) < 1e-3 and rel_l2(tg, want_tg) < 1e-3
    # element-wise on everything that is not tiny
    for got, want in ((mg, want_mg), (tg, want_tg)):
        big = np.abs(want) > 1e-3 * np.abs(want).max()
        assert big.sum() > 50
        np.testing.assert_allclose(got[big], want[big], rtol=5e-3)
    # a second call gives the same batch again (nothing was updated, gradients were cleared)
    loss2, tg2, mg2 = gtr.gradients()
    assert loss2 == pytest.approx(loss, rel=1e-6) and rel_l2(tg2, tg) < 1e-5 and rel_l2(mg2, mg) < 1e-5


@pytest.mark.parametrize("rule", list(RULES))
def test_optimiser_steps_track_the_oracle(ctx, oracle, scene, rule):
    f, otr, gtr = start(ctx, oracle, scene, **dict(dict(occ_every=4, n_samples=24, occ_sigma_thresh=0.3), **RULES[rule]))
    want = np.array([otr.step() for _ in range(6)])
    got = gtr.steps(6)
    np.testing.assert_allclose(got, want, rtol=2e-3)
    assert gtr.info()["steps"] == 6
    wt, wm = otr.master()
    gt_, gm = gtr.master()
    # Adam's first steps are sign-like (+-lr): a gradient that is ~0 on one side and exactly 0 on the other, or
    # whose sign flips inside the f32 noise, moves by lr on one side only -- tolerate a small fraction
    assert np.mean(np.abs(gm - wm) > 2e-3) < 0.01 and np.mean(np.abs(gt_ - wt) > 2e-3) < 0.01
    assert rel_l2(gm, wm) < 2e-2 and rel_l2(gt_, wt) < 2e-2
    # the slot now holds the trained field: fp16 copies of the masters, refreshed occupancy
    d_p = api.field_desc(**scene[0])
    t16, m16, occ = ctx.export_model(3, d_p)
    assert np.array_equal(m16, gm.astype(np.float16).view(np.uint16)) and np.array_equal(t16, gt_.astype(np.float16).view(np.uint16))
    _, _, want_occ = otr.params()
    diff = np.unpackbits((occ ^ want_occ).view(np.uint8)).sum()
    assert diff <= 0.02 * d_p.occ_res ** 3  # cells whose EMA sits at the threshold may flip
    assert 0 < np.unpackbits(occ.view(np.uint8)).sum()


def test_density_refresh_matches_the_oracle(ctx, oracle, scene):
    f, otr, gtr = start(ctx, oracle, scene, occ_sigma_thresh=2.5)
    otr.refresh_occupancy()
    gtr.refresh_occupancy()
    d_p = api.field_desc(**scene[0])
    _, _, occ = ctx.export_model(3, d_p)
    _, _, want = otr.params()
    n_on = np.unpackbits(want.view(np.uint8)).sum()
    assert 0 < n_on < d_p.occ_res ** 3
    assert np.unpackbits((occ ^ want).view(np.uint8)).sum() <= 4  # sigma within 1e-6 of the threshold


def test_training_fits_and_the_slot_renders_the_result(ctx, oracle, scene):
    kw, ocams, cams, imgs = scene
    f, otr, gtr = start(ctx, oracle, scene, table_amp=1e-4, n_rays=1024, occ_every=16, occ_sigma_thresh=0.01 * 24 / 3 ** 0.5)
    opts = api.render_opts(24, 16, 24, 1, 1e-4, background=(0, 0, 0, 1))
    gt_img = ctx.torch.from_numpy(imgs.astype(np.float32) / 255.0).to(ctx.device)
    lin = ctx.torch.where(gt_img[..., :3] <= 0.04045, gt_img[..., :3] / 12.92, ((gt_img[..., :3] + 0.055) / 1.055) ** 2.4)
    gt_lin = ctx.torch.cat([lin * gt_img[..., 3:4], gt_img[..., 3:4]], dim=-1).contiguous()
    psnr0, _ = ctx.evaluate(3, cams, None, opts, gt_lin)
    losses = gtr.steps(300)
    assert np.mean(losses[-20:]) < 0.15 * np.mean(losses[:5])
    psnr1, _ = ctx.evaluate(3, cams, None, opts, gt_lin)  # the slot renders with the trained weights
    assert psnr1 > psnr0 + 6.0
    # trainer follows a reloaded slot only if the field is the same
    ctx.synthetic_model(3, api.field_desc(**dict(kw, finest_res=32)), 1)
    with pytest.raises(api.PrvError):
        gtr.steps(1)


def test_error_behaviour(ctx, oracle, scene):
    kw, ocams, cams, imgs = scene
    ctx.synthetic_model(3, api.field_desc(**kw), 5)
    with pytest.raises(api.PrvError):
        api.Trainer(ctx, 3, cams, ctx.torch.from_numpy(imgs), api.train_opts(n_samples=129))
    with pytest.raises(api.PrvError):
        api.Trainer(ctx, 3, cams, ctx.torch.from_numpy(imgs), api.train_opts(n_rays=0))
    with pytest.raises(api.PrvError):
        api.Trainer(ctx, 99, cams, ctx.torch.from_numpy(imgs))  # no such slot
    with pytest.raises(ValueError):
        api.Trainer(ctx, 3, cams, ctx.torch.from_numpy(imgs[:3]))


def test_training_images_at_another_size_than_the_dataset(ctx, oracle, scene):
    """images half the dataset size: the trainer scales the intrinsics per axis; same batch as an oracle
    given the rescaled cameras"""
    kw, ocams, cams, imgs = scene
    half = np.ascontiguousarray(imgs[:, ::2, ::2])
    pts = util.fibonacci_hemisphere(8)
    tms, scale, offset = util.hemisphere_transforms(oracle, pts)
    ocams_half = oracle.cameras_from_dataset(tms, INTR, scale, offset, 12, 8)
    f = oracle.OracleField(oracle.desc(**kw), seed=util.SEED_A)
    t, m, o = f.params()
    o = np.full_like(o, 0xFFFFFFFF)
    f = oracle.OracleField(f.desc, params=(t, m, o))
    ctx.load_model(3, api.field_desc(**kw), t, m, o)
    base = dict(n_rays=128, n_samples=24, occ_every=0)
    otr = oracle.OracleTrainer(f, oracle.train_opts(**base), ocams_half, half)
    gtr = api.Trainer(ctx, 3, cams, ctx.torch.from_numpy(half), api.train_opts(**base))
    want_loss, want_tg, want_mg = otr.gradients()
    loss, tg, mg = gtr.gradients()
    assert gtr.info()["samples_last"] == otr.samples_last
    assert loss == pytest.approx(want_loss, rel=1e-3) and rel_l2(mg, want_mg) < 1e-3 and rel_l2(tg, want_tg) < 1e-3


@pytest.mark.parametrize("rule", ["fixed_s", "ngp"])
def test_ensemble_members_step_side_by_side(ctx, oracle, scene, rule):
    """prv_train_steps_multi: members on their own streams give what each gives alone (up to atomics order)"""
    kw, ocams, cams, imgs = scene
    extra = dict(n_samples=24) if rule == "fixed_s" else dict(step_mode=api.L.STEP_NGP, n_samples=1024, target_samples=20000)
    d = api.field_desc(**dict(kw, table_amp=1e-4))
    u8 = ctx.torch.from_numpy(imgs)
    alone = []
    for e in range(3):
        ctx.fresh_model(e, d, 500 + e)
        tr = api.Trainer(ctx, e, cams, u8, api.train_opts(n_rays=256, seed=900 + e, occ_every=8, occ_sigma_thresh=0.1, **extra))
        alone.append(tr.steps(24))
        tr.close()
    trs = []
    for e in range(3):
        ctx.fresh_model(e, d, 500 + e)
        trs.append(api.Trainer(ctx, e, cams, u8, api.train_opts(n_rays=256, seed=900 + e, occ_every=8, occ_sigma_thresh=0.1, **extra)))
    together = api.train_many(trs, 24)
    assert together.shape == (3, 24) and all(t.info()["steps"] == 24 for t in trs)
    np.testing.assert_allclose(together, np.stack(alone), rtol=2e-2)
    np.testing.assert_allclose(together[:, :4], np.stack(alone)[:, :4], rtol=1e-4)  # before the noise compounds
    assert not np.allclose(together[0], together[1])  # members differ (seeds)
    with pytest.raises(api.PrvError):
        api.train_many([trs[0], trs[0]], 1)  # the same slot twice


def test_trainer_refuses_a_slot_that_was_reinstalled_under_it(ctx, oracle, scene):
    """a slot re-installed with the SAME descriptor while a trainer is alive: the trainer's fp32 masters, moments and
    captured graph belong to the old parameters -- stepping it would overwrite the new model without a word.  It
    fails with PRV_E_STATE instead, the new model stays what was installed, and a new trainer works."""
    kw, ocams, cams, imgs = scene
    d = api.field_desc(**dict(kw, table_amp=1e-4))
    u8 = ctx.torch.from_numpy(imgs)
    ctx.fresh_model(3, d, 41)
    tr = api.Trainer(ctx, 3, cams, u8, api.train_opts(n_rays=128, n_samples=24))
    tr.steps(3)
    trained = [a.copy() for a in ctx.export_model(3, d)]
    for reinstall in (lambda: ctx.fresh_model(3, d, 42), lambda: ctx.load_model(3, d, *trained)):
        reinstall()
        installed = [a.copy() for a in ctx.export_model(3, d)]
        with pytest.raises(api.PrvError) as e:
            tr.steps(1)
        assert e.value.code == api.L.PRV_E_STATE and "re-installed" in str(e.value)
        with pytest.raises(api.PrvError):
            tr.gradients()
        assert all(np.array_equal(a, b) for a, b in zip(installed, ctx.export_model(3, d)))  # untouched
    tr.close()
    tr2 = api.Trainer(ctx, 3, cams, u8, api.train_opts(n_rays=128, n_samples=24))
    assert np.isfinite(tr2.steps(2)).all()
    tr2.close()


def test_trainer_outliving_its_context_is_inert(oracle, scene):
    """destroying the context first must not leave a dangling trainer (interpreter shutdown order)"""
    kw, ocams, cams_unused, imgs = scene
    c2 = api.Context(0)
    pts = util.fibonacci_hemisphere(8)
    tms, scale, offset = util.hemisphere_transforms(oracle, pts)
    cams = c2.cameras_from_matrices_intr(tms, INTR, scale, offset)
    c2.fresh_model(0, api.field_desc(**kw), 3)
    tr = api.Trainer(c2, 0, cams, c2.torch.from_numpy(imgs), api.train_opts(n_rays=64, n_samples=24))
    tr.steps(2)
    cams.close()
    lib, handle = c2.lib, tr.handle
    c2.close()  # context goes first
    assert lib.prv_train_steps(handle, 1, None) != 0  # inert: an error code, no crash
    lib.prv_train_destroy(handle)
    tr.handle = None


@pytest.mark.parametrize("n_rays,n_samples", [(37, 5), (1, 1), (130, 128)])
def test_ragged_batch_sizes(ctx, oracle, scene, n_rays, n_samples):
    """ray counts that do not fill a block / a wave, one sample per ray, the full 128: same batch, same loss"""
    f, otr, gtr = start(ctx, oracle, scene, n_rays=n_rays, n_samples=n_samples)
    want_loss, want_tg, want_mg = otr.gradients()
    loss, tg, mg = gtr.gradients()
    assert gtr.info()["samples_last"] == otr.samples_last
    assert loss == pytest.approx(want_loss, rel=1e-3, abs=1e-9)
    if otr.samples_last:
        assert rel_l2(mg, want_mg) < 2e-3 and rel_l2(tg, want_tg) < 2e-3
    else:
        assert not tg.any() and not mg.any()
    got, want = gtr.steps(3), [otr.step() for _ in range(3)]
    np.testing.assert_allclose(got, want, rtol=5e-3, atol=1e-9)


@pytest.mark.parametrize("patch,n_rays,n_samples", [((2, 2), 192, 24), ((4, 4), 192, 24), ((4, 2), 190, 24), ((3, 2), 37, 5),
                                                    ((4, 4), 130, 128), ((16, 1), 50, 24), ((1, 3), 7, 24)])
def test_patch_batches_match_the_oracle(ctx, oracle, scene, patch, n_rays, n_samples):
    """patch mode (prv_train_opts.patch_w x patch_h): the step's rays are patches of adjacent pixels that share a
    jitter and the sample list runs depth step by depth step inside a patch -- the batch (pixels, live samples,
    termination) and its loss / gradients are the oracle's under the same rule; ray counts that leave the last
    patch partial, patches as wide as a row segment, one-sample rays"""
    f, otr, gtr = start(ctx, oracle, scene, n_rays=n_rays, n_samples=n_samples, patch_w=patch[0], patch_h=patch[1])
    want_loss, want_tg, want_mg = otr.gradients()
    loss, tg, mg = gtr.gradients()
    assert gtr.info()["samples_last"] == otr.samples_last > 0
    assert loss == pytest.approx(want_loss, rel=1e-3)
    assert rel_l2(mg, want_mg) < 2e-3 and rel_l2(tg, want_tg) < 2e-3
    assert np.array_equal(tg != 0, want_tg.astype(np.float32) != 0) or rel_l2(tg, want_tg) < 1e-3
    # not the i.i.d. batch: the same options without patches draw other pixels
    _, otr1, _ = start(ctx, oracle, scene, n_rays=n_rays, n_samples=n_samples)
    assert otr1.gradients()[0] != want_loss
    f, otr, gtr = start(ctx, oracle, scene, n_rays=n_rays, n_samples=n_samples, patch_w=patch[0], patch_h=patch[1], occ_every=2,
                        occ_sigma_thresh=0.3)
    got, want = gtr.steps(4), [otr.step() for _ in range(4)]  # across a density-grid refresh: ragged live masks inside a patch
    np.testing.assert_allclose(got[:2], want[:2], rtol=5e-3, atol=1e-9)
    np.testing.assert_allclose(got[2:], want[2:], rtol=2e-2, atol=1e-9)  # (a cell whose EMA sits at the threshold may flip: other live samples)
    wt, wm = otr.master()
    gt_, gm = gtr.master()
    assert rel_l2(gm, wm) < 2e-2 and rel_l2(gt_, wt) < 2e-2


def test_patch_mode_with_the_sample_budget_and_bad_patches(ctx, oracle, scene):
    f, otr, gtr = start(ctx, oracle, scene, n_rays=700, n_samples=24, target_samples=2400, patch_w=4, patch_h=2)
    assert otr.active_rays == gtr.info()["active_rays"] == 100  # 12 whole patches and half of the 13th
    sched_o, sched_g = [], []
    for _ in range(6):
        otr.step()
        gtr.steps(1)
        sched_o.append(otr.active_rays)
        sched_g.append(gtr.info()["active_rays"])
    np.testing.assert_allclose(sched_g, sched_o, rtol=0.03)
    kw, ocams, cams, imgs = scene
    for bad in (dict(patch_w=5, patch_h=4), dict(patch_w=-1, patch_h=2), dict(patch_w=2, patch_h=17), dict(patch_w=1, patch_h=32)):
        with pytest.raises(api.PrvError):
            api.Trainer(ctx, 3, cams, ctx.torch.from_numpy(imgs), api.train_opts(**bad))


def test_rays_that_miss_everything(ctx, oracle, scene):
    """an empty density grid: no sample is live, the loss is the background mismatch alone, nothing moves"""
    kw, ocams, cams, imgs = scene
    f = oracle.OracleField(oracle.desc(**kw), seed=util.SEED_A)
    t, m, o = f.params()
    o = np.zeros_like(o)
    f = oracle.OracleField(f.desc, params=(t, m, o))
    ctx.load_model(3, api.field_desc(**kw), t, m, o)
    for base in (dict(n_rays=200, n_samples=24, occ_every=0, l2_reg=0.0),
                 dict(n_rays=200, step_mode=api.L.STEP_NGP, n_samples=1024, occ_every=0, l2_reg=0.0)):  # both sampling rules
        otr = oracle.OracleTrainer(f, oracle.train_opts(**base), ocams, imgs)
        gtr = api.Trainer(ctx, 3, cams, ctx.torch.from_numpy(imgs), api.train_opts(**base))
        loss, tg, mg = gtr.gradients()
        assert gtr.info()["samples_last"] == otr.samples_last == 0
        assert loss == pytest.approx(otr.gradients()[0], rel=1e-5) and loss > 0 and not tg.any() and not mg.any()
        before = gtr.master()
        gtr.steps(2)
        after = gtr.master()
        assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
        gtr.close()


def test_sample_budget_adapts_the_ray_count(ctx, oracle, scene):
    """target_samples: the first step casts target / n_samples rays, afterwards the count follows
    clamp(target * active / used, active/2, 2*active) within [1, n_rays] -- same schedule as the oracle"""
    f, otr, gtr = start(ctx, oracle, scene, n_rays=700, n_samples=24, target_samples=2400, occ_every=0)
    assert otr.active_rays == gtr.info()["active_rays"] == 100  # 2400 / 24
    want_loss, _, want_mg = otr.gradients()
    loss, _, mg = gtr.gradients()
    assert gtr.info()["samples_last"] == otr.samples_last and loss == pytest.approx(want_loss, rel=1e-3)
    assert rel_l2(mg, want_mg) < 1e-3  # the loss is normalised by the ACTIVE ray count
    sched_o, sched_g = [], []
    for _ in range(8):
        otr.step()
        gtr.steps(1)
        sched_o.append(otr.active_rays)
        sched_g.append(gtr.info()["active_rays"])
    assert sched_o[0] in (200, sched_g[0]) and max(sched_o) <= 700 and sched_o[-1] > 100  # grows towards the budget
    np.testing.assert_allclose(sched_g, sched_o, rtol=0.03)  # +-1 sample at a termination threshold moves it a little
    used = gtr.info()["samples_last"]
    assert 0.4 * 2400 < used < 2.5 * 2400 or gtr.info()["active_rays"] == 700
    # without a budget the count never moves
    f, otr, gtr = start(ctx, oracle, scene, n_rays=300, n_samples=24, target_samples=0, occ_every=0)
    gtr.steps(3)
    assert gtr.info()["active_rays"] == 300 == otr.active_rays


def test_a_trainer_on_parked_buffers_matches_the_oracle(ctx, oracle, scene):
    """a destroyed trainer's stream and device buffers stay with the context and the next trainer of the same sizes takes
    them (prv_train_api.inc: train_buffer): whatever the first one left in them -- moments, gradients, kept activations,
    density EMA, counters, a finished step count -- must not reach the second one"""
    f, otr, gtr = start(ctx, oracle, scene, seed=util.SEED_B, table_amp=2.0, occ_every=2, occ_sigma_thresh=0.3)
    gtr.steps(5)
    gtr.close()  # parks
    f, otr, gtr = start(ctx, oracle, scene, occ_every=4, occ_sigma_thresh=0.3)  # same sizes: the parked buffers and stream
    want_loss, want_tg, want_mg = otr.gradients()
    loss, tg, mg = gtr.gradients()
    assert loss == pytest.approx(want_loss, rel=1e-3) and rel_l2(mg, want_mg) < 1e-3 and rel_l2(tg, want_tg) < 1e-3
    want = np.array([otr.step() for _ in range(6)])
    got = gtr.steps(6)
    np.testing.assert_allclose(got, want, rtol=2e-3)
    assert gtr.info()["steps"] == 6
    wt, wm = otr.master()
    gt_, gm = gtr.master()
    assert rel_l2(gm, wm) < 2e-2 and rel_l2(gt_, wt) < 2e-2
    gtr.close()


def test_default_stream_work_of_another_thread_while_step_graphs_are_captured(ctx, oracle, scene):
    """a trainer's stream owns a hardware queue (created with a CU mask: a BLOCKING stream) -- its step graphs are captured on
    a non-blocking stream of the context's instead, so that legacy-default-stream work of another thread (PyTorch's default
    stream) cannot become an implicit dependency on a capture: neither side may see an error, and the steps still track the
    oracle (this runtime did not raise the error without the capture stream either: the test pins the behaviour, not a bug)"""
    import threading

    t = ctx.torch
    stop, errs = threading.Event(), []

    def worker():
        try:
            t.cuda.set_device(ctx.device)
            x = t.ones(1 << 16, device=ctx.device)
            while not stop.is_set():
                x = x * 1.0001 + 0.0  # kernels on the legacy default stream
                t.cuda.current_stream().query()
            t.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = threading.Thread(target=worker)
    th.start()
    try:
        for k in range(4):  # every new trainer captures its step graphs anew
            f, otr, gtr = start(ctx, oracle, scene, occ_every=2, occ_sigma_thresh=0.3, seed=util.SEED_A + k)
            want = np.array([otr.step() for _ in range(4)])
            got = gtr.steps(4)
            np.testing.assert_allclose(got, want, rtol=2e-3)
            gtr.close()
    finally:
        stop.set()
        th.join()
    assert not errs, errs


def test_trainer_lifecycle_returns_its_memory(ctx, oracle, scene):
    """an NBV loop creates and destroys a trainer per member and iteration: device memory must come back"""
    kw, ocams, cams, imgs = scene
    t = ctx.torch
    u8 = t.from_numpy(imgs).to(ctx.device)
    d = api.field_desc(**kw)

    def cycle():
        ctx.fresh_model(2, d, 9)
        tr = api.Trainer(ctx, 2, cams, u8, api.train_opts(n_rays=2048, n_samples=64))
        tr.steps(3)
        tr.close()

    cycle()
    t.cuda.synchronize()
    free0, _ = t.cuda.mem_get_info()
    for _ in range(25):
        cycle()
    t.cuda.synchronize()
    free1, _ = t.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20  # nothing accumulates (a trainer of this size holds ~60 MB)


def test_loss_slices_when_the_rays_outnumber_the_backward_grid(ctx, oracle, scene, monkeypatch):
    """the step's loss and used-sample count are summed per 1024-ray slice by the blocks of the backward launch; with
    more slices than blocks (here 4 slices on 2 blocks; on the chip: n_rays > 2^18 side by side) every block has to take
    several -- loss, used samples and the budget's next ray count must still be the oracle's"""
    monkeypatch.setenv("PRV_TRAIN_BWD_BLOCKS", "2")
    f, otr, gtr = start(ctx, oracle, scene, n_rays=4096, n_samples=8, target_samples=20000, occ_every=0)
    want_loss, _, want_mg = otr.gradients()
    loss, _, mg = gtr.gradients()
    assert gtr.info()["samples_last"] == otr.samples_last > 8000
    assert loss == pytest.approx(want_loss, rel=1e-3) and rel_l2(mg, want_mg) < 1e-3
    want = np.array([otr.step() for _ in range(3)])
    got = gtr.steps(3)  # end_step and the list-ahead ray pass read every slice
    np.testing.assert_allclose(got, want, rtol=2e-3)
    assert gtr.info()["active_rays"] == pytest.approx(otr.active_rays, rel=0.03) and gtr.info()["active_rays"] > 2048


def test_engine_marcher_batches_with_the_sample_budget_and_a_carved_grid(ctx, oracle, scene):
    """PRV_STEP_NGP with target_samples: the first step casts target / n_samples rays, the count then follows the budget
    rule; the density grid is refreshed on the way (rays skip the carved cells); ragged step caps; same schedule, same
    losses as the oracle"""
    f, otr, gtr = start(ctx, oracle, scene, step_mode=api.L.STEP_NGP, n_samples=1024, n_rays=300, target_samples=8192, occ_every=2,
                        occ_sigma_thresh=2.5)
    assert otr.active_rays == gtr.info()["active_rays"] == 8
    sched_o, sched_g, lo, lg = [], [], [], []
    for _ in range(6):
        lo.append(otr.step())
        lg.append(gtr.steps(1)[0])
        sched_o.append(otr.active_rays)
        sched_g.append(gtr.info()["active_rays"])
    np.testing.assert_allclose(lg, lo, rtol=5e-3)
    np.testing.assert_allclose(sched_g, sched_o, rtol=0.05, atol=1)
    assert sched_o[-1] > 8
    # a cap below the ray's length: the ray stops after n_samples steps
    f, otr, gtr = start(ctx, oracle, scene, step_mode=api.L.STEP_NGP, n_samples=70, n_rays=33)
    want_loss, want_tg, want_mg = otr.gradients()
    loss, tg, mg = gtr.gradients()
    assert gtr.info()["samples_last"] == otr.samples_last > 33 * 40
    assert loss == pytest.approx(want_loss, rel=1e-3) and rel_l2(mg, want_mg) < 1e-3 and rel_l2(tg, want_tg) < 1e-3
    with pytest.raises(api.PrvError):  # patches are defined for the fixed rule only
        start(ctx, oracle, scene, step_mode=api.L.STEP_NGP, n_samples=1024, patch_w=2, patch_h=2)
    with pytest.raises(api.PrvError):
        api.Trainer(ctx, 3, scene[2], ctx.torch.from_numpy(scene[3]), api.train_opts(step_mode=api.L.STEP_NGP, n_samples=1025))


@pytest.mark.parametrize("rule", list(RULES))
def test_deterministic_training_is_bit_reproducible(ctx, oracle, scene, rule):
    """prv_train_opts.deterministic (tests): ray batches listed in ray order, the table gradient summed in 64-bit fixed
    point -- two runs give the same masters bit for bit (the default path's f32 atomics do not), and the gradients are
    still the oracle's"""
    opts = dict(dict(occ_every=4, occ_sigma_thresh=0.3, n_rays=200), **RULES[rule])
    runs = []
    for _ in range(2):
        f, otr, gtr = start(ctx, oracle, scene, deterministic=1, **opts)
        loss, tg, mg = gtr.gradients()
        losses = gtr.steps(12)
        runs.append((loss, tg, mg, losses, gtr.master(), ctx.export_model(3, api.field_desc(**scene[0]))))
        gtr.close()
    a, b = runs
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert np.array_equal(a[4][0], b[4][0]) and np.array_equal(a[4][1], b[4][1])
    assert all(np.array_equal(x, y) for x, y in zip(a[5], b[5]))  # fp16 table, MLP and the refreshed occupancy
    want_loss, want_tg, want_mg = start(ctx, oracle, scene, **opts)[1].gradients()
    assert a[0] == pytest.approx(want_loss, rel=1e-3) and rel_l2(a[1], want_tg) < 1e-3 and rel_l2(a[2], want_mg) < 1e-3
    # side by side with the default path: the same training up to the atomics' noise
    f, otr, gtr = start(ctx, oracle, scene, **opts)
    np.testing.assert_allclose(gtr.steps(12)[:4], a[3][:4], rtol=1e-3)


def test_a_batch_beyond_the_sample_list_fails_loudly(ctx, oracle, scene):
    """PRV_STEP_NGP: a step may list at most 2^24 samples (64 x the default budget).  200,000 rays of these cameras through an
    all-occupied cube are well beyond it: the call fails with a message, nothing is written out of bounds, and the trainer keeps working for a
    batch that fits"""
    kw, ocams, cams, imgs = scene
    f, otr, gtr = start(ctx, oracle, scene, n_rays=8)
    gtr.close()
    big = api.Trainer(ctx, 3, cams, ctx.torch.from_numpy(imgs), api.train_opts(step_mode=api.L.STEP_NGP, n_rays=200000, target_samples=0, occ_every=0))
    with pytest.raises(api.PrvError, match="sample list"):
        big.steps(2)
    big.close()
