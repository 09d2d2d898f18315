"""training throughput + quality on a reference-sized problem (dev tool): N views of a synthetic ground-truth
field -> fresh 256^3 field trained in process -> PSNR on held-out views"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=64)
ap.add_argument("--size", type=int, default=400)
ap.add_argument("--steps", type=int, default=500)
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--samples", type=int, default=128)
ap.add_argument("--chunk", type=int, default=100)
ap.add_argument("--patch", default="", help="WxH: rays drawn as patches of adjacent pixels; default: the library's")
ap.add_argument("--members", type=int, default=0, help="also time an ensemble of this many members stepping side by side")
ap.add_argument("--rule", choices=["fixed", "ngp"], default="", help="how a training ray is sampled (prv_train_opts.step_mode); default: the library's")
ap.add_argument("--det", action="store_true", help="prv_train_opts.deterministic")
ap.add_argument("--thresh", type=float, default=0.0, help="prv_train_opts.occ_sigma_thresh (0: the library's)")
ap.add_argument("--occ-every", type=int, default=0, help="prv_train_opts.occ_every (0: the library's)")
ap.add_argument("--eval-rule", choices=["fixed", "ngp"], default="fixed", help="the stepping rule the held-out views are rendered with (ngp: the engine's, what the planner renders candidates with)")
args = ap.parse_args()
pk = dict(patch_w=int(args.patch.split("x")[0]), patch_h=int(args.patch.split("x")[1])) if args.patch else {}
if args.rule:
    pk["step_mode"] = 1 if args.rule == "ngp" else 0
if args.det:
    pk["deterministic"] = 1
if args.thresh > 0:
    pk["occ_sigma_thresh"] = args.thresh
if args.occ_every > 0:
    pk["occ_every"] = args.occ_every
import torch
from nerf_prv_amd import api, planner
ctx = api.Context(0)
fd = dict(api.FIELD_256)
ctx.synthetic_model(1, api.L.FieldDesc(**fd), 0x5EED0002)  # ground truth object
n_all = args.views + 8
pts = planner.hemisphere_generate(n_all)
fov = 2.0 * np.arctan(0.5 * 1280 / 915.60668945312500)
tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
W = H = args.size
fl = 0.5 * W / np.tan(0.5 * fov)
intr = dict(fl_x=fl, fl_y=fl, cx=W / 2, cy=H / 2, w=W, h=H)
cams = ctx.cameras_from_matrices_intr(tms, intr, scale, offset)
ropts = api.render_opts(W, H, 128, 1, 1e-4, background=(0, 0, 0, 0))
u8, _ = ctx.render_rgba8(1, cams, None, ropts)          # straight-alpha sRGB bytes, as dataset PNGs would be
gt_lin, _ = ctx.render(1, cams, None, ropts)
train_ids, test_ids = np.arange(8, n_all), np.arange(0, 8)
train_cams = ctx.cameras_from_matrices_intr(np.asarray(tms)[train_ids], intr, scale, offset)
# fresh field: tiny table values, Xavier MLP, every cell occupied
init = dict(fd, table_amp=1e-4, density_bias=0.0)
d = api.L.FieldDesc(**init)
ctx.synthetic_model(0, d, 0x1234)
t, m, o = ctx.export_model(0, d)
ctx.load_model(0, d, t, m, np.full_like(o, 0xFFFFFFFF))
eopts = api.engine_render_opts(W, H, 0, 1, 1e-4, background=(0, 0, 0, 1)) if args.eval_rule == "ngp" else api.render_opts(W, H, 128, 1, 1e-4, background=(0, 0, 0, 1))
p0, s0 = ctx.evaluate(0, cams, test_ids, eopts, gt_lin[test_ids].contiguous())
tr = api.Trainer(ctx, 0, train_cams, u8[train_ids].contiguous(), api.train_opts(n_rays=args.rays, n_samples=1024 if args.rule == "ngp" else args.samples, **pk))
tr.steps(2)  # warm-up (allocations, LDS attribute)
torch.cuda.synchronize()
done, t0, used = 0, time.perf_counter(), 0
while done < args.steps:
    n = min(args.chunk, args.steps - done)
    losses = tr.steps(n)
    done += n
    used += tr.info()["samples_last"] * n
    p, s = ctx.evaluate(0, cams, test_ids, eopts, gt_lin[test_ids].contiguous())
    print(f"step {done:5d}  loss {losses[-1]:.5f}  held-out PSNR {p:.2f} SSIM {s:.3f}  samples/batch {tr.info()['samples_last']}", flush=True)
dt = time.perf_counter() - t0
print(f"{args.steps} steps in {dt:.2f} s incl. evaluation = {args.steps/dt:.1f} steps/s; PSNR {p0:.2f} -> {p:.2f}; "
      f"~{used/dt/1e6:.1f} M used samples/s")
torch.cuda.synchronize()
t0 = time.perf_counter()
tr.steps(args.chunk)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"pure training: {args.chunk/dt:.1f} steps/s ({dt/args.chunk*1e3:.2f} ms/step, {args.rays}-ray cap, rule {args.rule or 'default'}, last batch {tr.info()['samples_last']} samples / {tr.info()['active_rays']} rays = {tr.info()['samples_last']/dt*args.chunk/1e6:.1f} M trained samples/s)")

if args.members > 1:
    trs = []
    for e in range(args.members):
        ctx.fresh_model(e, d, 0x1234 + e)
        trs.append(api.Trainer(ctx, e, train_cams, u8[train_ids].contiguous(), api.train_opts(n_rays=args.rays, n_samples=1024 if args.rule == "ngp" else args.samples, seed=0x7EA10001 + e, **pk)))
    api.train_many(trs, 300)  # past the all-occupied start
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    api.train_many(trs, args.chunk)
    torch.cuda.synchronize()
    dt_many = time.perf_counter() - t0
    t0 = time.perf_counter()
    for t in trs:
        t.steps(args.chunk)
    torch.cuda.synchronize()
    dt_seq = time.perf_counter() - t0
    print(f"ensemble of {args.members}: side by side {dt_many/args.chunk*1e3:.2f} ms per round of {args.members} member-steps "
          f"({args.members*args.chunk/dt_many:.0f} member-steps/s); one after another {dt_seq/args.chunk*1e3:.2f} ms "
          f"({args.members*args.chunk/dt_seq:.0f} member-steps/s)")
