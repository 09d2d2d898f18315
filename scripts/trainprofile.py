"""time and batch size of a training run chunk by chunk, in the planner loop's configuration (dev tool): N views of the
ground-truth field at 1280x720 (the reference camera), a fresh field, n_rays 4096, 2500 steps in chunks of 100;
--members M trains M fields side by side (prv_train_steps_multi), as a scoring round of the loop does"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=5)
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--steps", type=int, default=2500)
ap.add_argument("--chunk", type=int, default=100)
ap.add_argument("--members", type=int, default=5)
ap.add_argument("--patch", default="", help="WxH: rays drawn as patches of adjacent pixels (prv_train_opts.patch_w / patch_h); default: the library's")
ap.add_argument("--tail-single", type=int, default=0, help="after the run: this many single steps, member 0's composited-sample count of each printed (pairs with a PMC pass's last launches)")
ap.add_argument("--eval", action="store_true", help="after the run: PSNR / SSIM of every member on 8 held-out views of the 144-view set")
ap.add_argument("--save-state", help="after the run: store every member's field under this directory")
ap.add_argument("--load-state", help="start from the fields stored there and keep them (learning rate 0): ablation builds time the same batches")
ap.add_argument("--rule", choices=["fixed", "ngp"], default="", help="how a training ray is sampled (prv_train_opts.step_mode); default: the library's")
ap.add_argument("--det", action="store_true", help="prv_train_opts.deterministic")
ap.add_argument("--thresh", type=float, default=0.0, help="prv_train_opts.occ_sigma_thresh (0: the library's)")
ap.add_argument("--occ-every", type=int, default=0, help="prv_train_opts.occ_every (0: the library's)")
args = ap.parse_args()
import torch
from nerf_prv_amd import api, planner
ctx = api.Context(0)
fd = dict(api.FIELD_256)
ctx.synthetic_model(6, api.L.FieldDesc(**fd), 1592590338)
pts = planner.hemisphere_read(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hemisphere", "144.txt"), 144)
tms, scale, offset = planner.hemisphere_transforms(pts[:: 144 // args.views][: args.views], 0.3, 0.1, [1e-10] * 3)
W, H = 1280, 720
intr = dict(fl_x=915.606689453125, fl_y=913.32666015625, cx=647.1453247070312, cy=372.51531982421875, w=W, h=H)
cams = ctx.cameras_from_matrices_intr(tms, intr, scale, offset)
u8, _ = ctx.render_rgba8(6, cams, None, api.engine_render_opts(W, H, 0, 1, 1e-4, background=(0, 0, 0, 0)))
d = api.L.FieldDesc(**dict(fd, table_amp=1e-4, density_bias=0.0))
pk = {}
if args.patch:
    pk = dict(patch_w=int(args.patch.split("x")[0]), patch_h=int(args.patch.split("x")[1]))
if args.rule:
    pk["step_mode"] = 1 if args.rule == "ngp" else 0
if args.det:
    pk["deterministic"] = 1
if args.thresh > 0:
    pk["occ_sigma_thresh"] = args.thresh
if args.occ_every > 0:
    pk["occ_every"] = args.occ_every
trs = []
for e in range(args.members):
    if args.load_state:
        ctx.load_model_file(e, os.path.join(args.load_state, f"member{e}.prvf"))
        trs.append(api.Trainer(ctx, e, cams, u8, api.train_opts(n_rays=args.rays, seed=0x7EA10001 + e, lr=1e-30, l2_reg=0.0, **pk)))
    else:
        ctx.fresh_model(e, d, 0x1234 + e)
        trs.append(api.Trainer(ctx, e, cams, u8, api.train_opts(n_rays=args.rays, seed=0x7EA10001 + e, **pk)))
torch.cuda.synchronize()
t_all = time.perf_counter()
done = 0
while done < args.steps:
    n = min(args.chunk, args.steps - done)
    t0 = time.perf_counter()
    api.train_many(trs, n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    done += n
    info = trs[0].info()
    print(f"steps {done - n:5d}..{done:5d}: {dt / n * 1e3:7.3f} ms per round of {args.members} member-steps; member 0: {info['samples_last']:7d} samples, "
          f"{info['active_rays']:5d} rays in its last batch", flush=True)
if args.tail_single:
    counts = []
    for _ in range(args.tail_single):
        api.train_many(trs, 1)
        counts.append(trs[0].info()["samples_last"])
    print("tail single steps, member 0 composited samples:", " ".join(str(c) for c in counts), f"mean {np.mean(counts):.0f}")
if os.environ.get("STAMP_SUMS"):  # a PRV_TRAIN_ABLATE=48 build: phase time sums of block 0 of member 0's backward launches
    import ctypes
    st = np.zeros(64, np.uint64)
    ctx.lib.prv_train_debug_stamps(trs[0].handle, st.ctypes.data_as(ctypes.c_void_p))
    v = st[32:64].astype(np.float64) / 2400.0 / args.steps  # s_memtime ticks at the shader clock (~2.4 GHz) -> us per launch
    print("bwd phase sums per launch, block 0 (us):", " ".join(f"{i}:{x:.1f}" for i, x in enumerate(v) if x > 0), f"total {v.sum():.1f}")
if args.eval:  # held-out views: rows of the 144-view set the training views skip
    step = 144 // args.views
    held = [i for i in range(step // 2, 144, step)][:8]
    htms, _, _ = planner.hemisphere_transforms(np.asarray(pts)[held], 0.3, 0.1, [1e-10] * 3)
    hcams = ctx.cameras_from_matrices_intr(htms, intr, scale, offset)
    eo = api.engine_render_opts(W, H, 0, 1, 1e-4, background=(0, 0, 0, 1))
    gt_lin, _ = ctx.render(6, hcams, None, api.engine_render_opts(W, H, 0, 1, 1e-4, background=(0, 0, 0, 0)))
    res = [ctx.evaluate(e, hcams, None, eo, gt_lin) for e in range(args.members)]
    o0 = trs[0].opts
    print(f"held-out ({len(held)} views, {W}x{H}) after {args.steps} steps, patch {o0.patch_w}x{o0.patch_h}: PSNR " + " ".join(f"{p:.2f}" for p, _ in res) +
          f" (mean {np.mean([p for p, _ in res]):.2f}); SSIM " + " ".join(f"{q:.4f}" for _, q in res) + f" (mean {np.mean([q for _, q in res]):.4f})")
if args.save_state:
    os.makedirs(args.save_state, exist_ok=True)
    for e in range(args.members):
        ctx.save_model(e, os.path.join(args.save_state, f"member{e}.prvf"))
print(f"{args.steps} steps x {args.members} members in {time.perf_counter() - t_all:.2f} s")
