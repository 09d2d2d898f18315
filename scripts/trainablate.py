"""time one training batch (forward + composite + backward, no update) on a FIXED trained state (dev tool):
   --save P : train 500 steps with the current build, store the field;   --load P : time gradients() on it"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_prv_amd import api, planner
ap = argparse.ArgumentParser()
ap.add_argument("--save"); ap.add_argument("--load"); ap.add_argument("--rays", type=int, default=4096); ap.add_argument("--tag", default="")
args = ap.parse_args()
ctx = api.Context(0)
fd = dict(api.FIELD_256)
ctx.synthetic_model(1, api.L.FieldDesc(**fd), 0x5EED0002)
pts = planner.hemisphere_generate(64)
fov = 2.0 * np.arctan(0.5 * 1280 / 915.60668945312500)
tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
W = H = 400
fl = 0.5 * W / np.tan(0.5 * fov)
cams = ctx.cameras_from_matrices_intr(tms, dict(fl_x=fl, fl_y=fl, cx=W / 2, cy=H / 2, w=W, h=H), scale, offset)
u8, _ = ctx.render_rgba8(1, cams, None, api.render_opts(W, H, 128, 1, 1e-4, background=(0, 0, 0, 0)))
d = api.L.FieldDesc(**dict(fd, table_amp=1e-4, density_bias=0.0))
if args.save:
    ctx.fresh_model(0, d, 0x1234)
    tr = api.Trainer(ctx, 0, cams, u8, api.train_opts(n_rays=args.rays))
    tr.steps(500)
    ctx.save_model(0, args.save)
    print("saved; samples/batch", tr.info()["samples_last"])
else:
    ctx.load_model_file(0, args.load)
    # a huge sample target: the first batch casts all n_rays (the adaptive count would start at target / n_samples)
    tr = api.Trainer(ctx, 0, cams, u8, api.train_opts(n_rays=args.rays, target_samples=1 << 30))
    tr.gradients()
    torch.cuda.synchronize()
    ctx.lib.prv_train_gradients.argtypes  # keep
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        ctx._chk(ctx.lib.prv_train_gradients(tr.handle, None, None, None))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{args.tag}: {dt*1e3:.3f} ms per batch (fwd + composite + bwd + grad clear), samples/batch {tr.info()['samples_last']}")
    st = np.zeros(64, np.uint64)
    ctx.lib.prv_train_debug_stamps(tr.handle, st.ctypes.data_as(__import__("ctypes").c_void_p))
    if st.any() and os.environ.get("STAMP_SUMS"):
        v = st[32:64].astype(np.float64) / 2400.0 / (n + 1)  # us per launch, block 0, summed over its tiles
        print("bwd phase sums per launch (us):", " ".join(f"{i}:{x:.1f}" for i, x in enumerate(v) if x > 0), f"total {v.sum():.1f}")
    elif st.any():
        for name, base in (("fwd", 0), ("bwd", 32)):
            v = st[base:base + 32].astype(np.int64)
            nz = np.flatnonzero(v)
            if len(nz) > 1:
                d = np.diff(v[nz]) / 2400.0  # s_memtime ticks at the shader clock (~2.4 GHz) -> microseconds
                print(name, "stamps", [int(x) for x in nz], "deltas us:", " ".join(f"{x:.2f}" for x in d), f"total {d.sum():.1f}")
