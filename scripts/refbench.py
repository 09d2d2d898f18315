"""the reference's own scoring round: N candidate views at 80x45, spp 16, E ensemble members (dev tool)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=540)
ap.add_argument("--members", type=int, default=5)
ap.add_argument("--method", type=int, default=3)
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--step", choices=["fixed", "ngp", "both"], default="both", help="fixed = 128 uniform samples per ray; ngp = the engine's dt = sqrt(3)/1024 rule (run.py:304)")
args = ap.parse_args()
import torch
from nerf_prv_amd import api, planner
ctx = api.Context(0)
desc = api.L.FieldDesc(**api.FIELD_256)
for e in range(args.members):
    ctx.synthetic_model(e, desc, 0x5EED0001 + e)
pts = planner.hemisphere_generate(args.views)
fov = 2.0 * np.arctan(0.5 * 1280 / 915.60668945312500)
tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
cams = ctx.cameras_from_matrices(tms, fov, 80, 45, scale, offset)
slots = list(range(args.members))
for mode in (["fixed", "ngp"] if args.step == "both" else [args.step]):
    opts = api.engine_render_opts(80, 45, 128 if mode == "fixed" else 0, args.spp, 0.01, background=(0, 0, 0, 1))
    rec, st = ctx.score_views(args.method, slots, cams, None, opts, want_stats=True)
    torch.cuda.synchronize()
    ctx.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        rec, _ = ctx.score_views(args.method, slots, cams, None, opts)
    dt = (time.perf_counter() - t0) / args.reps
    p = ctx.profile_end()
    print(f"step={mode} views={args.views} E={args.members} spp={args.spp} method={args.method}: {dt*1e3:.2f} ms per scoring round "
          f"({args.views/dt:.0f} views/s, {st.samples_evaluated/dt/1e9:.2f} Gsamp/s evaluated, {st.rays/dt/1e9:.2f} Grays/s); "
          f"kernels: render {p['render_ms']/args.reps:.2f} ms in {p['render_launches']//args.reps} launches, "
          f"march {p['march_ms']/args.reps:.2f} ms; slot util {st.samples_evaluated/max(1,st.wave_rounds*32):.3f}, "
          f"{st.samples_evaluated/max(1,st.rays):.1f} evaluated and {st.samples_live/max(1,st.rays):.1f} live samples/ray; "
          f"best view {ctx.argmax(rec, np.arange(args.views))}")
