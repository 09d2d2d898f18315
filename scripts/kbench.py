"""kernel-level microbench of the render path on the standard workload (dev tool, not bench.py)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=64)
ap.add_argument("--size", type=int, default=800)
ap.add_argument("--samples", type=int, default=128)
ap.add_argument("--field", default="256")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--bias", type=float, default=None)
ap.add_argument("--scene", choices=["baseline", "dense"], default="baseline", help="bench.py's scenes: BASELINE.md section 6 literally | opaque object")
ap.add_argument("--step", choices=["fixed", "ngp"], default="fixed")
ap.add_argument("--tag", default="")
ap.add_argument("--width", type=int, default=0)
ap.add_argument("--height", type=int, default=0)
ap.add_argument("--spp", type=int, default=1)
ap.add_argument("--min-t", type=float, default=1e-4)
args = ap.parse_args()
W, H = (args.width or args.size), (args.height or args.size)
import torch
from nerf_prv_amd import api, planner
ctx = api.Context(0)
fd = dict(api.FIELD_256 if args.field == "256" else api.FIELD_HBM if args.field == "hbm" else api.FIELD_512)
if args.scene == "baseline": fd.update(table_amp=0.1, density_bias=0.0)
if args.bias is not None: fd["density_bias"] = args.bias
ctx.synthetic_model(0, api.L.FieldDesc(**fd), 0x5EED0001)
pts = planner.hemisphere_generate(args.views)
fov = 2.0 * np.arctan(0.5 * 1280 / 915.60668945312500)
tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
cams = ctx.cameras_from_matrices(tms, fov, W, H, scale, offset)
opts = api.engine_render_opts(W, H, args.samples if args.step == "fixed" else 0, args.spp, args.min_t)
out = torch.empty((args.views, H, W, 4), dtype=torch.float32, device="cuda")
_, st = ctx.render(0, cams, None, opts, out=out)
ctx.profile_begin()
t0 = time.perf_counter()
clocks = []
for _ in range(args.reps):
    ctx.render(0, cams, None, opts, out=out, want_stats=False)
    if os.environ.get("KBENCH_CLOCKS"):
        clocks.append(ctx.render_clock_ghz())
torch.cuda.synchronize()
if clocks:
    print(args.tag, "clock GHz per rep:", " ".join(f"{c:.3f}" for c in clocks))
dt = (time.perf_counter() - t0) / args.reps
p = ctx.profile_end()
rms, mms = p["render_ms"] / args.reps, p["march_ms"] / args.reps
print(f"{args.tag} scene={args.scene} step={args.step} BPC={os.environ.get('PRV_BLOCKS_PER_CU','auto')} "
      f"eval_exact={st.samples_evaluated} rounds={st.wave_rounds} eval={st.samples_evaluated/1e6:.1f}M ({100*st.samples_evaluated/st.samples_nominal:.2f}% of nominal) "
      f"render={rms:.2f}ms march={mms:.2f}ms wall={dt*1e3:.2f}ms "
      f"kernel_rate={st.samples_evaluated/rms/1e6:.2f} Gsamp/s wall_rate={st.samples_evaluated/dt/1e9:.2f} Gsamp/s "
      f"frac={st.samples_evaluated/rms/1e6*512/8000:.3f} util={st.samples_evaluated/max(1,32*st.wave_rounds):.3f} "
      f"ns/round/SIMD={rms*1e6/max(1,st.wave_rounds/1024):.0f} clock={ctx.render_clock_ghz():.3f}GHz")
