"""one full NBV iteration at the reference's own sizes (BASELINE config 5 analogue; dev tool): train a 5-member
ensemble for 2500 steps each on the views chosen so far (1280x720 images), then render + score the remaining
candidates of the 540-view set at 80x45, spp 16 (EnsembleRGBDensity), arg-max"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--chosen", type=int, default=10)
ap.add_argument("--members", type=int, default=5)
ap.add_argument("--steps", type=int, default=2500)
ap.add_argument("--views", type=int, default=540)
args = ap.parse_args()
import torch
from nerf_prv_amd import api, planner
ctx = api.Context(0)
fd = dict(api.FIELD_256)
ctx.synthetic_model(7, api.L.FieldDesc(**fd), 0x5EED0002)  # ground truth
pts = planner.hemisphere_generate(args.views)
tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
tms = np.asarray(tms)
K = dict(fl_x=915.60668945312500, fl_y=913.32666015625, cx=647.14532470703125, cy=372.51531982421875, w=1280, h=720,
         k1=0.12042199820280075, k2=-0.21373499929904938, p1=-0.0021210000850260258, p2=0.0)
chosen = np.linspace(0, args.views - 1, args.chosen).astype(int)
rest = np.setdiff1d(np.arange(args.views), chosen)
t0 = time.perf_counter()
train_cams = ctx.cameras_from_matrices_intr(tms[chosen], K, scale, offset)
u8, _ = ctx.render_rgba8(7, train_cams, None, api.render_opts(1280, 720, 128, 1, 1e-4, background=(0, 0, 0, 0)))
torch.cuda.synchronize()
t_gt = time.perf_counter() - t0
desc = api.L.FieldDesc(**dict(fd, table_amp=1e-4, density_bias=0.0))
def iteration():
    t0 = time.perf_counter()
    trs = []
    for e in range(args.members):
        ctx.fresh_model(e, desc, 1000 + e)
        trs.append(api.Trainer(ctx, e, train_cams, u8, api.train_opts(seed=0x7EA10001 + e)))
    losses = api.train_many(trs, args.steps)
    for t in trs:
        t.close()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    fov = 2 * np.arctan(0.5 * 1280 / K["fl_x"])
    cand = ctx.cameras_from_matrices(tms[rest], fov, 80, 45, scale, offset)
    rec, _ = ctx.score_views(api.L.SCORE_ENSEMBLE_RGB_DENSITY, list(range(args.members)), cand, None,
                             api.render_opts(80, 45, 128, 16, 0.01, background=(0, 0, 0, 1)))
    best = ctx.argmax(rec, rest.astype(np.int32))
    cand.close()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, best, losses[:, -1]
iteration()  # warm-up (allocations)
tt, ts, best, last = iteration()
print(f"NBV iteration at reference sizes: {args.chosen} chosen views (1280x720), ensemble of {args.members} x {args.steps} steps: "
      f"training {tt:.2f} s; {len(rest)} candidates 80x45 spp16 rendered by every member + scored: {ts*1e3:.1f} ms; "
      f"next view {best}; total {tt+ts:.2f} s (ground-truth images of the chosen views: {t_gt*1e3:.0f} ms once)")
print("final losses", " ".join(f"{x:.2e}" for x in last))
