"""dev: per-step wall time of back-to-back scoring rounds (outliers = host / runtime stalls), with the per-kernel timing on"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_prv_amd import api, planner
ctx = api.Context(0)
pts = planner.hemisphere_generate(64)
fov = 2.0 * np.arctan(0.5 * 1280 / 915.60668945312500)
tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
cams = ctx.cameras_from_matrices(tms, fov, 800, 800, scale, offset)
opts = api.render_opts(800, 800, 128, 1, 1e-4)
for name, fd, slots in (("256", api.FIELD_256, (0, 1)), ("512", api.FIELD_512, (2, 3))):
    d = api.L.FieldDesc(**fd)
    ctx.synthetic_model(slots[0], d, 1); ctx.synthetic_model(slots[1], d, 2)
    gt, _ = ctx.render(slots[1], cams, None, opts, want_stats=False)
    rec_dev = torch.zeros(64 * 16, dtype=torch.uint8, device="cuda")
    for prof in (0, 1):
        ts = []
        torch.cuda.synchronize()
        if prof: ctx.profile_begin()
        t_prev = time.perf_counter()
        for it in range(60):
            ctx.score_views(api.L.SCORE_PSNR_COVERAGE, [slots[0]], cams, None, opts, gt=gt, records_dev=rec_dev, to_host=False)
            r = rec_dev.cpu()
            t = time.perf_counter(); ts.append(t - t_prev); t_prev = t
        p = ctx.profile_end() if prof else None
        ts = np.array(ts) * 1e3
        print(f"{name} profiling={prof}: median {np.median(ts):.2f} ms, mean {ts.mean():.2f}, max {ts.max():.2f} at step {ts.argmax()}, >1.5x median: {(ts > 1.5*np.median(ts)).sum()}"
              + (f"; render {p['render_ms']/p['render_launches']:.2f} ms march {p['march_ms']/p['march_launches']:.2f}" if p else ""), flush=True)
    del gt
