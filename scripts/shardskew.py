"""dev: per-rank work of the 8-GPU weak-scaling bench (512 views) under contiguous and round-robin sharding,
measured on ONE GPU by running each rank's shard in turn"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_prv_amd import api, planner
world, per = 8, 64
ctx = api.Context(0)
desc = api.L.FieldDesc(**api.FIELD_256)
ctx.synthetic_model(0, desc, 0x5EED0001)
ctx.synthetic_model(1, desc, 0x5EED0002)
n = world * per
pts = planner.hemisphere_generate(n)
fov_x = 2.0 * np.arctan(0.5 * 1280 / 915.60668945312500)
tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
cams = ctx.cameras_from_matrices(tms, fov_x, 800, 800, scale, offset)
opts = api.render_opts(800, 800, 128, 1, 1e-4)
for name in ("contiguous", "round-robin"):
    times, evs = [], []
    for r in range(world):
        ids = np.arange(r * per, (r + 1) * per, dtype=np.int32) if name == "contiguous" else np.arange(r, n, world, dtype=np.int32)
        gt, _ = ctx.render(1, cams, ids, opts, want_stats=False)
        ctx.score_views(5, [0], cams, ids, opts, gt=gt, to_host=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            _, st = ctx.score_views(5, [0], cams, ids, opts, gt=gt, to_host=False, want_stats=True)
        torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / 5 * 1e3); evs.append(st.samples_evaluated / 1e6)
        del gt
    print(name, "ms per rank:", " ".join(f"{t:.2f}" for t in times), "| M samples:", " ".join(f"{e:.0f}" for e in evs),
          f"| max/mean time {max(times) / np.mean(times):.3f}")
