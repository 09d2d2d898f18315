"""dev: degenerate trainer batches -- no occupied cell at all (zero samples per step), and a single ray"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nerf_prv_amd import api, planner
ctx = api.Context(0)
kw = dict(n_levels=8, n_features=4, log2_hashmap=12, base_res=4, finest_res=32, occ_res=16, density_bias=0.0, table_amp=0.1)
d = api.field_desc(**kw)
ctx.synthetic_model(0, d, 7)
t, m, o = ctx.export_model(0, d)
pts = planner.hemisphere_generate(6)
tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
W, H = 24, 16
fl = 20.0
cams = ctx.cameras_from_matrices_intr(tms, dict(fl_x=fl, fl_y=fl, cx=W / 2, cy=H / 2, w=W, h=H), scale, offset)
imgs = ctx.torch.from_numpy(np.random.default_rng(1).integers(0, 256, (6, H, W, 4), dtype=np.uint8))
for name, occ, rays in (("empty occupancy", np.zeros_like(o), 256), ("one ray", np.full_like(o, 0xFFFFFFFF), 1), ("empty, one ray", np.zeros_like(o), 1)):
    ctx.load_model(0, d, t, m, occ)
    tr = api.Trainer(ctx, 0, cams, imgs, api.train_opts(n_rays=rays, n_samples=32, occ_every=0))
    losses = tr.steps(5)
    loss, tg, mg = tr.gradients()
    print(name, "losses", [float(f"{x:.4g}") for x in losses], "samples", tr.info()["samples_last"], "finite", bool(np.isfinite(tg).all() and np.isfinite(mg).all()))
    tr.close()
print("ok")
