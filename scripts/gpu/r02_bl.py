"""dev: gradients of one batch through the register-resident bf16-split chain vs the LDS / f32-MFMA chain (same state, same batch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from nerf_prv_amd import api, planner
ctx = api.Context(0)
fd = dict(api.FIELD_256)
ctx.synthetic_model(1, api.L.FieldDesc(**fd), 0x5EED0002)
pts = planner.hemisphere_generate(32)
fov = 2.0 * np.arctan(0.5 * 1280 / 915.60668945312500)
tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
W = H = 200
fl = 0.5 * W / np.tan(0.5 * fov)
cams = ctx.cameras_from_matrices_intr(tms, dict(fl_x=fl, fl_y=fl, cx=W / 2, cy=H / 2, w=W, h=H), scale, offset)
u8, _ = ctx.render_rgba8(1, cams, None, api.render_opts(W, H, 128, 1, 1e-4, background=(0, 0, 0, 0)))
d = api.L.FieldDesc(**dict(fd, table_amp=1e-4, density_bias=0.0))
ctx.fresh_model(0, d, 0x1234)
tr = api.Trainer(ctx, 0, cams, u8, api.train_opts(n_rays=8192))
tr.steps(300)
ctx.save_model(0, "/tmp/cmp_state.prvf")
res = {}
for chain in ("1", "0"):
    os.environ["PRV_TRAIN_REG_CHAIN"] = chain
    ctx.load_model_file(0, "/tmp/cmp_state.prvf")
    t2 = api.Trainer(ctx, 0, cams, u8, api.train_opts(n_rays=8192, target_samples=1 << 30))
    loss, tg, mg = t2.gradients()
    res[chain] = (loss, np.asarray(tg, np.float64), np.asarray(mg, np.float64), t2.info()["samples_last"])
    t2.close()
rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
print("samples", res["1"][3], res["0"][3], "loss", res["1"][0], res["0"][0])
print("table grad rel L2 (chain vs LDS chain):", rel(res["1"][1], res["0"][1]), " mlp grad rel L2:", rel(res["1"][2], res["0"][2]))
