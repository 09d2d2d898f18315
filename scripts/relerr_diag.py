"""dev: distribution of the GPU-vs-oracle pixel error relative to max(|want|, floor), to choose the floor of the
relative 1e-3 pixel bar (tests/util.py).  Rows 380-420 of one 800x800 view per scene."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nerf_prv_amd import api, planner
from oracle import oracle as orc
from tests import util

ctx = api.Context(0)
W = H = 800
pts = planner.hemisphere_generate(8)
tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
cams = ctx.cameras_from_matrices(tms, util.FOV_X, W, H, scale, offset)
ocams = orc.cameras_from_transforms(tms, util.FOV_X, W, H, scale, offset)
scenes = {"256 default": dict(api.FIELD_256), "512 default": dict(api.FIELD_512),
          "256 baseline": dict(api.FIELD_256, table_amp=0.1, density_bias=0.0),
          "256 mid": dict(api.FIELD_256, table_amp=1.0, density_bias=1.0)}
rows = (380, 420)
for name, kw in scenes.items():
    ctx.synthetic_model(0, api.L.FieldDesc(**kw), util.SEED_A)
    f = orc.OracleField(orc.desc(**kw), seed=util.SEED_A)
    for v in (1, 3, 6):
        for min_t in (1e-4, 1e-2):
            img, st = ctx.render(0, cams, [v], api.render_opts(W, H, 128, 1, min_t))
            got = img[0].cpu().numpy()[rows[0]:rows[1]]
            want, ne = f.render(ocams[v], W, H, 128, 1, min_t, threads=16, rows=rows)
            want = want[rows[0]:rows[1]]
            err = np.abs(got.astype(np.float64) - want)
            line = f"{name:13s} v{v} minT={min_t:g} max|want|={np.abs(want).max():.3f} abs_err_max={err.max():.2e}"
            for floor in (0.1, 0.05, 0.02, 0.01, 0.001):
                line += f" | floor {floor:g}: {(err / np.maximum(np.abs(want), floor)).max():.2e}"
            nz = np.abs(want) > 1e-6
            line += f" | pure rel (|want|>1e-6): {(err[nz] / np.abs(want[nz])).max():.2e}, px {nz.sum()}"
            print(line, flush=True)
