import sys, numpy as np
sys.path.insert(0, '.')
from nerf_prv_amd import api
from oracle import oracle as orc
from tests import util
ctx = api.Context(0)
for kw in (util.SMALL, util.SMALL_F2):
    ctx.synthetic_model(0, api.field_desc(**kw), util.SEED_A)
    f = orc.OracleField(orc.desc(**kw), seed=util.SEED_A)
    rng = np.random.default_rng(1)
    pos = rng.random((4096, 3), dtype=np.float32)
    pos[:8] = [[0, 0, 0], [1, 1, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0.5, 0.5, 0.5], [1, 1, 0], [-0.1, 1.2, 0.3]]
    got, want = ctx.debug_encode(0, pos), f.encode(pos)
    bad = np.argwhere(got != want)
    print("mismatches", len(bad), "of", got.size)
    gf, wf = got.view(np.float16).astype(np.float64), want.view(np.float16).astype(np.float64)
    for (i, k) in bad[:12]:
        print(i, k, hex(got[i, k]), hex(want[i, k]), gf[i, k], wf[i, k], pos[i])
    print("cols", np.bincount(bad[:, 1], minlength=32))
    print("max ulp diff", np.abs(got.astype(int) - want.astype(int)).max())
