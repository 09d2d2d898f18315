"""where does the trained field differ from the ground truth? (dev tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nerf_prv_amd import api, planner
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
thresh = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01 * 128 / 3 ** 0.5
ctx = api.Context(0)
fd = dict(api.FIELD_256)
ctx.synthetic_model(1, api.L.FieldDesc(**fd), 0x5EED0002)
pts = planner.hemisphere_generate(72)
fov = 2.0 * np.arctan(0.5 * 1280 / 915.60668945312500)
tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
W = H = 400
fl = 0.5 * W / np.tan(0.5 * fov)
intr = dict(fl_x=fl, fl_y=fl, cx=W / 2, cy=H / 2, w=W, h=H)
cams = ctx.cameras_from_matrices_intr(tms, intr, scale, offset)
ropts = api.render_opts(W, H, 128, 1, 1e-4, background=(0, 0, 0, 0))
u8, _ = ctx.render_rgba8(1, cams, None, ropts)
gt, _ = ctx.render(1, cams, None, ropts)
train_cams = ctx.cameras_from_matrices_intr(np.asarray(tms)[8:], intr, scale, offset)
d = api.L.FieldDesc(**dict(fd, table_amp=1e-4, density_bias=0.0))
ctx.fresh_model(0, d, 0x1234)
tr = api.Trainer(ctx, 0, train_cams, u8[8:].contiguous(), api.train_opts(n_rays=rays, occ_sigma_thresh=thresh))
for chunk in range(steps // 500):
    tr.steps(500)
    img, st = ctx.render(0, cams, [0, 1], ropts)
    g = gt[:2]
    empty = g[..., 3] < 1e-3
    solid = g[..., 3] > 0.99
    t16, m16, occ = ctx.export_model(0, d)
    tab = t16.view(np.float16).astype(np.float32)
    print(f"step {(chunk+1)*500}: alpha in GT-empty pixels mean {img[..., 3][empty].mean().item():.5f} max {img[..., 3][empty].max().item():.4f}; "
          f"|rgb err| solid {(img[..., :3] - g[..., :3]).abs()[solid].mean().item():.4f} empty {(img[..., :3] - g[..., :3]).abs()[empty].mean().item():.5f}; "
          f"occupied cells {np.unpackbits(occ.view(np.uint8)).sum()} ; table |max| {np.abs(tab).max():.2f} nan {np.isnan(tab).sum()} ; "
          f"evaluated {st.samples_evaluated}")
    p, s = ctx.evaluate_images(img, g.contiguous(), background=(0, 0, 0, 1))
    print("   per-image psnr", p, "ssim", s)
    import torch.nn.functional as F
    def srgb(x):
        return torch.where(x <= 0.0031308, 12.92 * x, 1.055 * x.clamp(min=1e-9) ** (1 / 2.4) - 0.055).clamp(0, 1)
    a, b = srgb(img[..., :3]).mean(-1)[:, None], srgb(g[..., :3]).mean(-1)[:, None]
    k = torch.ones(1, 1, 7, 7, device=a.device) / 49
    mu_a, mu_b = F.conv2d(a, k, padding=3), F.conv2d(b, k, padding=3)
    va, vb = F.conv2d(a * a, k, padding=3) - mu_a ** 2, F.conv2d(b * b, k, padding=3) - mu_b ** 2
    cov = F.conv2d(a * b, k, padding=3) - mu_a * mu_b
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    smap = ((2 * mu_a * mu_b + C1) * (2 * cov + C2)) / ((mu_a ** 2 + mu_b ** 2 + C1) * (va + vb + C2))
    e, so = empty[:, None], solid[:, None]
    print(f"   ssim map: empty {smap[e].mean().item():.3f} (frac {e.float().mean().item():.2f}) solid {smap[so].mean().item():.3f} (frac {so.float().mean().item():.2f}) "
          f"rest {smap[~e & ~so].mean().item():.3f}; local var rendered in empty {va[e].mean().item():.2e}; srgb max in empty {a[e].max().item():.3f}")
