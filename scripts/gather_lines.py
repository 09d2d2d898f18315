#!/usr/bin/env python3
"""How many distinct 64-byte lines does one sample's hash-grid gather touch, level by level -- as laid out today, and
under the brick-tiled layouts the round-2 verdict proposed?  (analysis tool, numpy only, no GPU)

The render kernel's wave marches 32-ray cohorts (an 8x4 pixel block) in lockstep: lanes = adjacent rays at the same
sample index.  For the bench workload's cameras this script generates such cohorts, takes their samples inside the
synthetic object, computes the 8 corner addresses per level under each layout and counts
  * lines per sample, alone                      (what one lane asks for),
  * lines per sample, amortised over the cohort  (distinct lines of the 32 lanes x 8 corners / 32: what reaches L1/L2).
Layouts:  as built  -- dense levels: power-of-two strides, x fastest; hashed: ((x ^ y p1 ^ z p2) & (T-1)) entries;
          brick     -- dense levels in 4x2x2 (F=2) / 2x2x2 (F=4) vertex bricks of 64 B (value-preserving, buildable);
          brickhash -- hashed levels indexed by a hash of the BRICK coordinate, corners of a brick contiguous.
                       NOT value-preserving: it is a different hash function, i.e. a different field -- no permutation of
                       the canonical table can produce it, because canonical neighbours in y/z are scattered by construction.
  python scripts/gather_lines.py [256|512] [cohort shape: 8x4 (as built) | 32x1 | 1x32 | 16x2 | auto]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerf_prv_amd import api, planner  # noqa: E402  (host-side helpers only; nothing here needs the GPU)

P1, P2 = np.uint64(2654435761), np.uint64(805459861)
SPHERES = np.array([[0.50, 0.50, 0.50, 0.35], [0.80, 0.50, 0.62, 0.13], [0.36, 0.80, 0.45, 0.11], [0.40, 0.24, 0.78, 0.10]])


def levels(kw):
    L, T = kw["n_levels"], 1 << kw["log2_hashmap"]
    g = np.exp((np.log(kw["finest_res"]) - np.log(kw["base_res"])) / (L - 1))
    out = []
    for l in range(L):
        s = kw["base_res"] * g ** l - 1.0
        s = round(s) if abs(s - round(s)) < 1e-9 else s
        res = int(np.ceil(s)) + 1
        out.append((float(s), res, res ** 3 > T))
    return out, T


def cohort_samples(n_views=6, tiles_per_view=40, S=128, W=800, seed=1, shape=(8, 4)):
    """positions [cohort, lane(32), 3] of lockstep samples inside the object; shape = the cohort's pixel block (w, h), or
    "auto": 32x1 rows or 1x32 columns, whichever runs along the image direction of the table's x axis in that view"""
    rng = np.random.default_rng(seed)
    pts = planner.hemisphere_generate(64)
    tms, scale, offset = planner.hemisphere_transforms(pts, 0.3, 0.1, [1e-10] * 3)
    fov = 2.0 * np.arctan(0.5 * 1280 / 915.60668945312500)
    f = 0.5 * W / np.tan(0.5 * fov)
    out = []
    for v in rng.choice(64, n_views, replace=False):
        tm = np.asarray(tms[v], np.float64)
        m = tm[:3].copy()
        m[:, 1] *= -1
        m[:, 2] *= -1
        m[:, 3] = m[:, 3] * scale + offset
        c2w = m[[1, 2, 0], :]  # nerf -> engine frame
        cw, ch = shape if shape != "auto" else ((32, 1) if abs(np.linalg.inv(c2w[:, :3])[0, 0]) >= abs(np.linalg.inv(c2w[:, :3])[1, 0]) else (1, 32))
        for _ in range(tiles_per_view):
            x0, y0 = rng.integers(250, 550), rng.integers(250, 550)
            px, py = np.meshgrid(np.arange(cw) + x0, np.arange(ch) + y0)
            d = np.stack([(px.ravel() + 0.5 - W / 2) / f, (py.ravel() + 0.5 - W / 2) / f, np.ones(32)], 1) @ c2w[:, :3].T
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            o = c2w[:, 3]
            inv = 1.0 / d
            t0 = np.max(np.minimum((0 - o) * inv, (1 - o) * inv), 1)
            t1 = np.min(np.maximum((0 - o) * inv, (1 - o) * inv), 1)
            for i in range(S):
                t = t0 + (i + 0.5) * (t1 - t0) / S
                p = o + t[:, None] * d
                inside = (np.linalg.norm(p[:, None, :] - SPHERES[None, :, :3], axis=2) < SPHERES[None, :, 3]).any(1)
                if inside.all() and (t1 > t0).all():
                    out.append(p)
    return np.clip(np.array(out), 0.0, 1.0)


def corner_addresses(pos, scale, res, hashed, T, ebytes, layout):
    """byte addresses [..., 8] of the 8 corners of each position's cell on one level"""
    q = pos * scale + 0.5
    c0 = np.floor(q).astype(np.int64)
    c1 = np.minimum(c0 + 1, res - 1)
    addr = []
    for c in range(8):
        x = np.where(c & 1, c1[..., 0], c0[..., 0]).astype(np.uint64)
        y = np.where(c & 2, c1[..., 1], c0[..., 1]).astype(np.uint64)
        z = np.where(c & 4, c1[..., 2], c0[..., 2]).astype(np.uint64)
        if not hashed:
            sx = int(np.ceil(np.log2(res + 1)))
            if layout == "brick":
                bx = 4 if ebytes == 4 else 2  # a 64-byte brick: 4x2x2 entries of 4 B, or 2x2x2 of 8 B
                nbx, nb = (1 << sx) // bx, (1 << sx) // 2
                brick = (x // bx) + nbx * ((y // 2) + nb * (z // 2))
                a = brick * 64 + ((x % bx) + bx * ((y % 2) + 2 * (z % 2))) * ebytes
            else:
                a = (x + (y << sx) + (z << (2 * sx))) * ebytes
        elif layout == "brickhash":
            bx = 4 if ebytes == 4 else 2
            nent = 64 // ebytes
            brick = ((x // bx) ^ ((y // 2) * P1) ^ ((z // 2) * P2)) & np.uint64(T // nent - 1)
            a = brick * 64 + ((x % bx) + bx * ((y % 2) + 2 * (z % 2))) * ebytes
        else:
            a = ((x ^ (y * P1) ^ (z * P2)) & np.uint64(T - 1)) * ebytes
        addr.append(a)
    return np.stack(addr, -1)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "512"
    kw = dict(api.FIELD_256 if which == "256" else api.FIELD_512)
    lv, T = levels(kw)
    ebytes = kw["n_features"] * 2
    shape = sys.argv[2] if len(sys.argv) > 2 else "8x4"
    pos = cohort_samples(shape="auto" if shape == "auto" else tuple(int(v) for v in shape.split("x")))
    print(f"field {which}^3: L={kw['n_levels']} F={kw['n_features']} T=2^{kw['log2_hashmap']}, {len(pos)} cohort-samples of 32 lanes, cohort = {shape} pixels")
    print(f"{'level':>5} {'res':>5} {'kind':>6} | {'as built: alone':>15} {'cohort':>7} | {'brick: alone':>12} {'cohort':>7}")
    tot = {"built": [0.0, 0.0], "brick": [0.0, 0.0]}
    for l, (s, res, hashed) in enumerate(lv):
        row = []
        for layout in ("built", "brickhash" if hashed else "brick"):
            a = corner_addresses(pos, s, res, hashed, T, ebytes, layout) >> np.uint64(6)  # 64-byte lines
            alone = np.mean([len(np.unique(x)) for x in a.reshape(-1, 8)[:: max(1, a.size // 8 // 4000)]])
            cohort = np.mean([len(np.unique(x)) for x in a.reshape(len(pos), -1)]) / 32.0
            row += [alone, cohort]
            k = "built" if layout == "built" else "brick"
            tot[k][0] += alone
            tot[k][1] += cohort
        print(f"{l:5d} {res:5d} {'hashed' if hashed else 'dense':>6} | {row[0]:15.2f} {row[1]:7.2f} | {row[2]:12.2f} {row[3]:7.2f}"
              + ("   (brick = a DIFFERENT hash: not value-preserving)" if hashed else ""))
    print(f"{'sum':>18} | {tot['built'][0]:15.2f} {tot['built'][1]:7.2f} | {tot['brick'][0]:12.2f} {tot['brick'][1]:7.2f}   lines per sample")


if __name__ == "__main__":
    main()
