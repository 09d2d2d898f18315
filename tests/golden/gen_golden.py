#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from an INDEPENDENT numpy restatement.

Why this exists: the reference (psc0628/NeRF-PRV) ships no tests or golden vectors, cannot be
compiled or imported here, and reaches the field arithmetic through an unvendored, unpinned
instant-ngp.  So the C oracle (oracle/prv_oracle.c) is pinned three ways instead:
  1. against the reference's own DATA (Hemisphere/N.txt view sets),
  2. against hand-derived known answers (tests/test_oracle_known_answers.py),
  3. against THIS script: a second restatement of the same published algorithms written in
     numpy, sharing no code with the C oracle.  Two independent restatements agreeing is the
     strongest pin available; it is still NOT parity with the reference binary.

Run from the repo root:  python tests/golden/gen_golden.py
Writes golden_cameras.json, golden_scores.json, golden_field.json, golden_render.json, golden_lens.json,
golden_render_ngp.json (name files on the command line to regenerate only those).
Nothing here reads /root/reference.
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32

# ---------------------------------------------------------------- cameras (View_Space.hpp:67-140, main.cpp:1626-1641)


def view_pose_np(init_pos, center):
    view = np.asarray(init_pos, np.float64)
    obj = np.asarray(center, np.float64)

    def normalized(v):
        n2 = float(v @ v)
        return v / math.sqrt(n2) if n2 > 0 else v

    Z = normalized(obj - view)
    X = normalized(np.cross(Z, view))
    Y = normalized(np.cross(Z, X))
    T = np.eye(4)
    T[:3, 3] = -view
    R = np.eye(4)
    R[:3, 0], R[:3, 1], R[:3, 2] = X, Y, Z

    def axes(M):
        with np.errstate(invalid="ignore"):
            return math_acos((M @ np.array([0, 1, 0, 1.0]))[1]), math_acos((M @ np.array([1, 0, 0, 1.0]))[0])

    Rz_min = np.eye(4)
    min_y, min_x = axes(np.linalg.inv(R) @ T)
    i = 5.0
    while i < 360:
        a = i * math.acos(-1.0) / 180.0
        c, s = math.cos(a), math.sin(a)
        Rz = np.eye(4)
        Rz[0, 0], Rz[0, 1], Rz[1, 0], Rz[1, 1] = c, -s, s, c
        cy, cx = axes(np.linalg.inv(R @ Rz) @ T)
        if cy < min_y:
            Rz_min, min_y, min_x = Rz, cy, cx
        elif abs(cy - min_y) < 1e-6 and cx < min_x:
            Rz_min, min_y, min_x = Rz, cy, cx
        i += 5
    return np.linalg.inv(R @ Rz_min) @ T


def math_acos(x):
    return math.acos(x) if -1.0 <= x <= 1.0 else float("nan")


def transform_matrix_np(pose):
    P = np.array([[0, 0, 1, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]])
    P1 = np.diag([1.0, -1.0, -1.0, 1.0])
    return P @ np.linalg.inv(pose) @ P1


def gen_cameras():
    out = {}
    center = np.array([1e-10, 1e-10, 1e-10])
    for n in (5, 64):
        pts = np.loadtxt(os.path.join(HERE, "hemisphere", f"{n}.txt"))
        pt_norm = np.linalg.norm(pts[0])
        pos = [p * (1.0 / pt_norm * 0.3) + center for p in pts if p[2] >= 0]
        tms = [transform_matrix_np(view_pose_np(p, center)) for p in pos]
        out[str(n)] = {"radius": 0.3, "center": center.tolist(), "positions": [p.tolist() for p in pos],
                       "transform_matrix": [t.tolist() for t in tms]}
    # assumed instant-ngp conversion, on a generic matrix
    tm = np.array(out["5"]["transform_matrix"][2])
    scale, offset = 5.0, [0.5 + 1e-10] * 3
    m = tm[:3].copy()
    m[:, 1] *= -1
    m[:, 2] *= -1
    m[:, 3] = m[:, 3] * scale + np.array(offset)
    out["nerf_to_ngp"] = {"tm": tm.tolist(), "scale": scale, "offset": offset,
                          "c2w": m[[1, 2, 0]].astype(np.float32).astype(np.float64).tolist()}
    # bounding sphere of a small cloud (View_Space.hpp:534-548)
    rng = np.random.default_rng(7)
    cloud = rng.normal(size=(200, 3)) * [0.03, 0.02, 0.05] + [0.01, -0.02, 0.0]
    c = cloud.mean(axis=0)
    out["bbx"] = {"cloud": cloud.tolist(), "center": c.tolist(),
                  "predicted_size": float(np.linalg.norm(cloud - c, axis=1).max() * 17.0 / 16.0)}
    return out


# ---------------------------------------------------------------- scores (main.cpp:2053-2086, 2113-2150; run.py:257-263)


def ensemble_rgb_np(imgs):
    v = np.stack(imgs).astype(np.float64)[..., :3]  # E,P,3
    var = ((v - v.mean(axis=0)) ** 2).mean(axis=0)
    return float(np.log(var[var > 1e-10]).sum())


def ensemble_rgbdensity_np(imgs):
    v = np.stack(imgs).astype(np.float64)
    var = ((v[..., :3] - v[..., :3].mean(axis=0)) ** 2).mean(axis=0)
    md = (v[..., 3] / 255.0).mean(axis=0)
    return float((var.sum(axis=-1) / 3.0 + (1.0 - md) ** 2).sum())


def srgb_np(x):
    x = np.asarray(x, np.float32)
    with np.errstate(invalid="ignore"):
        hi = f32(1.055) * np.power(x, f32(0.41666666)) - f32(0.055)
    return np.where(x <= f32(0.0031308), f32(12.92) * x, hi).astype(np.float32)


def psnr_coverage_np(img, gt):
    a = np.clip(srgb_np(img[..., :3]), 0, 1).astype(np.float64)
    r = np.clip(srgb_np(gt[..., :3]), 0, 1).astype(np.float64)
    mse = ((a - r) ** 2).mean()
    return float(-10.0 * math.log10(mse)), float(img[..., 3].astype(np.float64).mean())


def ssim_np(img, gt):
    """mean SSIM as instant-ngp's scripts/common.py computes it (ASSUMED recipe, see oracle orc_ssim)"""
    k = np.array([0.120078, 0.233881, 0.292082, 0.233881, 0.120078], np.float32)

    def lum(x):
        c = np.clip(srgb_np(x[..., :3]), 0, 1).astype(np.float32)
        c = np.power(np.maximum(c, f32(0)), f32(0.4545454545)).astype(np.float32)
        return (f32(0.2126) * c[..., 0] + f32(0.7152) * c[..., 1] + f32(0.0722) * c[..., 2]).astype(np.float32)

    def blur(a):
        h, w = a.shape
        t = sum(k[j] * a[:, j:w - 4 + j] for j in range(5)).astype(np.float32)
        return sum(k[i] * t[i:h - 4 + i, :] for i in range(5)).astype(np.float32)

    a, b = lum(img), lum(gt)
    mA, mB = blur(a), blur(b)
    sA, sB, sAB = blur(a * a) - mA * mA, blur(b * b) - mB * mB, blur(a * b) - mA * mB
    c1, c2 = f32(0.01) ** 2, f32(0.03) ** 2
    m = ((f32(2) * mA * mB + c1) / (mA * mA + mB * mB + c1)) * ((f32(2) * sAB + c2) / (sA + sB + c2))
    return float(m.astype(np.float64).mean())


def quantize_np(rgba, bg):
    rgba = np.asarray(rgba, np.float32)
    bg = np.asarray(bg, np.float32)
    c = rgba + (f32(1.0) - rgba[..., 3:4]) * bg
    a = c[..., 3:4]
    with np.errstate(divide="ignore", invalid="ignore"):
        rgb = np.where(a != 0, c[..., :3] / a, c[..., :3]).astype(np.float32)
    rgb = np.clip(srgb_np(rgb), 0, 1)
    out = np.concatenate([rgb, np.clip(a, 0, 1)], axis=-1).astype(np.float32)
    return (out * f32(255.0) + f32(0.5)).astype(np.uint8)


def gen_scores():
    rng = np.random.default_rng(11)
    out = {"ensemble": []}
    for E in (2, 5):
        base = rng.integers(0, 256, (16 * 12, 4), dtype=np.uint8)
        imgs = []
        for _ in range(E):
            noise = rng.integers(-4, 5, base.shape)
            noise[rng.random(base.shape) < 0.6] = 0
            imgs.append(np.clip(base.astype(int) + noise, 0, 255).astype(np.uint8))
        out["ensemble"].append({"E": E, "seed_note": "default_rng(11) stream", "images": [i.tolist() for i in imgs],
                                "rgb": ensemble_rgb_np(imgs), "rgbdensity": ensemble_rgbdensity_np(imgs)})
    img = rng.random((24, 24, 4)).astype(np.float32) * 0.9
    gt = np.clip(img + rng.normal(size=img.shape).astype(np.float32) * 0.05, 0, 1).astype(np.float32)
    p, c = psnr_coverage_np(img, gt)
    out["psnr"] = {"img": img.tolist(), "gt": gt.tolist(), "psnr": p, "coverage": c}
    opaque_img, opaque_gt = img.copy(), gt.copy()
    opaque_img[..., 3] = 1
    opaque_gt[..., 3] = 1
    out["ssim"] = {"value": ssim_np(opaque_img, opaque_gt), "self": ssim_np(opaque_img, opaque_img)}
    q_in = rng.random((64, 4)).astype(np.float32)
    q_in[:4] = [[0, 0, 0, 0], [1, 1, 1, 1], [0.001, 0.002, 0.003, 0.5], [0.2, 0.1, 0.05, 0.25]]
    out["quantize"] = {"rgba": q_in.tolist(), "bg_opaque": quantize_np(q_in, [0, 0, 0, 1]).tolist(),
                       "bg_clear": quantize_np(q_in, [0, 0, 0, 0]).tolist()}
    scores = [3.0, 7.5, 7.5, -1.0, 7.5, 0.0]
    ids = [4, 9, 2, 7, 5, 1]
    order = [i for _, i in sorted(zip([-s for s in scores], ids))]
    out["rank"] = {"scores": scores, "ids": ids, "order": order}
    return out


# ---------------------------------------------------------------- field (published instant-ngp algorithm)

M64 = (1 << 64) - 1


def mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def rng_sym(seed, stream, i, amp):
    h = mix64((seed + (stream + 1) * 0xD1B54A32D192ED03 + i * 0x9E3779B97F4A7C15) & M64)
    u = h >> 40
    v = f32(f32(u) * f32(1.0 / 8388608.0) - f32(1.0))
    return f32(v * f32(amp))


def levels_np(d):
    b = math.exp((math.log(d["finest_res"]) - math.log(d["base_res"])) / (d["n_levels"] - 1))
    T = 1 << d["log2_hashmap"]
    out, off = [], 0
    for l in range(d["n_levels"]):
        s = d["base_res"] * b ** l - 1.0
        if abs(s - round(s)) < 1e-9:
            s = float(round(s))
        res = int(math.ceil(s)) + 1
        dense = res ** 3
        hashed = dense > T
        size = T if hashed else (dense + 7) & ~7
        out.append(dict(scale=f32(s), res=res, offset=off, size=size, hashed=hashed))
        off += size
    return out, off


KIN, KOUT = [32, 64, 32, 64, 64], [64, 16, 64, 64, 16]
SPHERES = [(0.50, 0.50, 0.50, 0.35), (0.80, 0.50, 0.62, 0.13), (0.36, 0.80, 0.45, 0.11), (0.40, 0.24, 0.78, 0.10)]


class FieldNP:
    def __init__(self, d, seed):
        self.d = d
        self.levels, total = levels_np(d)
        F = d["n_features"]
        self.table = np.array([rng_sym(seed, 0, i, d["table_amp"]) for i in range(total * F)], np.float32).astype(np.float16)
        self.W = []
        for l in range(5):
            amp = f32(math.sqrt(f32(6.0) / f32(KIN[l] + KOUT[l])))
            amp = np.sqrt(f32(6.0) / f32(KIN[l] + KOUT[l])).astype(np.float32)
            w = np.array([rng_sym(seed, l + 1, i, amp) for i in range(KIN[l] * KOUT[l])], np.float32).astype(np.float16)
            self.W.append(w.reshape(KIN[l], KOUT[l]))
        R = d["occ_res"]
        g = (np.arange(R, dtype=np.float32) + f32(0.5)) * f32(1.0 / R)
        occ = np.zeros((R, R, R), bool)  # [z,y,x]
        for (cx, cy, cz, r) in SPHERES:
            dx = (g - f32(cx))[None, None, :]
            dy = (g - f32(cy))[None, :, None]
            dz = (g - f32(cz))[:, None, None]
            # fmaf(dx,dx,fmaf(dy,dy,dz*dz)) -- evaluated in float64 then rounded: identical unless a tie
            d2 = (dx.astype(np.float64) ** 2 + (dy.astype(np.float64) ** 2 + (dz * dz).astype(np.float64)).astype(np.float32).astype(np.float64)).astype(np.float32)
            occ |= d2 <= f32(r) * f32(r)
        self.occ = occ

    def occupied(self, p):
        R = self.d["occ_res"]
        c = [min(int(f32(min(max(f32(p[a]), f32(0)), f32(1))) * f32(R)), R - 1) for a in range(3)]
        return bool(self.occ[c[2], c[1], c[0]])

    def encode(self, p):
        F = self.d["n_features"]
        p = [min(max(f32(x), f32(0)), f32(1)) for x in p]
        feat = np.zeros(32, np.float16)
        for l, L in enumerate(self.levels):
            pos = [f32(np.float64(L["scale"]) * np.float64(p[a]) + 0.5) for a in range(3)]  # fma: exact in f64, one rounding
            c0 = [int(math.floor(x)) for x in pos]
            w = [f32(pos[a] - f32(c0[a])) for a in range(3)]
            # blend in binary16 (tiny-cuda-nn style): fp16 weights, fp16 products, fp16 fma chain
            h = np.float16
            wh = [(h(f32(1.0) - w[a]), h(w[a])) for a in range(3)]
            acc = [h(0)] * F
            for c in range(8):
                cc = [min(c0[a] + ((c >> a) & 1), L["res"] - 1) for a in range(3)]
                weight = h(np.float32(h(np.float32(wh[0][c & 1]) * np.float32(wh[1][(c >> 1) & 1]))) * np.float32(wh[2][c >> 2]))
                if L["hashed"]:
                    idx = (cc[0] ^ ((cc[1] * 2654435761) & 0xFFFFFFFF) ^ ((cc[2] * 805459861) & 0xFFFFFFFF)) & (L["size"] - 1)
                else:
                    idx = cc[0] + L["res"] * (cc[1] + L["res"] * cc[2])
                e = self.table[(L["offset"] + idx) * F:(L["offset"] + idx) * F + F]
                for k in range(F):  # fp16 fma: exact in float64, ONE rounding to binary16
                    acc[k] = (np.float64(weight) * np.float64(e[k]) + np.float64(acc[k])).astype(np.float16)
            feat[l * F:(l + 1) * F] = np.array(acc, np.float16)
        return feat

    @staticmethod
    def sh4(d):
        x, y, z = (f32(v) for v in d)
        xy, xz, yz, x2, y2, z2 = x * y, x * z, y * z, x * x, y * y, z * z
        return np.array([
            f32(0.28209479177387814), f32(-0.48860251190291987) * y, f32(0.48860251190291987) * z,
            f32(-0.48860251190291987) * x, f32(1.0925484305920792) * xy, f32(-1.0925484305920792) * yz,
            f32(0.94617469575755997) * z2 - f32(0.31539156525251999), f32(-1.0925484305920792) * xz,
            f32(0.54627421529603959) * x2 - f32(0.54627421529603959) * y2,
            (f32(0.59004358992664352) * y) * (f32(-3.0) * x2 + y2), (f32(2.8906114426405538) * xy) * z,
            (f32(0.45704579946446572) * y) * (f32(1.0) - f32(5.0) * z2),
            (f32(0.3731763325901154) * z) * (f32(5.0) * z2 - f32(3.0)),
            (f32(0.45704579946446572) * x) * (f32(1.0) - f32(5.0) * z2), (f32(1.4453057213202769) * z) * (x2 - y2),
            (f32(0.59004358992664352) * x) * (-x2 + f32(3.0) * y2)], np.float32)

    def eval(self, p, d):
        def layer(W, x):
            return (x.astype(np.float64) @ W.astype(np.float64)).astype(np.float32)

        def relu16(v):
            return np.maximum(v, 0).astype(np.float16)

        x = self.encode(p)
        h = relu16(layer(self.W[0], x))
        od = layer(self.W[1], h)
        sigma = np.exp(f32(od[0] + f32(self.d["density_bias"])), dtype=np.float32)
        rin = np.concatenate([od.astype(np.float16), self.sh4(d).astype(np.float16)])
        h = relu16(layer(self.W[2], rin))
        h = relu16(layer(self.W[3], h))
        orr = layer(self.W[4], h)
        rgb = (f32(1.0) / (f32(1.0) + np.exp(-orr[:3], dtype=np.float32))).astype(np.float32)
        return x, sigma, rgb, od, orr


def raygen_np(c2w, fx, fy, cx, cy, px, py, ox=0.5, oy=0.5):
    dx = f32((f32(f32(px) + f32(ox)) - f32(cx)) / f32(fx))
    dy = f32((f32(f32(py) + f32(oy)) - f32(cy)) / f32(fy))
    m = np.asarray(c2w, np.float32).reshape(3, 4)
    v = [f32(np.float64(m[r, 0]) * np.float64(dx) + np.float64(f32(np.float64(m[r, 1]) * np.float64(dy) + np.float64(m[r, 2]))))
         for r in range(3)]
    n2 = f32(np.float64(v[0]) * np.float64(v[0]) + np.float64(f32(np.float64(v[1]) * np.float64(v[1]) + np.float64(f32(v[2] * v[2])))))
    inv = f32(f32(1.0) / np.sqrt(n2, dtype=np.float32))
    return m[:, 3].copy(), np.array([f32(x * inv) for x in v], np.float32)


def fma32(a, b, c):
    """float32 fma: the product is exact in float64, so this is one rounding (up to a 2^-29 double-rounding tie)"""
    return f32(np.float64(f32(a)) * np.float64(f32(b)) + np.float64(f32(c)))


def lens_eval_np(L, x, y):
    """OpenCV radial + tangential model on normalised coordinates and its Jacobian (float32, spelled out)"""
    k1, k2, p1, p2 = (f32(v) for v in L)
    two, six = f32(2), f32(6)
    r2 = fma32(x, x, f32(y * y))
    kr = fma32(k2, r2, k1)
    radial = fma32(kr, r2, f32(1))
    dk = f32(two * fma32(f32(two * k2), r2, k1))
    xy = f32(x * y)
    fx = fma32(x, radial, fma32(f32(two * p1), xy, f32(p2 * fma32(f32(two * x), x, r2))))
    fy = fma32(y, radial, fma32(p1, fma32(f32(two * y), y, r2), f32(f32(two * p2) * xy)))
    a = f32(two * fma32(p1, x, f32(p2 * y)))
    j0 = fma32(f32(x * x), dk, fma32(f32(two * p1), y, fma32(f32(six * p2), x, radial)))
    j1 = fma32(xy, dk, a)
    j3 = fma32(f32(y * y), dk, fma32(f32(six * p1), y, fma32(f32(two * p2), x, radial)))
    return fx, fy, (j0, j1, j1, j3)


def lens_undistort_np(L, xd, yd, iters=8):
    xd, yd = f32(xd), f32(yd)
    x, y = xd, yd
    for _ in range(iters):
        fx, fy, J = lens_eval_np(L, x, y)
        ex, ey = f32(fx - xd), f32(fy - yd)
        det = fma32(J[0], J[3], f32(-f32(J[1] * J[2])))
        sx = f32(fma32(J[3], ex, f32(-f32(J[1] * ey))) / det)
        sy = f32(fma32(J[0], ey, f32(-f32(J[2] * ex))) / det)
        x, y = f32(x - sx), f32(y - sy)
    return x, y


def raygen_lens_np(c2w, fx, fy, cx, cy, lens, px, py, ox=0.5, oy=0.5):
    dx = f32((f32(f32(px) + f32(ox)) - f32(cx)) / f32(fx))
    dy = f32((f32(f32(py) + f32(oy)) - f32(cy)) / f32(fy))
    dx, dy = lens_undistort_np(lens, dx, dy)
    m = np.asarray(c2w, np.float32).reshape(3, 4)
    v = [fma32(m[r, 0], dx, fma32(m[r, 1], dy, m[r, 2])) for r in range(3)]
    n2 = fma32(v[0], v[0], fma32(v[1], v[1], f32(v[2] * v[2])))
    inv = f32(f32(1.0) / np.sqrt(n2, dtype=np.float32))
    return m[:, 3].copy(), np.array([f32(x * inv) for x in v], np.float32)


def gen_lens():
    """the reference camera (DefaultConfiguration.yaml:38-49) as the dataset json carries it BY KEY
    (Share_Data.hpp:395-399 + main.cpp:1589-1593: k1, k2, p1, p2 = color_k1, color_k2, color_p1, color_p2;
    k3 is written too but is not a term of the engine's lens): undistortion grid + a few rays.
    lens2 adds a non-zero p2 so every term is exercised."""
    lens = [1.2042199820280075e-01, -2.1373499929904938e-01, -2.1210000850260258e-03, 0.0]
    lens2 = [1.2042199820280075e-01, -2.1373499929904938e-01, -2.1210000850260258e-03, 7.5e-04]
    intr = {"fl_x": 9.1560668945312500e+02, "fl_y": 9.1332666015625000e+02, "cx": 6.4714532470703125e+02,
            "cy": 3.7251531982421875e+02, "w": 1280, "h": 720}
    grid = []
    for L in (lens, lens2):
        for xd in (-0.7, -0.31, 0.0, 0.2, 0.69):
            for yd in (-0.4, 0.0, 0.13, 0.39):
                x, y = lens_undistort_np(L, xd, yd)
                fx, fy, _ = lens_eval_np(L, x, y)
                grid.append({"lens": L, "xd": float(f32(xd)), "yd": float(f32(yd)), "x": float(x), "y": float(y),
                             "back_x": float(fx), "back_y": float(fy)})
    c2w = [1, 0, 0, 0.5, 0, 1, 0, 0.5, 0, 0, -1, 2.0]
    rays = []
    for (px, py) in ((0, 0), (1279, 719), (640, 360), (100, 700), (1200, 15)):
        o, d = raygen_lens_np(c2w, f32(intr["fl_x"]), f32(intr["fl_y"]), f32(intr["cx"]), f32(intr["cy"]), lens2, px, py)
        rays.append({"px": px, "py": py, "o": o.astype(np.float64).tolist(), "d": d.astype(np.float64).tolist()})
    return {"intr": intr, "lens_rays": lens2, "c2w": c2w, "grid": grid, "rays": rays}


def aabb_np(o, d):
    tmin, tmax = f32(0), f32(np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        for a in range(3):
            inv = f32(1.0) / f32(d[a])
            ta, tb = f32((f32(0) - o[a]) * inv), f32((f32(1) - o[a]) * inv)
            tmin = np.fmax(tmin, np.fmin(ta, tb))
            tmax = np.fmin(tmax, np.fmax(ta, tb))
    return f32(tmin), f32(tmax)


def march_np(field, o, d, S, min_T, ngp_step=False):
    """ngp_step: instant-ngp's rule for aabb_scale 1 -- fixed dt = sqrt(3)/1024 from the AABB entry,
    samples at t0 + (i + 1/2) dt while inside the box, at most 1024; returns (pixel, evaluated, live)"""
    t0, t1 = aabb_np(o, d)
    if not t1 > t0:
        return np.zeros(4, np.float32), 0, 0
    dt = f32(np.sqrt(f32(3)) / f32(1024)) if ngp_step else f32((t1 - t0) / f32(S))
    T, rgb, n, live, dead = f32(1), np.zeros(3, np.float32), 0, 0, False
    for i in range(1024 if ngp_step else S):
        t = f32(np.float64(f32(i) + f32(0.5)) * np.float64(dt) + np.float64(t0))
        if ngp_step and not t < t1:
            break
        p = [f32(np.float64(t) * np.float64(d[a]) + np.float64(o[a])) for a in range(3)]
        if not field.occupied(p):
            continue
        live += 1
        if dead:
            continue
        _, sigma, c, _, _ = field.eval(p, d)
        alpha = f32(f32(1) - np.exp(-(f32(sigma * dt)), dtype=np.float32))
        w = f32(alpha * T)
        rgb = np.array([f32(np.float64(w) * np.float64(c[k]) + np.float64(rgb[k])) for k in range(3)], np.float32)
        T = f32(T * f32(f32(1) - alpha))
        n += 1
        if T < f32(min_T):
            dead = True  # the march count (live) keeps running, the compositing is over
    return np.array([rgb[0], rgb[1], rgb[2], f32(1) - T], np.float32), n, live


TINY = dict(n_levels=8, n_features=4, log2_hashmap=9, base_res=4, finest_res=32, occ_res=16, density_bias=3.0,
            table_amp=4.0)
TINY_F2 = dict(n_levels=16, n_features=2, log2_hashmap=9, base_res=4, finest_res=40, occ_res=16, density_bias=2.0,
               table_amp=4.0)


def gen_field():
    out = []
    for name, d, seed in (("F4", TINY, 0x5EED0001), ("F2", TINY_F2, 0x5EED0003)):
        fld = FieldNP(d, seed)
        rng = np.random.default_rng(5)
        pos = rng.random((48, 3)).astype(np.float32)
        pos[:4] = [[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 0.999]]
        dirs = rng.normal(size=(48, 3)).astype(np.float32)
        dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
        rec = {"name": name, "desc": d, "seed": seed, "pos": pos.tolist(), "dir": dirs.tolist(), "feat_bits": [],
               "sigma": [], "rgb": [], "dens_out": [], "rgb_out": [], "occupied": [],
               "levels": [[float(L["scale"]), L["res"], L["offset"], L["size"], int(L["hashed"])] for L in fld.levels],
               "table_head_bits": fld.table[:64].view(np.uint16).tolist(),
               "mlp_head_bits": [w.reshape(-1)[:16].view(np.uint16).tolist() for w in fld.W],
               "occ_count": int(fld.occ.sum())}
        for p, dd in zip(pos, dirs):
            x, s, c, od, orr = fld.eval(p, dd)
            rec["feat_bits"].append(x.view(np.uint16).tolist())
            rec["sigma"].append(float(s))
            rec["rgb"].append(c.astype(np.float64).tolist())
            rec["dens_out"].append(od.astype(np.float64).tolist())
            rec["rgb_out"].append(orr.astype(np.float64).tolist())
            rec["occupied"].append(int(fld.occupied(p)))
        out.append(rec)
    return out


def gen_render():
    d, seed = TINY, 0x5EED0001
    fld = FieldNP(d, seed)
    # one camera looking at the cube centre from (0.5, 0.5, 2.0) along -z, 12x12 pixels
    c2w = [1, 0, 0, 0.5, 0, 1, 0, 0.5, 0, 0, -1, 2.0]
    w = h = 12
    fx = fy = f32(0.5 * w / math.tan(0.5 * 0.8))
    img = np.zeros((h, w, 4), np.float32)
    n_eval = n_live = 0
    rays = []
    for y in range(h):
        for x in range(w):
            o, dd = raygen_np(c2w, fx, fy, f32(w / 2), f32(h / 2), x, y)
            px, n, nl = march_np(fld, o, dd, 32, 1e-4)
            img[y, x] = px
            n_eval += n
            n_live += nl
            if (x, y) in ((0, 0), (5, 6), (11, 3)):
                t0, t1 = aabb_np(o, dd)
                rays.append({"px": x, "py": y, "o": o.astype(np.float64).tolist(), "d": dd.astype(np.float64).tolist(),
                             "t0": float(t0), "t1": float(t1)})
    return {"desc": d, "seed": seed, "c2w": c2w, "fx": float(fx), "w": w, "h": h, "samples": 32, "min_T": 1e-4,
            "image": img.astype(np.float64).tolist(), "n_evaluated": n_eval, "n_live": n_live, "rays": rays}


def gen_render_ngp():
    """the same tiny scene through instant-ngp's stepping rule (dt = sqrt(3)/1024, no sample cap), from an oblique
    camera so that rays cross the box at every length up to the diagonal; min_T 0.01 = the engine's default
    (run.py:235 lowers it to 1e-4 for the evaluation views only)"""
    d, seed = dict(TINY, density_bias=3.0), 0x5EED0001
    fld = FieldNP(d, seed)
    c2w = [0.8, -0.36, -0.48, 1.3, 0.6, 0.48, 0.64, -0.55, 0.0, -0.8, 0.6, -0.2]  # rotation (columns orthonormal), looks at the cube
    w = h = 10
    fx = fy = f32(0.5 * w / math.tan(0.5 * 0.7))
    img = np.zeros((h, w, 4), np.float32)
    n_eval = n_live = 0
    per_ray = []
    for y in range(h):
        for x in range(w):
            o, dd = raygen_np(c2w, fx, fy, f32(w / 2), f32(h / 2), x, y)
            px, n, nl = march_np(fld, o, dd, 0, 1e-2, ngp_step=True)
            img[y, x] = px
            n_eval += n
            n_live += nl
            per_ray.append([n, nl])
    return {"desc": d, "seed": seed, "c2w": c2w, "fx": float(fx), "w": w, "h": h, "min_T": 1e-2,
            "image": img.astype(np.float64).tolist(), "n_evaluated": n_eval, "n_live": n_live, "per_ray": per_ray}


def main():
    for name, fn in (("golden_cameras.json", gen_cameras), ("golden_scores.json", gen_scores),
                     ("golden_field.json", gen_field), ("golden_render.json", gen_render),
                     ("golden_lens.json", gen_lens), ("golden_render_ngp.json", gen_render_ngp)):
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        data = fn()
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(data, f)
        print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")


if __name__ == "__main__":
    main()
