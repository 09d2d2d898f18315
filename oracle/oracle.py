"""ctypes binding of the CPU oracle (oracle/libprv_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py -- never by the product package nerf_prv_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libprv_oracle.so")
MLP_HALFS = 10240
MAX_LEVELS = 16
STEP_FIXED_S, STEP_NGP = 0, 1  # prv_oracle.h: ORC_STEP_*
NGP_MAX_STEPS = 1024


class FieldDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in "n_levels n_features log2_hashmap base_res finest_res occ_res".split()] + [
        ("density_bias", C.c_float), ("table_amp", C.c_float), ("per_level_scale", C.c_float)]


class Level(C.Structure):
    _fields_ = [("scale", C.c_float)] + [(n, C.c_uint32) for n in "res offset size hashed".split()]


class Field(C.Structure):
    _fields_ = [("desc", FieldDesc), ("levels", Level * MAX_LEVELS), ("total_entries", C.c_uint32),
                ("table", C.POINTER(C.c_uint16)), ("mlp", C.c_uint16 * MLP_HALFS), ("occ", C.POINTER(C.c_uint32)),
                ("mlp_f", C.c_float * MLP_HALFS)]


class Camera(C.Structure):
    _fields_ = [("c2w", C.c_float * 12), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("lens", C.c_float * 4)]


class TrainOpts(C.Structure):
    _fields_ = [("n_rays", C.c_int32), ("n_samples", C.c_int32), ("lr", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("l2_reg", C.c_float), ("min_T", C.c_float),
                ("seed", C.c_uint64), ("random_bg", C.c_int32), ("occ_every", C.c_int32), ("occ_decay", C.c_float),
                ("occ_sigma_thresh", C.c_float), ("target_samples", C.c_int32), ("patch_w", C.c_int32),
                ("patch_h", C.c_int32), ("step_mode", C.c_int32), ("deterministic", C.c_int32)]


TRAIN_DEFAULTS = dict(n_rays=65536, n_samples=1024, lr=1e-2, beta1=0.9, beta2=0.99, eps=1e-15, l2_reg=1e-6, min_T=1e-4,
                      seed=0x7EA10001, random_bg=1, occ_every=16, occ_decay=0.95,
                      occ_sigma_thresh=0.01 * 1024 / 3 ** 0.5, target_samples=1 << 18, patch_w=0, patch_h=0, step_mode=1, deterministic=0)


def train_opts(**kw):
    """the library's defaults (prv_train_default_opts) with overrides, under nerf_prv_amd.api.train_opts' rule: n_samples without
    step_mode selects STEP_FIXED_S, step_mode without n_samples gets that rule's default"""
    o = dict(TRAIN_DEFAULTS, **kw)
    if "step_mode" in kw and "n_samples" not in kw:
        o["n_samples"] = NGP_MAX_STEPS if kw["step_mode"] == STEP_NGP else 128
    if "n_samples" in kw and "step_mode" not in kw:
        o["step_mode"] = STEP_FIXED_S
    if "step_mode" not in kw and max(o["patch_w"], 1) * max(o["patch_h"], 1) > 1:
        o["step_mode"] = STEP_FIXED_S
        if "n_samples" not in kw:
            o["n_samples"] = 128
    return TrainOpts(**o)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        vp, f32p = C.c_void_p, C.POINTER(C.c_float)
        L.orc_f2h.restype, L.orc_f2h.argtypes = C.c_uint16, [C.c_float]
        L.orc_h2f.restype, L.orc_h2f.argtypes = C.c_float, [C.c_uint16]
        L.orc_d2h.restype, L.orc_d2h.argtypes = C.c_uint16, [C.c_double]
        L.orc_d2h_soft.restype, L.orc_d2h_soft.argtypes = C.c_uint16, [C.c_double]
        L.orc_field_levels.argtypes = [C.POINTER(FieldDesc), C.POINTER(Level), C.POINTER(C.c_uint32)]
        L.orc_field_synthetic.restype = C.POINTER(Field)
        L.orc_field_synthetic.argtypes = [C.POINTER(FieldDesc), C.c_uint64]
        L.orc_field_from_params.restype = C.POINTER(Field)
        L.orc_field_from_params.argtypes = [C.POINTER(FieldDesc), vp, vp, vp]
        L.orc_field_free.argtypes = [C.POINTER(Field)]
        L.orc_encode.argtypes = [C.POINTER(Field), vp, vp]
        L.orc_sh4.argtypes = [vp, vp]
        L.orc_eval.argtypes = [C.POINTER(Field), vp, vp, f32p, vp, vp]
        L.orc_occupied.argtypes = [C.POINTER(Field), vp]
        L.orc_view_pose.argtypes = [vp, vp, vp]
        L.orc_transform_matrix.argtypes = [vp, vp]
        L.orc_view_space.argtypes = [vp, C.c_int, C.c_double, vp, vp]
        L.orc_bbx.argtypes = [vp, C.c_int, vp, vp]
        L.orc_nerf_to_ngp.argtypes = [vp, C.c_double, vp, vp]
        L.orc_rs2_project.argtypes = [vp, vp, C.c_int, vp]
        L.orc_rs2_deproject.argtypes = [vp, vp, C.c_int, vp, C.c_float]
        L.orc_spp_offset.argtypes = [C.c_int, f32p, f32p]
        L.orc_raygen.argtypes = [C.POINTER(Camera), C.c_int, C.c_int, C.c_float, C.c_float, vp, vp]
        L.orc_ray_aabb.argtypes = [vp, vp, f32p, f32p]
        L.orc_rng_u24.restype, L.orc_rng_u24.argtypes = C.c_uint32, [C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_train_create.restype = vp
        L.orc_train_create.argtypes = [C.POINTER(Field), C.POINTER(TrainOpts), vp, C.c_int, C.c_int, C.c_int, vp, C.c_int]
        L.orc_train_free.argtypes = [vp]
        for n in ("orc_train_step", "orc_train_loss_only"):
            getattr(L, n).restype, getattr(L, n).argtypes = C.c_double, [vp]
        L.orc_train_gradients.restype, L.orc_train_gradients.argtypes = C.c_double, [vp, vp, vp]
        L.orc_train_refresh_occupancy.argtypes = [vp]
        L.orc_train_field.restype, L.orc_train_field.argtypes = C.POINTER(Field), [vp]
        L.orc_train_steps_done.restype, L.orc_train_steps_done.argtypes = C.c_uint32, [vp]
        L.orc_train_samples_last.restype, L.orc_train_samples_last.argtypes = C.c_uint64, [vp]
        L.orc_train_active_rays.restype, L.orc_train_active_rays.argtypes = C.c_uint32, [vp]
        L.orc_train_master_table.restype, L.orc_train_master_table.argtypes = C.POINTER(C.c_float), [vp]
        L.orc_train_master_mlp.restype, L.orc_train_master_mlp.argtypes = C.POINTER(C.c_float), [vp]
        L.orc_train_table_size.restype, L.orc_train_table_size.argtypes = C.c_size_t, [vp]
        L.orc_splat_points.argtypes = [vp, vp, C.c_size_t, C.c_float, vp, C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int, vp]
        L.orc_lens_distort.argtypes = [C.c_float * 4, C.c_float, C.c_float, f32p, f32p]
        L.orc_lens_undistort.argtypes = [C.c_float * 4, f32p, f32p]
        L.orc_render.argtypes = [C.POINTER(Field), C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp,
                                 C.POINTER(C.c_uint64), C.c_int]
        L.orc_render_rows.argtypes = [C.POINTER(Field), C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_float, vp, C.POINTER(C.c_uint64), C.c_int]
        L.orc_render_rows_mode.argtypes = [C.POINTER(Field), C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_float, vp, C.POINTER(C.c_uint64), C.c_int]
        L.orc_march_count_rows.restype = C.c_uint64
        L.orc_march_count_rows.argtypes = [C.POINTER(Field), C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int]
        L.orc_linear_to_srgb.restype, L.orc_linear_to_srgb.argtypes = C.c_float, [C.c_float]
        L.orc_quantize_rgba8.argtypes = [vp, C.c_size_t, vp, vp]
        L.orc_score_ensemble_rgb.restype = C.c_double
        L.orc_score_ensemble_rgb.argtypes = [vp, C.c_int, C.c_size_t]
        L.orc_score_ensemble_rgbdensity.restype = C.c_double
        L.orc_score_ensemble_rgbdensity.argtypes = [vp, C.c_int, C.c_size_t]
        L.orc_score_psnr_coverage.argtypes = [vp, vp, C.c_size_t, vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_score_view.argtypes = [vp, vp, C.c_size_t, vp, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]
        L.orc_ssim.restype, L.orc_ssim.argtypes = C.c_double, [vp, vp, C.c_int, C.c_int, vp]
        L.orc_rank.argtypes = [vp, vp, C.c_int, vp]
        L.orc_argmax.restype, L.orc_argmax.argtypes = C.c_int, [vp, vp, C.c_int]
        L.orc_first_hit.argtypes = [C.POINTER(Field), vp, vp, C.c_float, vp]
        L.orc_first_hit_rows.argtypes = [C.POINTER(Field), C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_float, vp]
        L.orc_precept.argtypes = [C.POINTER(Field), vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def desc(**kw):
    d = dict(n_levels=8, n_features=4, log2_hashmap=19, base_res=16, finest_res=256, occ_res=128, density_bias=3.0,
             table_amp=4.0)
    d.update(kw)
    return FieldDesc(**d)


def levels(d):
    lv = (Level * MAX_LEVELS)()
    tot = C.c_uint32()
    if lib().orc_field_levels(C.byref(d), lv, C.byref(tot)) != 0:
        raise ValueError("bad field descriptor")
    return [lv[i] for i in range(d.n_levels)], tot.value


class OracleField:
    def __init__(self, d, seed=None, params=None):
        self.desc = d
        if params is None:
            self.ptr = lib().orc_field_synthetic(C.byref(d), C.c_uint64(seed))
        else:
            t, m, o = (np.ascontiguousarray(params[0], np.uint16), np.ascontiguousarray(params[1], np.uint16),
                       np.ascontiguousarray(params[2], np.uint32))
            self.ptr = lib().orc_field_from_params(C.byref(d), _p(t), _p(m), _p(o))
        if not self.ptr:
            raise ValueError("oracle field creation failed")

    def params(self):
        f = self.ptr.contents
        n = f.total_entries * self.desc.n_features
        table = np.ctypeslib.as_array(f.table, shape=(n,)).copy()
        mlp = np.frombuffer(f.mlp, dtype=np.uint16).copy()
        R = self.desc.occ_res
        occ = np.ctypeslib.as_array(f.occ, shape=((R ** 3 + 31) // 32,)).copy()
        return table, mlp, occ

    def encode(self, pos):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        out = np.zeros((len(pos), 32), np.uint16)
        for i in range(len(pos)):
            lib().orc_encode(self.ptr, _p(pos[i]), _p(out[i]))
        return out

    def eval(self, pos, dirs):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.zeros((len(pos), 36), np.float32)
        occ = np.zeros(len(pos), np.int32)
        for i in range(len(pos)):
            s = C.c_float()
            rgb = np.zeros(3, np.float32)
            raw = np.zeros(32, np.float32)
            lib().orc_eval(self.ptr, _p(pos[i]), _p(dirs[i]), C.byref(s), _p(rgb), _p(raw))
            out[i, 0], out[i, 1:4], out[i, 4:] = s.value, rgb, raw
            occ[i] = lib().orc_occupied(self.ptr, _p(pos[i]))
        return out, occ

    def render(self, cam, w, h, n_samples=128, spp=1, min_T=1e-4, threads=8, rows=None, step_mode=STEP_FIXED_S):
        img = np.zeros((h, w, 4), np.float32)
        ne = C.c_uint64()
        y0, y1 = rows if rows else (0, h)
        lib().orc_render_rows_mode(self.ptr, C.byref(cam), w, h, y0, y1, step_mode, n_samples, spp, C.c_float(min_T),
                                   _p(img), C.byref(ne), threads)
        return img, ne.value

    def march_count(self, cam, w, h, n_samples=128, spp=1, threads=8, rows=None, step_mode=STEP_FIXED_S):
        """samples in occupied cells (no field evaluation, no early termination)"""
        y0, y1 = rows if rows else (0, h)
        return int(lib().orc_march_count_rows(self.ptr, C.byref(cam), w, h, y0, y1, step_mode, n_samples, spp, threads))

    def close(self):
        if self.ptr:
            lib().orc_field_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def camera(c2w, fx, fy, cx, cy, lens=(0.0, 0.0, 0.0, 0.0)):
    cam = Camera()
    flat = np.asarray(c2w, np.float32).reshape(12)
    for i in range(12):
        cam.c2w[i] = float(flat[i])
    cam.fx, cam.fy, cam.cx, cam.cy = float(fx), float(fy), float(cx), float(cy)
    for i in range(4):
        cam.lens[i] = float(np.float32(lens[i]))
    return cam


def lens_distort(lens, x, y):
    L = (C.c_float * 4)(*[float(np.float32(v)) for v in lens])
    xd, yd = C.c_float(), C.c_float()
    lib().orc_lens_distort(L, C.c_float(x), C.c_float(y), C.byref(xd), C.byref(yd))
    return xd.value, yd.value


def lens_undistort(lens, xd, yd):
    L = (C.c_float * 4)(*[float(np.float32(v)) for v in lens])
    x, y = C.c_float(xd), C.c_float(yd)
    lib().orc_lens_undistort(L, C.byref(x), C.byref(y))
    return x.value, y.value


def cameras_from_dataset(tms, intr, scale, offset, w=None, h=None):
    """oracle-side equivalent of prv_cameras_from_matrices_intr (+ the per-axis rescale to a render size w,h)"""
    sx = np.float32(w) / np.float32(intr["w"]) if w else np.float32(1)
    sy = np.float32(h) / np.float32(intr["h"]) if h else np.float32(1)
    fx, fy = np.float32(intr["fl_x"]) * sx, np.float32(intr["fl_y"]) * sy
    cx, cy = np.float32(intr["cx"]) * sx, np.float32(intr["cy"]) * sy
    lens = [intr.get(k, 0.0) for k in ("k1", "k2", "p1", "p2")]
    return [camera(nerf_to_ngp(tm, scale, offset), fx, fy, cx, cy, lens) for tm in tms]


def view_pose(init_pos, center):
    a, b = np.ascontiguousarray(init_pos, np.float64), np.ascontiguousarray(center, np.float64)
    out = np.zeros(16, np.float64)
    lib().orc_view_pose(_p(a), _p(b), _p(out))
    return out.reshape(4, 4)


def transform_matrix(pose):
    a = np.ascontiguousarray(pose, np.float64).reshape(16)
    out = np.zeros(16, np.float64)
    lib().orc_transform_matrix(_p(a), _p(out))
    return out.reshape(4, 4)


def view_space(pt_sphere, radius, center):
    pts = np.ascontiguousarray(pt_sphere, np.float64).reshape(-1, 3)
    c = np.ascontiguousarray(center, np.float64)
    out = np.zeros_like(pts)
    n = lib().orc_view_space(_p(pts), len(pts), float(radius), _p(c), _p(out))
    return out[:n]


def bbx(points):
    pts = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    c = np.zeros(3, np.float64)
    s = C.c_double()
    lib().orc_bbx(_p(pts), len(pts), _p(c), C.byref(s))
    return c, s.value


def nerf_to_ngp(tm, scale, offset):
    a = np.ascontiguousarray(tm, np.float64).reshape(16)
    off = np.ascontiguousarray(offset, np.float64)
    out = np.zeros(12, np.float32)
    lib().orc_nerf_to_ngp(_p(a), float(scale), _p(off), _p(out))
    return out.reshape(3, 4)


def cameras_from_transforms(tms, camera_angle_x, w, h, scale, offset):
    """oracle-side equivalent of prv_cameras_from_matrices"""
    focal = np.float32(0.5 * w / np.tan(0.5 * camera_angle_x))
    return [camera(nerf_to_ngp(tm, scale, offset), focal, focal, np.float32(0.5 * w), np.float32(0.5 * h)) for tm in tms]


def raygen(cam, w, h, spp_index=0):
    o, d, t = np.zeros((h * w, 3), np.float32), np.zeros((h * w, 3), np.float32), np.zeros((h * w, 2), np.float32)
    ox, oy = C.c_float(), C.c_float()
    lib().orc_spp_offset(spp_index, C.byref(ox), C.byref(oy))
    for y in range(h):
        for x in range(w):
            i = y * w + x
            lib().orc_raygen(C.byref(cam), x, y, ox, oy, _p(o[i]), _p(d[i]))
            t0, t1 = C.c_float(), C.c_float()
            lib().orc_ray_aabb(_p(o[i]), _p(d[i]), C.byref(t0), C.byref(t1))
            t[i] = (t0.value, t1.value)
    return o, d, t


def first_hit_image(field, cam, w, h, max_range=1e30, rows=None):
    """oracle: linear cell index of the first occupied voxel per pixel, or -1"""
    y0, y1 = rows if rows else (0, h)
    out = np.full((y1 - y0, w), -1, np.int32)
    lib().orc_first_hit_rows(field.ptr, C.byref(cam), w, y0, y1, C.c_float(max_range), _p(out))
    return out


def precept(field, voxels, c2w, intr9, width, height, model=2, max_range=1.0):
    v = np.ascontiguousarray(voxels, np.float32).reshape(-1, 3)
    c = np.ascontiguousarray(c2w, np.float64).reshape(16)
    w = np.ascontiguousarray(np.linalg.inv(c.reshape(4, 4)), np.float64).reshape(16)
    k = np.ascontiguousarray(intr9, np.float32)
    out = np.zeros(len(v), np.int32)
    lib().orc_precept(field.ptr, _p(v), len(v), _p(w), _p(c), _p(k), width, height, model, C.c_float(max_range), _p(out))
    return out


def quantize_rgba8(rgba, bg):
    rgba = np.ascontiguousarray(rgba, np.float32)
    out = np.zeros(rgba.shape, np.uint8)
    b = np.asarray(bg, np.float32)
    lib().orc_quantize_rgba8(_p(rgba), rgba.size // 4, _p(b), _p(out))
    return out


def _img_ptrs(imgs):
    imgs = [np.ascontiguousarray(i, np.uint8) for i in imgs]
    arr = (C.c_void_p * len(imgs))(*[i.ctypes.data for i in imgs])
    return imgs, arr


def score_ensemble_rgb(imgs):
    imgs, arr = _img_ptrs(imgs)
    return lib().orc_score_ensemble_rgb(arr, len(imgs), imgs[0].size // 4)


def score_ensemble_rgbdensity(imgs):
    imgs, arr = _img_ptrs(imgs)
    return lib().orc_score_ensemble_rgbdensity(arr, len(imgs), imgs[0].size // 4)


def score_psnr_coverage(rgba, gt, bg=(0, 0, 0, 0)):
    a, g = np.ascontiguousarray(rgba, np.float32), np.ascontiguousarray(gt, np.float32)
    b = np.asarray(bg, np.float32)
    p, c = C.c_double(), C.c_double()
    lib().orc_score_psnr_coverage(_p(a), _p(g), a.size // 4, _p(b), C.byref(p), C.byref(c))
    return p.value, c.value


def score_view(rgba, gt, bg=(0, 0, 0, 0), coverage_weight=1.0):
    """PRV_SCORE_PSNR_COVERAGE's ranking key and its parts -> (score, psnr, coverage)"""
    a, g = np.ascontiguousarray(rgba, np.float32), np.ascontiguousarray(gt, np.float32)
    b = np.asarray(bg, np.float32)
    s, p, c = C.c_double(), C.c_double(), C.c_double()
    lib().orc_score_view(_p(a), _p(g), a.size // 4, _p(b), float(coverage_weight), C.byref(s), C.byref(p), C.byref(c))
    return s.value, p.value, c.value


def ssim(rgba, gt, bg=(0, 0, 0, 0)):
    a, g = np.ascontiguousarray(rgba, np.float32), np.ascontiguousarray(gt, np.float32)
    b = np.asarray(bg, np.float32)
    return lib().orc_ssim(_p(a), _p(g), a.shape[1], a.shape[0], _p(b))


def rank(scores, ids):
    s, i = np.ascontiguousarray(scores, np.float64), np.ascontiguousarray(ids, np.int32)
    out = np.zeros(len(i), np.int32)
    lib().orc_rank(_p(s), _p(i), len(i), _p(out))
    return out


def argmax(scores, ids):
    s, i = np.ascontiguousarray(scores, np.float64), np.ascontiguousarray(ids, np.int32)
    return lib().orc_argmax(_p(s), _p(i), len(i))


def splat_points(xyz, rgb, scale, offset, cam, w, h, point_size=5, flip180=True):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    rgb = np.ascontiguousarray(rgb, np.uint8).reshape(-1, 3)
    off = np.ascontiguousarray(offset, np.float32)
    out = np.zeros((h, w, 4), np.uint8)
    lib().orc_splat_points(_p(xyz), _p(rgb), len(xyz), C.c_float(scale), _p(off), C.byref(cam), w, h, point_size,
                           int(flip180), _p(out))
    return out


class OracleTrainer:
    """the CPU oracle of the training step (oracle/prv_train.c)"""

    def __init__(self, field, opts, cams, images_rgba8, exact=False):
        self.images = np.ascontiguousarray(images_rgba8, np.uint8)
        n, h, w, _ = self.images.shape
        assert n == len(cams)
        self.cams = (Camera * n)(*cams)
        self.opts = opts
        self.desc = field.desc
        self.ptr = lib().orc_train_create(field.ptr, C.byref(opts), self.cams, n, w, h, _p(self.images), int(exact))
        if not self.ptr:
            raise ValueError("oracle trainer creation failed")
        self.n_table = lib().orc_train_table_size(self.ptr)

    def step(self):
        return lib().orc_train_step(self.ptr)

    def loss_only(self):
        return lib().orc_train_loss_only(self.ptr)

    def gradients(self):
        tg, mg = np.zeros(self.n_table, np.float64), np.zeros(MLP_HALFS, np.float64)
        loss = lib().orc_train_gradients(self.ptr, _p(tg), _p(mg))
        return loss, tg, mg

    def master(self):
        """views (not copies) of the fp32 master weights"""
        t = np.ctypeslib.as_array(lib().orc_train_master_table(self.ptr), shape=(self.n_table,))
        m = np.ctypeslib.as_array(lib().orc_train_master_mlp(self.ptr), shape=(MLP_HALFS,))
        return t, m

    def params(self):
        f = lib().orc_train_field(self.ptr).contents
        table = np.ctypeslib.as_array(f.table, shape=(self.n_table,)).copy()
        mlp = np.frombuffer(f.mlp, dtype=np.uint16).copy()
        R = self.desc.occ_res
        occ = np.ctypeslib.as_array(f.occ, shape=((R ** 3 + 31) // 32,)).copy()
        return table, mlp, occ

    def field(self):
        return OracleField(self.desc, params=self.params())

    @property
    def samples_last(self):
        return lib().orc_train_samples_last(self.ptr)

    @property
    def active_rays(self):
        return lib().orc_train_active_rays(self.ptr)

    def refresh_occupancy(self):
        lib().orc_train_refresh_occupancy(self.ptr)

    def __del__(self):
        try:
            lib().orc_train_free(self.ptr)
        except Exception:
            pass
