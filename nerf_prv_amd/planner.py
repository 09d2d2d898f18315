"""Planner-side host pieces in Python: binding of libprv_host.so (the C++ planner shell)
and the multi-GPU scoring round (views sharded across ranks, ONE all-gather of 16-byte
score records, identical ranking on every rank).

The sharding layer is backend-agnostic: it takes a `score_shard(view_ids) -> records`
callable.  In production that callable is `Context.score_views` (HIP); the CPU tests of the
N>1 path plug a checker in its place and run over gloo.
"""
import ctypes as C
import os

import numpy as np

from . import _lib as L
from .api import RECORD_DTYPE, rank_host

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "libprv_host.so")


class Intrinsics(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("ppx", C.c_double), ("ppy", C.c_double),
                ("fx", C.c_double), ("fy", C.c_double), ("coeffs", C.c_double * 5)]


class LoopResult(C.Structure):
    _fields_ = [("n_chosen", C.c_int), ("chosen", C.c_int * 1024), ("total_movement", C.c_double)]


SCORE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.c_int,
                       C.POINTER(C.c_double))

_vp, _i, _d = C.c_void_p, C.c_int, C.c_double
HOST_SIGNATURES = {
    "prvh_view_pose": (None, [_vp, _vp, _vp]),
    "prvh_transform_matrix": (None, [_vp, _vp]),
    "prvh_view_space": (_i, [_vp, _i, _d, _vp, _vp]),
    "prvh_bbx": (None, [_vp, _i, _vp, C.POINTER(_d)]),
    "prvh_hemisphere_read": (_i, [C.c_char_p, _i, _vp]),
    "prvh_hemisphere_generate": (_i, [_i, _vp]),
    "prvh_write_transforms": (_i, [C.c_char_p, C.POINTER(Intrinsics), _i, _d, _i, _d, _vp, _vp, _vp, _i, C.c_char_p]),
    "prvh_write_metrics": (_i, [C.c_char_p, _d, _d]),
    "prvh_read_metrics": (_i, [C.c_char_p, C.POINTER(_d), C.POINTER(_d)]),
    "prvh_local_path": (_d, [_vp, _vp, _vp, _d, C.POINTER(_i)]),
    "prvh_global_path": (_d, [_vp, _i, _i, _i, _vp, _d, _vp, C.POINTER(_i)]),
    "prvh_fit_curve": (_i, [_vp, _vp, _i, _d, _vp, C.POINTER(_i)]),
    "prvh_fit_labels": (None, [_vp, _d, _vp, _vp]),
    "prvh_write_label": (_i, [C.c_char_p, _vp, _i, _d]),
    "prvh_share_data_create": (_vp, [C.c_char_p, C.c_char_p, _i, _i, _i]),
    "prvh_share_data_destroy": (None, [_vp]),
    "prvh_share_data_error": (C.c_char_p, []),
    "prvh_share_data_string": (C.c_char_p, [_vp, C.c_char_p]),
    "prvh_share_data_number": (_d, [_vp, C.c_char_p]),
    "prvh_share_data_views": (_i, [_vp, _vp]),
    "prvh_share_data_intrinsics": (None, [_vp, C.POINTER(Intrinsics)]),
    "prvh_png_size": (_i, [C.c_char_p, C.POINTER(_i), C.POINTER(_i)]),
    "prvh_png_read_rgba8": (_i, [C.c_char_p, _i, _i, _vp]),
    "prvh_score_view_pngs": (_i, [_i, C.POINTER(C.c_char_p), _i, C.POINTER(C.c_double)]),
    "prvh_png_write_rgba8": (_i, [C.c_char_p, _i, _i, _vp]),
    "prvh_ingp_read": (_i, [C.c_char_p, _vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _vp, _vp, _vp, C.c_char_p, _i]),
    "prvh_ingp_write": (_i, [C.c_char_p, _vp, _vp, C.c_uint64, _vp, _vp, C.c_uint64, C.c_char_p, _i]),
    "prvh_star_open": (_vp, [_i, _i, C.c_char_p, _i, _d]),
    "prvh_star_close": (None, [_vp]),
    "prvh_star_all_gather": (_i, [_vp, _vp, C.c_uint64, _vp]),
    "prvh_star_broadcast": (_i, [_vp, _vp, C.c_uint64, _i]),
    "prvh_star_barrier": (_i, [_vp]),
    "prvh_nbv_loop": (_i, [_vp, _vp, _d, _i, _i, SCORE_FN, _vp, C.POINTER(LoopResult)]),
    "prvh_method_in_scope": (_i, [_i]),
    "prvh_member_owner": (_i, [_i, _i, _i, _i]),
    "prvh_pcd_read": (C.c_longlong, [C.c_char_p, _vp, _vp, C.c_longlong]),  # deprecated stub
    "prvh_nbv_loop_budget": (_i, [_vp, _vp, _d, _i, _i, SCORE_FN, _vp, _i, C.POINTER(LoopResult)]),  # deprecated stub
}

_host = None


def host():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError(f"{HOST_LIB_PATH} is missing: run __graft_entry__.build()")
        lib = C.CDLL(HOST_LIB_PATH)
        for name, (res, args) in HOST_SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError = missing export: fail loudly
            fn.restype, fn.argtypes = res, args
        _host = lib
    return _host


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ---- pose / view set (C++ View, View_Space) ---------------------------------------------

def view_pose(init_pos, center):
    a, b = np.ascontiguousarray(init_pos, np.float64), np.ascontiguousarray(center, np.float64)
    out = np.zeros(16, np.float64)
    host().prvh_view_pose(_p(a), _p(b), _p(out))
    return out.reshape(4, 4)


def transform_matrix(pose):
    a = np.ascontiguousarray(pose, np.float64).reshape(16)
    out = np.zeros(16, np.float64)
    host().prvh_transform_matrix(_p(a), _p(out))
    return out.reshape(4, 4)


def view_space(pt_sphere, radius, center):
    pts = np.ascontiguousarray(pt_sphere, np.float64).reshape(-1, 3)
    c = np.ascontiguousarray(center, np.float64)
    out = np.zeros_like(pts)
    n = host().prvh_view_space(_p(pts), len(pts), float(radius), _p(c), _p(out))
    return out[:n]


def bbx(points):
    pts = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    c, s = np.zeros(3, np.float64), C.c_double()
    host().prvh_bbx(_p(pts), len(pts), _p(c), C.byref(s))
    return c, s.value


def hemisphere_generate(n):
    out = np.zeros((n, 3), np.float64)
    host().prvh_hemisphere_generate(n, _p(out))
    return out


def hemisphere_read(path, n):
    out = np.zeros((n, 3), np.float64)
    rows = host().prvh_hemisphere_read(str(path).encode(), n, _p(out))
    if rows != n:
        raise IOError(f"{path}: expected {n} rows, read {rows}")
    return out


def hemisphere_transforms(pt_sphere, radius, predicted_size, center):
    """candidate set -> (transform_matrix[n,4,4], scale, offset) as the planner's json carries them
    (View_Space.hpp:550-556, main.cpp:1599-1602, 1626-1641)"""
    center = np.asarray(center, np.float64)
    pos = view_space(pt_sphere, radius, center)
    tms = np.stack([transform_matrix(view_pose(p, center)) for p in pos])
    scale = 0.5 / predicted_size
    offset = np.array([0.5 + center[2], 0.5 + center[0], 0.5 + center[1]])
    return tms, scale, offset


def write_transforms(path, intr, positions, center, predicted_size, ids=None, candidate=False, divisor=16.0,
                     aabb_scale=1, path_prefix="rgbaClip_"):
    pos = np.ascontiguousarray(positions, np.float64).reshape(-1, 3)
    c = np.ascontiguousarray(center, np.float64)
    idp = None if ids is None else _p(np.ascontiguousarray(ids, np.int32))
    rc = host().prvh_write_transforms(str(path).encode(), C.byref(intr), int(candidate), float(divisor), int(aabb_scale),
                                      float(predicted_size), _p(c), _p(pos), idp, len(pos), path_prefix.encode())
    if rc != 0:
        raise IOError(f"cannot write {path} (rc={rc})")


def write_metrics(path, psnr, ssim):
    if host().prvh_write_metrics(str(path).encode(), float(psnr), float(ssim)) != 0:
        raise IOError(f"cannot write {path}")


def read_metrics(path):
    p, s = C.c_double(), C.c_double()
    if host().prvh_read_metrics(str(path).encode(), C.byref(p), C.byref(s)) != 0:
        raise IOError(f"cannot read metrics from {path}")
    return p.value, s.value


def png_read(path):
    """RGBA8 pixels [h, w, 4] of a PNG through the host library (zlib only; what the C++ planner uses)"""
    w, h = C.c_int(), C.c_int()
    rc = host().prvh_png_size(str(path).encode(), C.byref(w), C.byref(h))
    if rc != 0:
        raise IOError(f"{path}: png error {rc}")
    out = np.zeros((h.value, w.value, 4), np.uint8)
    rc = host().prvh_png_read_rgba8(str(path).encode(), w.value, h.value, _p(out))
    if rc != 0:
        raise IOError(f"{path}: png error {rc}")
    return out


def score_view_pngs(method, files):
    """main.cpp:2045-2094 (method 2) / 2105-2158 (method 3) for one view: the E member screenshots -> view_uncertainty,
    on the host in the reference's operation order (prvh_score_view_pngs; what prv_planner's `score_path: png` runs)"""
    arr = (C.c_char_p * len(files))(*[str(f).encode() for f in files])
    out = C.c_double()
    rc = host().prvh_score_view_pngs(int(method), arr, len(files), C.byref(out))
    if rc != 0:
        raise IOError(f"score_view_pngs: error {rc}")
    return out.value


def png_write(path, rgba8):
    a = np.ascontiguousarray(rgba8, np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("png_write expects [h, w, 4] uint8")
    rc = host().prvh_png_write_rgba8(str(path).encode(), a.shape[1], a.shape[0], _p(a))
    if rc != 0:
        raise IOError(f"{path}: png error {rc}")


def ingp_read(path):
    """instant-ngp snapshot (.ingp / .msgpack) -> (FieldDesc, table u16, mlp u16, occ u32), canonical layout; no GPU"""
    d = L.FieldDesc()
    nt, no = C.c_uint64(), C.c_uint64()
    err = C.create_string_buffer(512)
    rc = host().prvh_ingp_read(str(path).encode(), C.byref(d), C.byref(nt), C.byref(no), None, None, None, err, 512)
    if rc != 0:
        raise (ValueError if rc == L.PRV_E_INVALID else IOError)(f"{path}: {err.value.decode()} (rc={rc})")
    table, mlp, occ = np.zeros(nt.value, np.uint16), np.zeros(L.MLP_HALFS, np.uint16), np.zeros(no.value, np.uint32)
    rc = host().prvh_ingp_read(str(path).encode(), C.byref(d), None, None, _p(table), _p(mlp), _p(occ), err, 512)
    if rc != 0:
        raise IOError(f"{path}: {err.value.decode()} (rc={rc})")
    return d, table, mlp, occ


def ingp_write(path, desc, table, mlp, occ):
    table, mlp, occ = (np.ascontiguousarray(table, np.uint16), np.ascontiguousarray(mlp, np.uint16),
                       np.ascontiguousarray(occ, np.uint32))
    err = C.create_string_buffer(512)
    rc = host().prvh_ingp_write(str(path).encode(), C.byref(desc), _p(table), table.size, _p(mlp), _p(occ), occ.size, err, 512)
    if rc != 0:
        raise (ValueError if rc == L.PRV_E_INVALID else IOError)(f"{path}: {err.value.decode()} (rc={rc})")


def local_path(M, N, O, r):
    """get_local_path (View_Space.hpp:206-305) -> (type, length)"""
    a, b, c = (np.ascontiguousarray(v, np.float64) for v in (M, N, O))
    t = C.c_int()
    d = host().prvh_local_path(_p(a), _p(b), _p(c), float(r), C.byref(t))
    return t.value, d


def global_path(positions, start, end=-1, center=(1e-10, 1e-10, 1e-10), radius=0.0):
    """Global_Path_Planner -> (visiting order, length, exact)"""
    pos = np.ascontiguousarray(positions, np.float64).reshape(-1, 3)
    c = np.ascontiguousarray(center, np.float64)
    order, exact = np.zeros(len(pos), np.int32), C.c_int()
    d = host().prvh_global_path(_p(pos), len(pos), int(start), int(end), _p(c), float(radius), _p(order), C.byref(exact))
    if d < 0:
        raise ValueError("global path: bad arguments")
    return order.tolist(), d, bool(exact.value)


def fit_curve(views, psnr, max_psnr):
    """PSNR-vs-#views LognormalCDF fit -> (params[y0,A,xc,w], converged)"""
    x, y = np.ascontiguousarray(views, np.float64), np.ascontiguousarray(psnr, np.float64)
    out, conv = np.zeros(4, np.float64), C.c_int()
    rc = host().prvh_fit_curve(_p(x), _p(y), len(x), float(max_psnr), _p(out), C.byref(conv))
    if rc != 0:
        raise ValueError(f"curve fit failed rc={rc}")
    return out, bool(conv.value)


def fit_labels(params, max_psnr):
    """-> (gap[11] view counts for 0..10 %, gradient[20] view counts for 0.01..0.20)"""
    p = np.ascontiguousarray(params, np.float64)
    gap, grad = np.zeros(11, np.int32), np.zeros(20, np.int32)
    host().prvh_fit_labels(_p(p), float(max_psnr), _p(gap), _p(grad))
    return gap, grad


def write_label(path, params, converged, max_psnr):
    p = np.ascontiguousarray(params, np.float64)
    if host().prvh_write_label(str(path).encode(), _p(p), int(converged), float(max_psnr)) != 0:
        raise IOError(f"cannot write {path}")


class ShareData:
    """binding of the C++ Share_Data (constructor signature of Share_Data.hpp:334)"""

    def __init__(self, config_file_path, test_name="", num_of_views=-1, id_of_batch=-1, test_method=-1):
        self.h = host().prvh_share_data_create(str(config_file_path).encode(), test_name.encode(), num_of_views,
                                               id_of_batch, test_method)
        if not self.h:
            raise IOError(host().prvh_share_data_error().decode())

    def string(self, field):
        return host().prvh_share_data_string(self.h, field.encode()).decode()

    def number(self, field):
        return host().prvh_share_data_number(self.h, field.encode())

    def views(self):
        n = host().prvh_share_data_views(self.h, None)
        out = np.zeros((n, 3), np.float64)
        host().prvh_share_data_views(self.h, _p(out))
        return out

    def intrinsics(self):
        k = Intrinsics()
        host().prvh_share_data_intrinsics(self.h, C.byref(k))
        return k

    def nbv_loop(self, center, predicted_size, score_fn, first_view_id=-1, test_id=0):
        """run NBV_Net_Labeler::nbv_loop; score_fn(method, iteration, scene_json, render_json, ids) -> scores"""
        def cb(user, method, iteration, scene, render, ids, n, scores):
            try:
                vals = score_fn(method, iteration, scene.decode(), render.decode(), [ids[i] for i in range(n)])
                for i in range(n):
                    scores[i] = float(vals[i])
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                self.last_error = e
                return -13

        c = np.ascontiguousarray(center, np.float64)
        res = LoopResult()
        keep = SCORE_FN(cb)
        rc = host().prvh_nbv_loop(self.h, _p(c), float(predicted_size), first_view_id, test_id, keep, None, C.byref(res))
        if rc != 0:
            raise RuntimeError(f"nbv_loop failed rc={rc}: {getattr(self, 'last_error', '')}")
        return [res.chosen[i] for i in range(res.n_chosen)]

    def close(self):
        if getattr(self, "h", None):
            host().prvh_share_data_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- multi-GPU scoring round --------------------------------------------------------------

def shard_views(n_views, rank, world, interleaved=False):
    """the views of rank -> (ids, per_rank = ceil(N/R)).  Contiguous block [rank*per, (rank+1)*per) by default;
    interleaved = round-robin rank, rank+R, ... : a hemisphere set is ordered pole -> equator and top views cost
    more than grazing ones (measured on the 512-view bench set: slowest rank 2.4 % above the mean with blocks,
    1.2 % interleaved), and the round's time is the slowest rank's"""
    per = -(-n_views // world)
    if interleaved:
        return np.arange(rank, n_views, world, dtype=np.int32), per
    lo = min(n_views, rank * per)
    return np.arange(lo, min(n_views, lo + per), dtype=np.int32), per


def gather_records(local, per_rank, n_views, group=None, device=None, interleaved=False):
    """ONE all-gather of the per-rank record blocks (16 bytes per view), padded to per_rank.

    `local` is either a numpy RECORD_DTYPE array (host; gloo) or a torch uint8 tensor of
    per_rank*16 bytes already on the device (RCCL).  Returns a host RECORD_DTYPE array of
    n_views records in view order, identical on every rank (interleaved shards are un-permuted here)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if isinstance(local, np.ndarray):
        buf = np.zeros(per_rank, RECORD_DTYPE)
        buf[: len(local)] = local
        send = torch.from_numpy(buf.view(np.uint8).copy())
        if device is not None:
            send = send.to(device)
    else:
        send = local
    if not dist.is_initialized():
        out = send
    else:
        out = torch.empty(world * per_rank * 16, dtype=torch.uint8, device=send.device)
        dist.all_gather_into_tensor(out, send, group=group)
    return assemble_records(out.cpu().numpy().view(RECORD_DTYPE), per_rank, n_views, world, interleaved)


def assemble_records(gathered, per_rank, n_views, world, interleaved=False):
    """the gathered array (world blocks of per_rank records, rank order) -> n_views records in view order:
    contiguous shards are already in order, interleaved shards are un-permuted (slot r*per_rank + k holds view
    k*world + r).  Padding slots of ragged shards are dropped."""
    gathered = np.asarray(gathered).view(RECORD_DTYPE).reshape(-1)
    if interleaved and world > 1:
        v = np.arange(n_views)
        return gathered[(v % world) * per_rank + v // world].copy()
    return gathered[:n_views].copy()


def scoring_round(n_views, score_shard, group=None, device=None, interleaved=False):
    """the distributed scoring round: shard -> score -> all-gather -> identical ranking.

    score_shard(view_ids) returns RECORD_DTYPE (host) or a uint8 device tensor of
    per_rank*16 bytes.  Returns (records[n_views], order[n_views])."""
    import torch.distributed as dist

    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    ids, per = shard_views(n_views, rank, world, interleaved)
    local = score_shard(ids)
    records = gather_records(local, per, n_views, group, device, interleaved)
    order = rank_host(records, np.arange(n_views, dtype=np.int32))
    return records, order


# ---------------------------------------------------------------- the ensemble across GPUs

def member_owner(e, world):
    """rank that trains ensemble member e (round-robin)"""
    return e % world


def exchange_members(local, n_members, sizes, group=None, device=None):
    """the real exchange step of a multi-GPU NBV iteration: rank r has trained the members e with
    e % world == r; every rank needs every member to score its shard of the candidate views.
    ONE all-gather of the packed fields (fp16 table | fp16 MLP | occupancy words per member, padded to
    ceil(E / world) members per rank).

    local: {e: (table_u16, mlp_u16, occ_u32)} for this rank's members; sizes = (n_table, n_mlp, n_occ).
    Returns {e: (table, mlp, occ)} for all e, identical on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_t, n_m, n_o = (int(x) for x in sizes)
    per_member = 2 * n_t + 2 * n_m + 4 * n_o
    per_rank = -(-n_members // world)
    mine = [e for e in range(n_members) if member_owner(e, world) == rank]
    if sorted(local) != mine:
        raise ValueError(f"rank {rank} must bring members {mine}, got {sorted(local)}")
    buf = np.zeros(per_rank * per_member, np.uint8)
    for k, e in enumerate(mine):
        t, m, o = local[e]
        if (t.size, m.size, o.size) != (n_t, n_m, n_o):
            raise ValueError("member arrays do not match the field's sizes")
        off = k * per_member
        buf[off: off + 2 * n_t] = np.ascontiguousarray(t, np.uint16).view(np.uint8)
        buf[off + 2 * n_t: off + 2 * n_t + 2 * n_m] = np.ascontiguousarray(m, np.uint16).view(np.uint8)
        buf[off + 2 * n_t + 2 * n_m: off + per_member] = np.ascontiguousarray(o, np.uint32).view(np.uint8)
    send = torch.from_numpy(buf)
    if device is not None:
        send = send.to(device)
    if dist.is_initialized():
        out = torch.empty(world * per_rank * per_member, dtype=torch.uint8, device=send.device)
        dist.all_gather_into_tensor(out, send, group=group)
    else:
        out = send
    host = out.cpu().numpy()
    result = {}
    for e in range(n_members):
        r, k = member_owner(e, world), e // world
        off = (r * per_rank + k) * per_member
        t = host[off: off + 2 * n_t].view(np.uint16).copy()
        m = host[off + 2 * n_t: off + 2 * n_t + 2 * n_m].view(np.uint16).copy()
        o = host[off + 2 * n_t + 2 * n_m: off + per_member].view(np.uint32).copy()
        result[e] = (t, m, o)
    return result


def train_ensemble(ctx, n_members, scene_json, n_steps, desc, seed=0x1234, opts=None, group=None, device=None, comm=None):
    """one NBV iteration's training (main.cpp:2041-2043: train_by_instantNGP once per ensemble member) over the
    GPUs of the job: this rank trains its members side by side on the scene json's views, the fields are
    exchanged, and slots 0..E-1 of `ctx` hold the whole ensemble on every rank.  comm (api.Comm): the exchange is
    prv_model_exchange -- device to device over RCCL, no host copy; without it (torch.distributed only, e.g. gloo in the
    CPU tests) the members travel through one all-gather of host-staged arrays."""
    import torch.distributed as dist

    from . import api
    from .compat_server import load_dataset_bytes

    if comm is not None:
        world, rank = comm.world, comm.rank
    else:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = [e for e in range(n_members) if member_owner(e, world) == rank]
    cams = ctx.cameras_from_dataset_json(scene_json)
    images = load_dataset_bytes(ctx, scene_json)
    trainers = []
    for e in mine:
        ctx.fresh_model(e, desc, seed + e)
        o = api.train_opts() if opts is None else opts
        member_opts = api.L.TrainOpts.from_buffer_copy(o)
        member_opts.seed = o.seed + e
        trainers.append(api.Trainer(ctx, e, cams, images, member_opts))
    losses = api.train_many(trainers, n_steps) if trainers else np.zeros((0, n_steps), np.float32)
    for t in trainers:
        t.close()
    cams.close()
    if world > 1 and comm is not None:
        comm.exchange_models(n_members, desc)
    elif world > 1:
        local = {e: ctx.export_model(e, desc) for e in mine}
        everyone = exchange_members(local, n_members, api.model_sizes(desc), group, device)
        for e in range(n_members):
            if e not in local:
                ctx.load_model(e, desc, *everyone[e])
    return dict(zip(mine, losses))
