"""Python host side of the render + view-scoring path, over the C ABI (include/prv.h).

Two layers:
  * `Context` / `CamSet`: thin object wrappers of the C entry points.  Device buffers are
    torch tensors (torch is only plumbing here: memory, streams, torch.distributed).
  * `Testbed`: mirrors the slice of `pyngp.Testbed` that the reference's
    Instantngp_scripts/run.py drives on this path (run.py:90-145, 226-247, 284-309):
    same attribute names and argument meaning, so a screenshot loop written against the
    reference reads the same here.

No CPU fallback: every call goes to libprv_hip.so; errors raise `PrvError`.
"""
import ctypes as C
import json
import math

import numpy as np

from . import _lib as L

RECORD_DTYPE = np.dtype([("score", "<f8"), ("psnr", "<f4"), ("coverage", "<f4")])

# the two synthetic fields BASELINE.md names
FIELD_256 = dict(n_levels=8, n_features=4, log2_hashmap=19, base_res=16, finest_res=256, occ_res=128,
                 density_bias=3.0, table_amp=4.0)
FIELD_512 = dict(n_levels=16, n_features=2, log2_hashmap=21, base_res=16, finest_res=512, occ_res=128,
                 density_bias=3.0, table_amp=4.0)
# a table that really leaves the caches (round 6): log2T = 24 at F = 2 -- seven hashed levels of 64 MiB each (448 MiB of random
# 4-byte gathers against a 256 MiB Infinity Cache) behind nine dense ones; levels beyond 16 MiB take the generic gather's 32-bit
# offsets (prv_api.cpp: FieldDev::wide_offsets)
FIELD_HBM = dict(n_levels=16, n_features=2, log2_hashmap=24, base_res=16, finest_res=2048, occ_res=128,
                 density_bias=3.0, table_amp=4.0)


class PrvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"prv error {code}: {msg}")
        self.code = code


def field_desc(**kw):
    d = dict(FIELD_256)
    d.update(kw)
    return L.FieldDesc(**d)


def render_opts(width, height, samples_per_ray=128, spp=1, min_transmittance=1e-4, background=(0.0, 0.0, 0.0, 0.0),
                step_mode=L.STEP_FIXED_S):
    o = L.RenderOpts()
    o.width, o.height, o.samples_per_ray, o.spp = int(width), int(height), int(samples_per_ray), int(spp)
    o.step_mode = int(step_mode)
    o.min_transmittance = float(min_transmittance)
    for k in range(4):
        o.background[k] = float(background[k])
    return o


def engine_render_opts(width, height, samples_per_ray, spp, min_transmittance, background=(0.0, 0.0, 0.0, 0.0)):
    """render options as the run.py mirrors (Testbed, the flag-file server) state them: samples_per_ray 0 = the engine's
    own stepping rule (PRV_STEP_NGP: dt = sqrt(3)/1024, what pyngp renders with behind run.py:304), N > 0 = N uniform
    samples per ray (PRV_STEP_FIXED_S, the BASELINE configs' rule)"""
    n = int(samples_per_ray)
    return render_opts(width, height, n, spp, min_transmittance, background, step_mode=L.STEP_NGP if n == 0 else L.STEP_FIXED_S)


def model_sizes(desc):
    lib = L.load()
    t, m, o = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = lib.prv_model_sizes(C.byref(desc), C.byref(t), C.byref(m), C.byref(o))
    if rc != 0:
        raise PrvError(rc, "invalid field descriptor")
    return t.value, m.value, o.value


def _ptr(a):
    """device or host pointer of a torch tensor / numpy array / None"""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(a.data_ptr())


class CamSet:
    def __init__(self, ctx, handle):
        self.ctx, self.handle = ctx, handle

    def __len__(self):
        return self.ctx.lib.prv_camset_count(self.handle)

    @property
    def size(self):
        w, h = C.c_int(), C.c_int()
        self.ctx.lib.prv_camset_size(self.handle, C.byref(w), C.byref(h))
        return w.value, h.value

    def get(self, i):
        c2w = np.zeros(12, np.float32)
        intr = np.zeros(4, np.float32)
        rc = self.ctx.lib.prv_camset_get(self.handle, i, _ptr(c2w), _ptr(intr))
        if rc != 0:
            raise PrvError(rc, "camera index out of range")
        return c2w.reshape(3, 4), intr

    def lens(self, i):
        """{k1, k2, p1, p2} of camera i"""
        lens = np.zeros(4, np.float32)
        if self.ctx.lib.prv_camset_lens(self.handle, i, _ptr(lens)) != 0:
            raise PrvError(L.PRV_E_INVALID, "camera index out of range")
        return lens

    def close(self):
        if self.handle:
            self.ctx.lib.prv_camset_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One context per process per GPU (one process per GPU in multi-GPU runs)."""

    def __init__(self, device=0, use_torch_stream=True):
        import torch

        self.torch = torch
        self.lib = L.load()
        h = C.c_void_p()
        rc = self.lib.prv_create(C.byref(h), int(device))
        if rc != 0:
            raise PrvError(rc, (self.lib.prv_last_error(None) or b"").decode())
        self.handle = h
        self.device = torch.device("cuda", int(device))
        if use_torch_stream:
            with torch.cuda.device(self.device):
                self.set_stream(torch.cuda.current_stream().cuda_stream)

    # -- plumbing
    def _chk(self, rc):
        if rc != 0:
            raise PrvError(rc, (self.lib.prv_last_error(self.handle) or b"").decode())

    def set_stream(self, raw_stream):
        self._chk(self.lib.prv_set_stream(self.handle, C.c_void_p(raw_stream)))

    def synchronize(self):
        self._chk(self.lib.prv_synchronize(self.handle))

    def set_coverage_weight(self, weight):
        """SCORE_PSNR_COVERAGE's key = -PSNR + weight * mean((1 - alpha)^2); default 1, 0 = PSNR alone"""
        self._chk(self.lib.prv_set_coverage_weight(self.handle, float(weight)))

    def profile_begin(self):
        self._chk(self.lib.prv_profile_begin(self.handle))

    def profile_end(self):
        """-> dict(render_ms, render_launches, march_ms, march_launches) from HIP events"""
        rm, mm, rn, mn = C.c_double(), C.c_double(), C.c_int(), C.c_int()
        self._chk(self.lib.prv_profile_end(self.handle, C.byref(rm), C.byref(rn), C.byref(mm), C.byref(mn)))
        each = (C.c_float * max(1, rn.value))()
        n = self.lib.prv_profile_render_launches(self.handle, each, rn.value)
        return dict(render_ms=rm.value, render_launches=rn.value, march_ms=mm.value, march_launches=mn.value,
                    render_launch_ms=[float(each[i]) for i in range(max(0, min(n, rn.value)))])

    def close(self):
        if getattr(self, "handle", None):
            self.lib.prv_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- model
    def load_model(self, slot, desc, table, mlp, occ):
        table = np.ascontiguousarray(table, np.uint16)
        mlp = np.ascontiguousarray(mlp, np.uint16)
        occ = np.ascontiguousarray(occ, np.uint32)
        t, m, o = model_sizes(desc)
        if table.size != t or mlp.size != m or occ.size != o:
            raise PrvError(L.PRV_E_INVALID, f"parameter sizes {table.size},{mlp.size},{occ.size} != {t},{m},{o}")
        self._chk(self.lib.prv_model_load(self.handle, slot, C.byref(desc), _ptr(table), _ptr(mlp), _ptr(occ)))

    def synthetic_model(self, slot, desc, seed):
        self._chk(self.lib.prv_model_synthetic(self.handle, slot, C.byref(desc), C.c_uint64(seed)))

    def fresh_model(self, slot, desc, seed):
        """training start: small random table, Xavier MLP, all cells occupied"""
        self._chk(self.lib.prv_model_fresh(self.handle, slot, C.byref(desc), C.c_uint64(seed)))

    def export_model(self, slot, desc):
        t, m, o = model_sizes(desc)
        table, mlp, occ = np.empty(t, np.uint16), np.empty(m, np.uint16), np.empty(o, np.uint32)
        self._chk(self.lib.prv_model_export(self.handle, slot, _ptr(table), _ptr(mlp), _ptr(occ)))
        return table, mlp, occ

    def save_model(self, slot, path):
        self._chk(self.lib.prv_model_save_file(self.handle, slot, str(path).encode()))

    def load_model_file(self, slot, path):
        self._chk(self.lib.prv_model_load_file(self.handle, slot, str(path).encode()))

    def load_ingp(self, slot, path):
        """an instant-ngp snapshot (.ingp / .msgpack) -> slot; returns its FieldDesc"""
        self._chk(self.lib.prv_model_load_ingp(self.handle, slot, str(path).encode()))
        return self.model_desc(slot)

    def save_ingp(self, slot, path):
        self._chk(self.lib.prv_model_save_ingp(self.handle, slot, str(path).encode()))

    def model_desc(self, slot):
        d = L.FieldDesc()
        self._chk(self.lib.prv_model_desc(self.handle, slot, C.byref(d)))
        return d

    # -- cameras
    def cameras_from_json(self, path):
        h = C.c_void_p()
        self._chk(self.lib.prv_cameras_from_json(self.handle, str(path).encode(), C.byref(h)))
        return CamSet(self, h)

    def cameras_from_dataset_json(self, path):
        """the json's own intrinsics block (fl, principal point, lens): the engine's view of a dataset"""
        h = C.c_void_p()
        self._chk(self.lib.prv_cameras_from_dataset_json(self.handle, str(path).encode(), C.byref(h)))
        return CamSet(self, h)

    def cameras_from_matrices_intr(self, tm, intr, scale, offset):
        """intr: dict with fl_x fl_y cx cy w h and optional k1 k2 p1 p2"""
        tm = np.ascontiguousarray(tm, np.float64).reshape(-1, 16)
        off = np.ascontiguousarray(offset, np.float64)
        ci = L.Intrinsics(**{k: (int(v) if k in ("w", "h") else float(v)) for k, v in intr.items()})
        h = C.c_void_p()
        self._chk(self.lib.prv_cameras_from_matrices_intr(self.handle, _ptr(tm), tm.shape[0], C.byref(ci), float(scale),
                                                          _ptr(off), C.byref(h)))
        return CamSet(self, h)

    def cameras_from_matrices(self, tm, camera_angle_x, width, height, scale, offset):
        tm = np.ascontiguousarray(tm, np.float64).reshape(-1, 16)
        off = np.ascontiguousarray(offset, np.float64)
        h = C.c_void_p()
        self._chk(self.lib.prv_cameras_from_matrices(self.handle, _ptr(tm), tm.shape[0], float(camera_angle_x),
                                                     int(width), int(height), float(scale), _ptr(off), C.byref(h)))
        return CamSet(self, h)

    def splat_points(self, xyz, rgb, scale, offset, camset, view_ids, width, height, point_size=5, flip180=True):
        """ground-truth images of a coloured cloud (get_coverage's rgbaClip images) -> uint8 [n, h, w, 4] on the device"""
        t = self.torch
        xyz = t.as_tensor(np.ascontiguousarray(xyz, np.float32)).to(self.device).contiguous().reshape(-1, 3)
        rgb = t.as_tensor(np.ascontiguousarray(rgb, np.uint8)).to(self.device).contiguous().reshape(-1, 3)
        ids = self._ids(camset, view_ids)
        off = np.ascontiguousarray(offset, np.float64)
        out = t.empty((len(ids), height, width, 4), dtype=t.uint8, device=self.device)
        self._chk(self.lib.prv_splat_points(self.handle, _ptr(xyz), _ptr(rgb), xyz.shape[0], float(scale), _ptr(off),
                                            camset.handle, _ptr(ids), len(ids), int(width), int(height), int(point_size),
                                            int(flip180), _ptr(out)))
        return out

    # -- render
    def _ids(self, camset, view_ids):
        if view_ids is None:
            view_ids = np.arange(len(camset), dtype=np.int32)
        return np.ascontiguousarray(view_ids, np.int32)

    def render(self, slot, camset, view_ids, opts, out=None, want_stats=True):
        ids = self._ids(camset, view_ids)
        if out is None:
            out = self.torch.empty((len(ids), opts.height, opts.width, 4), dtype=self.torch.float32, device=self.device)
        st = L.Stats()
        self._chk(self.lib.prv_render(self.handle, slot, camset.handle, _ptr(ids), len(ids), C.byref(opts), _ptr(out),
                                      C.byref(st) if want_stats else None))
        return out, st

    def render_rgba8(self, slot, camset, view_ids, opts, out=None, want_stats=True):
        ids = self._ids(camset, view_ids)
        if out is None:
            out = self.torch.empty((len(ids), opts.height, opts.width, 4), dtype=self.torch.uint8, device=self.device)
        st = L.Stats()
        self._chk(self.lib.prv_render_rgba8(self.handle, slot, camset.handle, _ptr(ids), len(ids), C.byref(opts),
                                            _ptr(out), C.byref(st) if want_stats else None))
        return out, st

    def first_hit(self, slot, camset, view_ids, width, height, max_range=1e30):
        ids = self._ids(camset, view_ids)
        out = self.torch.empty((len(ids), height, width), dtype=self.torch.int32, device=self.device)
        self._chk(self.lib.prv_first_hit(self.handle, slot, camset.handle, _ptr(ids), len(ids), width, height,
                                         float(max_range), _ptr(out)))
        return out

    def precept(self, slot, voxels, c2w, intr, max_range=1.0):
        """Perception_3D::precept per voxel (main.cpp:98-284): voxels = device float tensor n x 3 -> int32 cells"""
        c = np.ascontiguousarray(c2w, np.float64).reshape(16)
        out = self.torch.empty((voxels.shape[0],), dtype=self.torch.int32, device=self.device)
        self._chk(self.lib.prv_precept(self.handle, slot, _ptr(voxels), voxels.shape[0], _ptr(c), C.byref(intr),
                                       float(max_range), _ptr(out)))
        return out

    def quantize_rgba8(self, rgba, background):
        out = self.torch.empty(rgba.shape, dtype=self.torch.uint8, device=self.device)
        bg = np.asarray(background, np.float32)
        self._chk(self.lib.prv_quantize_rgba8(self.handle, _ptr(rgba), rgba.numel() // 4, _ptr(bg), _ptr(out)))
        return out

    # -- scores
    def score_ensemble_images(self, method, images):
        n_views = images[0].shape[0]
        npix = images[0].numel() // 4 // max(n_views, 1)
        arr = (C.c_void_p * len(images))(*[im.data_ptr() for im in images])
        rec = np.zeros(n_views, RECORD_DTYPE)
        self._chk(self.lib.prv_score_ensemble_images(self.handle, method, arr, len(images), n_views, npix, _ptr(rec)))
        return rec

    def score_psnr_images(self, rgba, gt, background=(0, 0, 0, 0)):
        n_views = rgba.shape[0]
        npix = rgba.numel() // 4 // max(n_views, 1)
        bg = np.asarray(background, np.float32)
        rec = np.zeros(n_views, RECORD_DTYPE)
        self._chk(self.lib.prv_score_psnr_images(self.handle, _ptr(rgba), _ptr(gt), n_views, npix, _ptr(bg), _ptr(rec)))
        return rec

    def evaluate_images(self, rgba, gt, background=(0, 0, 0, 0)):
        """per-image PSNR and SSIM (run.py:257-263) -> (psnr[n], ssim[n]) float64"""
        n, h, w = rgba.shape[0], rgba.shape[1], rgba.shape[2]
        bg = np.asarray(background, np.float32)
        ps, ss = np.zeros(n, np.float64), np.zeros(n, np.float64)
        self._chk(self.lib.prv_evaluate_images(self.handle, _ptr(rgba), _ptr(gt), n, w, h, _ptr(bg), _ptr(ps), _ptr(ss)))
        return ps, ss

    def evaluate(self, slot, camset, view_ids, opts, gt):
        """the evaluation loop of run.py:240-277 -> (mean psnr, mean ssim)"""
        ids = self._ids(camset, view_ids)
        p, s = C.c_double(), C.c_double()
        self._chk(self.lib.prv_evaluate(self.handle, slot, camset.handle, _ptr(ids), len(ids), C.byref(opts), _ptr(gt),
                                        C.byref(p), C.byref(s)))
        return p.value, s.value

    def score_views(self, method, slots, camset, view_ids, opts, gt=None, records_dev=None, to_host=True,
                    want_stats=False):
        ids = self._ids(camset, view_ids)
        slots = np.ascontiguousarray(slots, np.int32)
        rec = np.zeros(len(ids), RECORD_DTYPE) if to_host else None
        st = L.Stats()
        self._chk(self.lib.prv_score_views(self.handle, method, _ptr(slots), len(slots), camset.handle, _ptr(ids),
                                           len(ids), C.byref(opts), _ptr(gt), _ptr(rec), _ptr(records_dev),
                                           C.byref(st) if want_stats else None))
        return rec, st

    def rank(self, records, view_ids):
        records = np.ascontiguousarray(records, RECORD_DTYPE)
        ids = np.ascontiguousarray(view_ids, np.int32)
        order = np.zeros(len(ids), np.int32)
        self._chk(self.lib.prv_rank(_ptr(records), _ptr(ids), len(ids), _ptr(order)))
        return order

    def argmax(self, records, view_ids):
        records = np.ascontiguousarray(records, RECORD_DTYPE)
        ids = np.ascontiguousarray(view_ids, np.int32)
        return self.lib.prv_argmax(_ptr(records), _ptr(ids), len(ids))

    # -- stage hooks
    def model_layout(self, slot):
        """kernel-side layout of a loaded field and the render_queue64_kernel<F, NDENSE> instance it runs on"""
        out = L.ModelLayout()
        self._chk(self.lib.prv_debug_model_layout(self.handle, slot, C.byref(out)))
        return {k: getattr(out, k) for k, _ in out._fields_}

    def render_clock_ghz(self):
        """average shader clock of the render launches since the statistics were last cleared (prv_debug_render_clock)"""
        cyc, ticks, hz = C.c_uint64(), C.c_uint64(), C.c_double()
        self._chk(self.lib.prv_debug_render_clock(self.handle, C.byref(cyc), C.byref(ticks), C.byref(hz)))
        return cyc.value / (ticks.value / hz.value) / 1e9 if ticks.value else 0.0

    def debug_raygen(self, camset, view, width, height, spp_index=0):
        n = width * height
        o, d, t = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32), np.zeros((n, 2), np.float32)
        self._chk(self.lib.prv_debug_raygen(self.handle, camset.handle, view, width, height, spp_index, _ptr(o), _ptr(d),
                                            _ptr(t)))
        return o, d, t

    def debug_encode(self, slot, pos):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        feat = np.zeros((pos.shape[0], 32), np.uint16)
        self._chk(self.lib.prv_debug_encode(self.handle, slot, _ptr(pos), pos.shape[0], _ptr(feat)))
        return feat

    def debug_field(self, slot, pos, dirs):
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.zeros((pos.shape[0], 36), np.float32)
        occ = np.zeros(pos.shape[0], np.int32)
        self._chk(self.lib.prv_debug_field(self.handle, slot, _ptr(pos), _ptr(dirs), pos.shape[0], _ptr(out), _ptr(occ)))
        return out, occ


class Comm:
    """the ranks of one job as the C ABI sees them (include/prv.h, "several GPUs"): RCCL on device buffers, or the
    host-staged socket transport for ranks that share a GPU"""

    def __init__(self, ctx, rank, world, transport=None, rendezvous=None):
        self.ctx = ctx
        self.handle = C.c_void_p()
        ctx._chk(ctx.lib.prv_comm_create(ctx.handle, int(rank), int(world), transport.encode() if transport else None,
                                         rendezvous.encode() if rendezvous else None, C.byref(self.handle)))
        self.rank, self.world = int(rank), int(world)

    @property
    def transport(self):
        return self.ctx.lib.prv_comm_transport(self.handle).decode()

    @property
    def library(self):
        """{"path", "version", "found"}: the librccl file this communicator's calls land in (empty for "socket")"""
        path, how, ver = C.create_string_buffer(1024), C.create_string_buffer(128), C.c_int(0)
        self.ctx._chk(self.ctx.lib.prv_comm_library(self.handle, path, len(path), C.byref(ver), how, len(how)))
        return {"path": path.value.decode(), "version": int(ver.value), "found": how.value.decode()}

    def all_gather(self, send):
        """send: device tensor -> device uint8 tensor of world blocks in rank order"""
        t = self.ctx.torch
        send = send.contiguous()
        nbytes = send.numel() * send.element_size()
        out = t.empty(self.world * nbytes, dtype=t.uint8, device=send.device)
        self.ctx._chk(self.ctx.lib.prv_comm_all_gather(self.handle, _ptr(send), nbytes, _ptr(out)))
        self.ctx.synchronize()
        return out

    def barrier(self):
        self.ctx._chk(self.ctx.lib.prv_comm_barrier(self.handle))

    def score_views(self, method, slots, camset, n_views, opts, gt_shard=None, interleaved=True, want_stats=False):
        """the sharded scoring round -> (records[n_views] in view order, identical on every rank; local stats)"""
        slots = np.ascontiguousarray(slots, np.int32)
        rec = np.zeros(int(n_views), RECORD_DTYPE)
        st = L.Stats()
        self.ctx._chk(self.ctx.lib.prv_score_views_sharded(self.ctx.handle, self.handle, method, _ptr(slots), len(slots),
                                                           camset.handle, int(n_views), int(bool(interleaved)), C.byref(opts),
                                                           _ptr(gt_shard), _ptr(rec), C.byref(st) if want_stats else None))
        return rec, st

    def exchange_models(self, n_members, desc):
        """member e trained in slot e of rank e % world -> slot e of every rank (device to device)"""
        self.ctx._chk(self.ctx.lib.prv_model_exchange(self.ctx.handle, self.handle, int(n_members), C.byref(desc)))

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.lib.prv_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_views(n_views, rank, world, interleaved=False):
    """prv_shard_views -> (ids, per_rank)"""
    lib = L.load()
    per = -(-int(n_views) // int(world)) if world > 0 else 0
    ids = np.zeros(max(1, per), np.int32)
    n = C.c_int()
    per = lib.prv_shard_views(int(n_views), int(rank), int(world), int(bool(interleaved)), _ptr(ids), C.byref(n))
    if per < 0:
        raise PrvError(per, "prv_shard_views")
    return ids[: n.value].copy(), per


def rank_host(records, view_ids):
    """ranking without a context (pure host entry point of the C ABI)"""
    lib = L.load()
    records = np.ascontiguousarray(records, RECORD_DTYPE)
    ids = np.ascontiguousarray(view_ids, np.int32)
    order = np.zeros(len(ids), np.int32)
    rc = lib.prv_rank(_ptr(records), _ptr(ids), len(ids), _ptr(order))
    if rc != 0:
        raise PrvError(rc, "prv_rank")
    return order


def train_opts(**kw):
    """prv_train_default_opts with overrides.  n_samples is a parameter of the sampling rule (include/prv.h): given without
    step_mode it selects PRV_STEP_FIXED_S (n_samples uniform samples per ray); step_mode given without it gets that rule's
    default (128 samples / 1024 steps)."""
    o = L.TrainOpts()
    rc = L.load().prv_train_default_opts(C.byref(o))
    if rc != 0:
        raise PrvError(rc, "prv_train_default_opts")
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown training option {k}")
        setattr(o, k, v)
    if "step_mode" in kw and "n_samples" not in kw:
        o.n_samples = L.NGP_MAX_STEPS if kw["step_mode"] == L.STEP_NGP else 128
    if "n_samples" in kw and "step_mode" not in kw:
        o.step_mode = L.STEP_FIXED_S
    if "step_mode" not in kw and max(o.patch_w, 1) * max(o.patch_h, 1) > 1:  # pixel patches are an option of the fixed rule
        o.step_mode = L.STEP_FIXED_S
        if "n_samples" not in kw:
            o.n_samples = 128
    return o


class Trainer:
    """the in-process training step on one model slot (include/prv.h, training section)"""

    def __init__(self, ctx, slot, camset, images_u8, opts=None):
        t = ctx.torch
        self.ctx, self.slot, self.camset = ctx, slot, camset
        self.images = images_u8.to(device=ctx.device, dtype=t.uint8).contiguous()  # kept alive here
        n, h, w, ch = self.images.shape
        if ch != 4 or n != len(camset):
            raise ValueError("images must be [n_views, h, w, 4], one per dataset camera")
        self.opts = opts if opts is not None else train_opts()
        self.handle = C.c_void_p()
        ctx._chk(ctx.lib.prv_train_create(ctx.handle, slot, camset.handle, _ptr(self.images), w, h, C.byref(self.opts),
                                          C.byref(self.handle)))

    def steps(self, n):
        losses = np.zeros(int(n), np.float32)
        self.ctx._chk(self.ctx.lib.prv_train_steps(self.handle, int(n), _ptr(losses)))
        return losses

    def info(self):
        s, u, n = C.c_uint32(), C.c_uint64(), C.c_uint64()
        self.ctx.lib.prv_train_info(self.handle, C.byref(s), C.byref(u), C.byref(n))
        return {"steps": s.value, "samples_last": u.value, "table_scalars": n.value,
                "active_rays": self.ctx.lib.prv_train_active_rays(self.handle)}

    def gradients(self):
        n = self.info()["table_scalars"]
        tg, mg, loss = np.zeros(n, np.float32), np.zeros(L.MLP_HALFS, np.float32), C.c_float()
        self.ctx._chk(self.ctx.lib.prv_train_gradients(self.handle, _ptr(tg), _ptr(mg), C.byref(loss)))
        return loss.value, tg, mg

    def master(self):
        n = self.info()["table_scalars"]
        tw, mw = np.zeros(n, np.float32), np.zeros(L.MLP_HALFS, np.float32)
        self.ctx._chk(self.ctx.lib.prv_train_master(self.handle, _ptr(tw), _ptr(mw)))
        return tw, mw

    def refresh_occupancy(self):
        self.ctx._chk(self.ctx.lib.prv_train_refresh_occupancy(self.handle))

    def close(self):
        if self.handle:
            self.ctx.lib.prv_train_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def train_many(trainers, n):
    """n steps of every trainer side by side (prv_train_steps_multi) -> losses [n_trainers, n]"""
    ctx = trainers[0].ctx
    arr = (C.c_void_p * len(trainers))(*[t.handle for t in trainers])
    losses = np.zeros((len(trainers), int(n)), np.float32)
    ctx._chk(ctx.lib.prv_train_steps_multi(arr, len(trainers), int(n), _ptr(losses)))
    return losses


class _NerfSettings:
    """stands in for testbed.nerf (run.py:140,145,235)"""

    def __init__(self):
        self.render_min_transmittance = 0.01  # engine default; run.py:235 sets 1e-4 for evaluation
        self.render_with_lens_distortion = False
        self.sharpen = 0.0
        self.samples_per_ray = 0  # not a pyngp attribute: 0 = the engine's own stepping (dt = sqrt(3)/1024), N > 0 = N uniform samples per ray
        self.training = _Training()


class _FrameMeta:
    def __init__(self, w, h):
        self.resolution = (w, h)


class _Dataset:
    """testbed.nerf.training.dataset (run.py:240-241): n_images, metadata[i].resolution"""

    def __init__(self):
        self.n_images = 0
        self.metadata = []


class _Training:
    def __init__(self):
        self.dataset = _Dataset()


class Testbed:
    """The slice of pyngp.Testbed that run.py uses on the render path.

    run.py:90   testbed = ngp.Testbed()
    run.py:94   testbed.background_color = [0,0,0,1]
    run.py:109  testbed.load_training_data(scene)       -> scale/offset/intrinsics of the json
    run.py:127  testbed.load_snapshot(...)              -> load_model(...) / synthetic_model(...)
    run.py:231  testbed.snap_to_pixel_centers = True
    run.py:285  testbed.fov_axis = 0 ; testbed.fov = camera_angle_x * 180 / pi
    run.py:296  testbed.set_nerf_camera_matrix(M[:-1,:])
    run.py:304  image = testbed.render(w, h, spp, True)  -> float32 HxWx4, linear
    """

    def __init__(self, device=0):
        self.ctx = Context(device)
        self.background_color = [0.0, 0.0, 0.0, 1.0]
        self.fov_axis = 0
        self.fov = 50.625
        self.snap_to_pixel_centers = False
        self.shall_train = False
        self.exposure = 0.0
        self.nerf = _NerfSettings()
        self.scale = 0.33
        self.offset = [0.5, 0.5, 0.5]
        self._matrix = np.eye(4)[:3]
        self._slot = 0
        self._have_model = False
        self.render_ground_truth = False
        self.steps_per_frame = 16  # ASSUMED: upstream trains a batch of steps per frame() call
        self.train_options = None  # TrainOpts override
        self._trainer = None
        self._last_loss = 0.0
        self._dataset_cams = None
        self._dataset_path = None
        self._training_view = None

    def load_training_data(self, path):
        with open(path) as f:
            meta = json.load(f)
        self.scale = float(meta.get("scale", 0.33))
        self.offset = [float(x) for x in meta.get("offset", [0.5, 0.5, 0.5])]
        if "camera_angle_x" in meta:
            self.fov_axis, self.fov = 0, meta["camera_angle_x"] * 180.0 / math.pi
        self.training_meta = meta
        self._drop_trainer()
        if self._dataset_cams is not None:
            self._dataset_cams.close()
            self._dataset_cams = None
        ds = self.nerf.training.dataset
        ds.n_images, ds.metadata = 0, []
        if meta.get("frames") and "w" in meta and "h" in meta:  # the dataset's own cameras (run.py:238-242)
            self._dataset_cams = self.ctx.cameras_from_dataset_json(path)
            self._dataset_path = str(path)
            ds.n_images = len(self._dataset_cams)
            ds.metadata = [_FrameMeta(int(meta["w"]), int(meta["h"])) for _ in range(ds.n_images)]

    def reset_network(self, desc, seed=0x1234):
        """a new network to train (what a fresh pyngp.Testbed holds, run.py:90)"""
        self.ctx.fresh_model(self._slot, desc, seed)
        self._have_model = True
        self._drop_trainer()

    def _drop_trainer(self):
        if self._trainer is not None:
            self._trainer.close()
            self._trainer = None

    @property
    def training_step(self):  # run.py:191
        return self._trainer.info()["steps"] if self._trainer is not None else 0

    @property
    def loss(self):  # run.py:206
        return self._last_loss

    def frame(self):
        """one iteration of `while testbed.frame()` (run.py:187): a batch of optimiser steps when shall_train"""
        if self.shall_train:
            if not self._have_model:
                raise PrvError(L.PRV_E_STATE, "no network: call reset_network(desc) or load a snapshot first")
            if self._dataset_cams is None:
                raise PrvError(L.PRV_E_STATE, "no training data loaded")
            if self._trainer is None:
                from .compat_server import load_dataset_bytes

                self._trainer = Trainer(self.ctx, self._slot, self._dataset_cams,
                                        load_dataset_bytes(self.ctx, self._dataset_path), self.train_options)
            self._last_loss = float(self._trainer.steps(self.steps_per_frame)[-1])
        return True

    def set_camera_to_training_view(self, i):  # run.py:242
        if self._dataset_cams is None or not 0 <= int(i) < len(self._dataset_cams):
            raise PrvError(L.PRV_E_INVALID, "no such training view")
        self._training_view = int(i)

    def load_model(self, desc, table, mlp, occ):
        self.ctx.load_model(self._slot, desc, table, mlp, occ)
        self._have_model = True
        self._drop_trainer()

    def synthetic_model(self, desc, seed):
        self.ctx.synthetic_model(self._slot, desc, seed)
        self._have_model = True
        self._drop_trainer()

    def load_snapshot(self, path):  # run.py:127
        """instant-ngp's own snapshots (.ingp / .msgpack) or this build's .prvf"""
        if str(path).lower().endswith((".ingp", ".msgpack")):
            self.ctx.load_ingp(self._slot, path)
        else:
            self.ctx.load_model_file(self._slot, path)
        self._have_model = True
        self._drop_trainer()

    def save_snapshot(self, path, include_optimizer_state=False):  # run.py:211
        if str(path).lower().endswith((".ingp", ".msgpack")):
            self.ctx.save_ingp(self._slot, path)
        else:
            self.ctx.save_model(self._slot, path)

    def set_nerf_camera_matrix(self, m):
        m = np.asarray(m, np.float64)
        if m.shape != (3, 4):
            raise ValueError("set_nerf_camera_matrix expects a 3x4 matrix")
        self._matrix = m
        self._training_view = None

    def render(self, width, height, spp=1, linear=True):
        if not self._have_model:
            raise PrvError(L.PRV_E_STATE, "no model loaded")
        if not linear:
            raise NotImplementedError("only linear=True is used on the reference path (run.py:245,247,304)")
        if self.fov_axis != 0:
            raise NotImplementedError("fov_axis must be 0 (run.py:285)")
        eff_spp = 1 if self.snap_to_pixel_centers else int(spp)
        opts = engine_render_opts(width, height, self.nerf.samples_per_ray, eff_spp, self.nerf.render_min_transmittance)
        if self._training_view is not None:  # dataset camera: own intrinsics + lens (run.py:242-247)
            if self.render_ground_truth:
                img = self._ground_truth(self._training_view, width, height)
            else:
                img, _ = self.ctx.render(self._slot, self._dataset_cams, [self._training_view], opts, want_stats=False)
                img = img[0]
        else:
            tm = np.vstack([self._matrix, [0, 0, 0, 1]])
            cams = self.ctx.cameras_from_matrices(tm, self.fov * math.pi / 180.0, width, height, self.scale, self.offset)
            img, _ = self.ctx.render(self._slot, cams, None, opts, want_stats=False)
            img = img[0]
            cams.close()
        bg = self.ctx.torch.tensor(self.background_color, dtype=img.dtype, device=img.device)
        img = img + (1.0 - img[..., 3:4]) * bg  # composite over the background colour
        return img.cpu().numpy()

    def _ground_truth(self, i, width, height):
        """render_ground_truth (run.py:241-244): the dataset image of view i, linear premultiplied RGBA"""
        from .compat_server import load_reference_images

        if getattr(self, "_gt_cache_path", None) != self._dataset_path:
            self._gt_cache = load_reference_images(self.ctx, self._dataset_path)
            self._gt_cache_path = self._dataset_path
        img = self._gt_cache[i]
        if tuple(img.shape[:2]) != (height, width):
            raise NotImplementedError("ground-truth images are only served at their own resolution (run.py:240-244)")
        return img
