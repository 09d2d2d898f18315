"""ctypes binding of include/prv.h (libprv_hip.so).

The library is the product; there is NO fallback.  If the shared object is missing
or a symbol declared in include/prv.h is not exported, importing fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libprv_hip.so")

PRV_OK = 0
PRV_E_INVALID = -1
PRV_E_HIP = -2
PRV_E_IO = -3
PRV_E_NODEVICE = -4
PRV_E_STATE = -5
PRV_E_INTERNAL = -6

SCORE_ENSEMBLE_RGB = 2
SCORE_ENSEMBLE_RGB_DENSITY = 3
SCORE_PSNR_COVERAGE = 5
STEP_FIXED_S = 0  # prv.h: samples_per_ray uniform samples between the AABB hits (BASELINE configs[1], [3])
STEP_NGP = 1      # instant-ngp's rule, what run.py:304 renders with: dt = sqrt(3)/1024, every step tested, no cap
NGP_MAX_STEPS = 1024

MAX_MODELS = 8
MAX_SLOTS = 64
MLP_HALFS = 10240


class FieldDesc(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int32),
        ("n_features", C.c_int32),
        ("log2_hashmap", C.c_int32),
        ("base_res", C.c_int32),
        ("finest_res", C.c_int32),
        ("occ_res", C.c_int32),
        ("density_bias", C.c_float),
        ("table_amp", C.c_float),
        ("per_level_scale", C.c_float),
    ]


class RenderOpts(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("samples_per_ray", C.c_int32),
        ("spp", C.c_int32),
        ("min_transmittance", C.c_float),
        ("background", C.c_float * 4),
        ("step_mode", C.c_int32),
    ]


class Rs2Intrinsics(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("ppx", C.c_float), ("ppy", C.c_float), ("fx", C.c_float),
                ("fy", C.c_float), ("model", C.c_int32), ("coeffs", C.c_float * 5)]


class ModelLayout(C.Structure):
    _fields_ = [("table_bytes_canonical", C.c_uint64), ("table_bytes_physical", C.c_uint64),
                ("kernel_features", C.c_int32), ("kernel_dense_levels", C.c_int32), ("n_hashed_levels", C.c_int32),
                ("kernel_slots", C.c_int32), ("n_dense_levels", C.c_int32)]


class ScoreRecord(C.Structure):
    _fields_ = [("score", C.c_double), ("psnr", C.c_float), ("coverage", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples_nominal", C.c_uint64), ("samples_evaluated", C.c_uint64),
                ("wave_rounds", C.c_uint64), ("samples_live", C.c_uint64)]


class Intrinsics(C.Structure):
    _fields_ = [(n, C.c_double) for n in "fl_x fl_y cx cy k1 k2 p1 p2".split()] + [("w", C.c_int32), ("h", C.c_int32)]


class TrainOpts(C.Structure):
    _fields_ = [("n_rays", C.c_int32), ("n_samples", C.c_int32), ("lr", C.c_float), ("beta1", C.c_float),
                ("beta2", C.c_float), ("eps", C.c_float), ("l2_reg", C.c_float), ("min_T", C.c_float),
                ("seed", C.c_uint64), ("random_bg", C.c_int32), ("occ_every", C.c_int32), ("occ_decay", C.c_float),
                ("occ_sigma_thresh", C.c_float), ("target_samples", C.c_int32), ("patch_w", C.c_int32),
                ("patch_h", C.c_int32), ("step_mode", C.c_int32), ("deterministic", C.c_int32)]


_vp = C.c_void_p
_i = C.c_int
_P = C.POINTER

# name -> (restype, argtypes); must list every function include/prv.h declares
SIGNATURES = {
    "prv_create": (_i, [_P(_vp), _i]),
    "prv_destroy": (None, [_vp]),
    "prv_last_error": (C.c_char_p, [_vp]),
    "prv_abi_version": (_i, []),
    "prv_set_stream": (_i, [_vp, _vp]),
    "prv_runtime_shutdown": (_i, []),
    "prv_synchronize": (_i, [_vp]),
    "prv_set_coverage_weight": (_i, [_vp, C.c_double]),
    "prv_device_count": (_i, []),
    "prv_profile_begin": (_i, [_vp]),
    "prv_profile_end": (_i, [_vp, _P(C.c_double), _P(_i), _P(C.c_double), _P(_i)]),
    "prv_profile_render_launches": (_i, [_vp, _P(C.c_float), _i]),
    "prv_malloc": (_i, [_vp, _P(_vp), C.c_size_t]),
    "prv_free": (_i, [_vp, _vp]),
    "prv_memcpy_h2d": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "prv_memcpy_d2h": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "prv_model_sizes": (_i, [_P(FieldDesc), _P(C.c_uint64), _P(C.c_uint64), _P(C.c_uint64)]),
    "prv_model_load": (_i, [_vp, _i, _P(FieldDesc), _vp, _vp, _vp]),
    "prv_model_synthetic": (_i, [_vp, _i, _P(FieldDesc), C.c_uint64]),
    "prv_model_fresh": (_i, [_vp, _i, _P(FieldDesc), C.c_uint64]),
    "prv_model_export": (_i, [_vp, _i, _vp, _vp, _vp]),
    "prv_model_save_file": (_i, [_vp, _i, C.c_char_p]),
    "prv_model_load_ingp": (_i, [_vp, _i, C.c_char_p]),
    "prv_model_save_ingp": (_i, [_vp, _i, C.c_char_p]),
    "prv_model_desc": (_i, [_vp, _i, C.POINTER(FieldDesc)]),
    "prv_model_load_file": (_i, [_vp, _i, C.c_char_p]),
    "prv_cameras_from_json": (_i, [_vp, C.c_char_p, _P(_vp)]),
    "prv_cameras_from_matrices": (_i, [_vp, _vp, _i, C.c_double, _i, _i, C.c_double, _vp, _P(_vp)]),
    "prv_cameras_from_dataset_json": (_i, [_vp, C.c_char_p, _P(_vp)]),
    "prv_cameras_from_matrices_intr": (_i, [_vp, _vp, _i, _P(Intrinsics), C.c_double, _vp, _P(_vp)]),
    "prv_camset_lens": (_i, [_vp, _i, _vp]),
    "prv_camset_count": (_i, [_vp]),
    "prv_camset_size": (_i, [_vp, _P(_i), _P(_i)]),
    "prv_camset_get": (_i, [_vp, _i, _vp, _vp]),
    "prv_camset_destroy": (None, [_vp]),
    "prv_render": (_i, [_vp, _i, _vp, _vp, _i, _P(RenderOpts), _vp, _P(Stats)]),
    "prv_render_rgba8": (_i, [_vp, _i, _vp, _vp, _i, _P(RenderOpts), _vp, _P(Stats)]),
    "prv_first_hit": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, C.c_float, _vp]),
    "prv_precept": (_i, [_vp, _i, _vp, _i, _vp, _P(Rs2Intrinsics), C.c_float, _vp]),
    "prv_quantize_rgba8": (_i, [_vp, _vp, C.c_size_t, _vp, _vp]),
    "prv_score_ensemble_images": (_i, [_vp, _i, _vp, _i, _i, C.c_size_t, _vp]),
    "prv_score_psnr_images": (_i, [_vp, _vp, _vp, _i, C.c_size_t, _vp, _vp]),
    "prv_evaluate_images": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "prv_evaluate": (_i, [_vp, _i, _vp, _vp, _i, _P(RenderOpts), _vp, _P(C.c_double), _P(C.c_double)]),
    "prv_score_views": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _P(RenderOpts), _vp, _vp, _vp, _P(Stats)]),
    "prv_rank": (_i, [_vp, _vp, _i, _vp]),
    "prv_argmax": (_i, [_vp, _vp, _i]),
    "prv_splat_points": (_i, [_vp, _vp, _vp, C.c_size_t, C.c_double, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "prv_train_default_opts": (_i, [_P(TrainOpts)]),
    "prv_train_create": (_i, [_vp, _i, _vp, _vp, _i, _i, _P(TrainOpts), _P(_vp)]),
    "prv_train_steps": (_i, [_vp, _i, _vp]),
    "prv_train_steps_multi": (_i, [_vp, _i, _i, _vp]),
    "prv_train_debug_stamps": (_i, [_vp, _vp]),
    "prv_train_info": (_i, [_vp, _P(C.c_uint32), _P(C.c_uint64), _P(C.c_uint64)]),
    "prv_train_active_rays": (_i, [_vp]),
    "prv_train_destroy": (None, [_vp]),
    "prv_train_gradients": (_i, [_vp, _vp, _vp, _P(C.c_float)]),
    "prv_train_master": (_i, [_vp, _vp, _vp]),
    "prv_train_refresh_occupancy": (_i, [_vp]),
    "prv_comm_create": (_i, [_vp, _i, _i, C.c_char_p, C.c_char_p, C.POINTER(_vp)]),
    "prv_comm_destroy": (None, [_vp]),
    "prv_comm_rank": (_i, [_vp]),
    "prv_comm_world": (_i, [_vp]),
    "prv_comm_transport": (C.c_char_p, [_vp]),
    "prv_comm_library": (_i, [_vp, C.c_char_p, _i, _P(_i), C.c_char_p, _i]),
    "prv_comm_all_gather": (_i, [_vp, _vp, C.c_size_t, _vp]),
    "prv_comm_barrier": (_i, [_vp]),
    "prv_shard_views": (_i, [_i, _i, _i, _i, _vp, C.POINTER(_i)]),
    "prv_score_views_sharded": (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _i, C.POINTER(RenderOpts), _vp, _vp, C.POINTER(Stats)]),
    "prv_model_exchange": (_i, [_vp, _vp, _i, C.POINTER(FieldDesc)]),
    "prv_model_exchange_slots": (_i, [_vp, _vp, _i, _P(_i), _P(_i), C.POINTER(FieldDesc)]),
    "prv_debug_model_layout": (_i, [_vp, _i, C.POINTER(ModelLayout)]),
    "prv_debug_render_clock": (_i, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "prv_debug_raygen": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "prv_debug_encode": (_i, [_vp, _i, _vp, _i, _vp]),
    "prv_debug_field": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp]),
}

_lib = None


def load():
    """Load libprv_hip.so and bind every declared symbol; raise if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the render path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def device_code_digest(path=None):
    """sha256 (first 16 hex digits) of the .hip_fatbin section of libprv_hip.so: the gfx950 code objects themselves.  What
    a PMC pass measures (instructions per wave-round) is a property of the DEVICE code; host-side changes to the library do
    not touch it, so the profiles name this digest and bench.py compares it with the library it loads."""
    import hashlib
    import struct

    path = path or LIB_PATH
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        raise ValueError(f"{path} is not a 64-bit ELF file")
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    sections = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    stroff = sections[shstrndx][4]
    for name_off, _type, _flags, _addr, off, size, *_ in sections:
        end = data.index(b"\0", stroff + name_off)
        if data[stroff + name_off:end] == b".hip_fatbin":
            return hashlib.sha256(data[off:off + size]).hexdigest()[:16]
    raise ValueError(f"{path} has no .hip_fatbin section")
