"""ctypes binding of libpcgx.so (include/pcgx.h).  No CPU fallback: if the
HIP library is missing or no GPU is present, calls fail loudly."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get("PCGX_LIB") or os.path.join(HERE, "libpcgx.so")  # PCGX_LIB: experiments with another build

PCGX_OK = 0
PCGX_E_NO_POINT = 1
PCGX_E_NOT_ENOUGH_PAIRS = 2
PCGX_E_BAD_FIELD = 3
PCGX_E_HIP = 4
PCGX_E_OOM = 5
PCGX_E_INVALID = 6
PCGX_E_OUT_OF_RANGE = 7
PCGX_E_TOO_LARGE = 8
PCGX_E_NEED_GRADIENT = 9
PCGX_E_SINGULAR = 10
PCGX_E_SYNTAX = 11
PCGX_E_EOF = 12
PCGX_E_CORRUPT = 13
PCGX_E_BAD_HEADER = 14
PCGX_E_RCCL = 15
PCGX_WEIGHT_ONE, PCGX_WEIGHT_CONSTANT, PCGX_WEIGHT_INVERSE, PCGX_WEIGHT_HUBER, PCGX_WEIGHT_TUKEY = range(5)
PCGX_SUMS_REFERENCE, PCGX_SUMS_F64_TREE, PCGX_SUMS_REFERENCE_CHAIN = range(3)  # pcgx_icp_params.sums_mode
PCGX_PCD_MAX_FIELDS = 64

PCGX_KNN_PRESORT = 1
PROF_ICP_WALK, PROF_KNN_WALK, PROF_VOXEL_ALL, PROF_SORT_SCATTER, PROF_ICP_GRID, PROF_KNN_GRID = range(6)
PROF_STRICT_TERMS, PROF_STRICT_SUM, PROF_STRICT_CHAIN, PROF_STRICT_JOB, PROF_ICP_LEFTOVER = 6, 7, 8, 9, 10


def prof_enable(on=True):
    """True / 1: time every launch; n > 1: every n-th launch of each kind; False / 0: off."""
    check(lib().pcgx_prof_enable(int(on)))


def prof_reset():
    check(lib().pcgx_prof_reset())


def prof_read(kind):
    """(total milliseconds, launches) of kernel class `kind` since prof_reset()."""
    ms, n = C.c_double(), C.c_int64()
    check(lib().pcgx_prof_read(kind, C.byref(ms), C.byref(n)))
    return ms.value, n.value


def prof_read_max(kind):
    """milliseconds of the longest single launch of kernel class `kind` since prof_reset()."""
    ms = C.c_double()
    check(lib().pcgx_prof_read_max(kind, C.byref(ms)))
    return ms.value


class PcgxError(RuntimeError):
    def __init__(self, code, msg):
        self.code = code
        super().__init__("pcgx error %d: %s" % (code, msg))


class ErrNoPoint(PcgxError):            # pc/minmax.go:11
    pass


class ErrNotEnoughPairs(PcgxError):     # icp/evaluator.go:16
    pass


class ErrNeedGradient(PcgxError):       # icp/icp.go:15
    pass


class ErrInvalidField(PcgxError):       # pc/pointcloud.go:115
    pass


class ErrSyntax(PcgxError):             # strconv.ErrSyntax (pc/io.go header / ascii tokens)
    pass


class ErrEOF(PcgxError):                # io.EOF / io.ErrUnexpectedEOF
    pass


class ErrDataCorruption(PcgxError):     # lzf.ErrDataCorruption
    pass


class ErrBadHeader(PcgxError):          # the errors.New cases of pc/io.go
    pass


class ErrSingular(PcgxError):           # point-to-plane extension: normal equations not positive definite
    pass


_ERR = {PCGX_E_NO_POINT: ErrNoPoint, PCGX_E_NOT_ENOUGH_PAIRS: ErrNotEnoughPairs,
        PCGX_E_NEED_GRADIENT: ErrNeedGradient, PCGX_E_BAD_FIELD: ErrInvalidField,
        PCGX_E_SINGULAR: ErrSingular, PCGX_E_SYNTAX: ErrSyntax, PCGX_E_EOF: ErrEOF,
        PCGX_E_CORRUPT: ErrDataCorruption, PCGX_E_BAD_HEADER: ErrBadHeader}


class IcpEvaluated(C.Structure):
    _fields_ = [("value", C.c_float), ("gradient", C.c_float * 6), ("dist_rms", C.c_float),
                ("num_pairs", C.c_int64)]


class IcpParams(C.Structure):
    _fields_ = [("max_dist", C.c_float), ("min_dist_sq", C.c_float), ("min_pairs", C.c_int32),
                ("weight", C.c_float * 6), ("threshold", C.c_float * 6), ("max_iteration", C.c_int32),
                ("weight_fn", C.c_int32), ("weight_fn_param", C.c_float), ("sums_mode", C.c_int32)]


class PcdHeader(C.Structure):  # pcgx_pcd_header
    _fields_ = [("version", C.c_float), ("n_fields", C.c_int32), ("fields", (C.c_char * 32) * 64),
                ("size", C.c_int32 * 64), ("type", C.c_char * 64), ("count", C.c_int32 * 64),
                ("width", C.c_int64), ("height", C.c_int64), ("n_viewpoint", C.c_int32),
                ("viewpoint", C.c_float * 16), ("points", C.c_int64), ("format", C.c_int32),
                ("stride", C.c_int64), ("data_offset", C.c_int64)]


class IcpStat(C.Structure):
    _fields_ = [("evaluated", IcpEvaluated), ("num_iteration", C.c_int32)]


# name -> (restype, argtypes); the complete export list of include/pcgx.h
_vp, _i64, _i32, _u32, _f32, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_uint32, C.c_float, C.c_size_t
ABI_VERSION = 6   # include/pcgx.h PCGX_ABI_VERSION

SIGNATURES = {
    "pcgx_init": (_i32, [_i32]),
    "pcgx_shutdown": (_i32, []),
    "pcgx_last_error": (_i32, [C.c_char_p, _sz]),
    "pcgx_version": (C.c_char_p, []),
    "pcgx_abi_version": (_i32, []),
    "pcgx_icp_params_init": (_i32, [_vp, C.c_size_t]),
    "pcgx_sync": (_i32, [_vp]),
    "pcgx_prof_enable": (_i32, [_i32]),
    "pcgx_prof_read": (_i32, [_i32, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "pcgx_prof_reset": (_i32, []),
    "pcgx_prof_read_max": (_i32, [_i32, C.POINTER(C.c_double)]),
    "pcgx_debug_walk_stats": (_i32, [_vp, _vp, _i64, _f32, _i32, _vp, _vp, _vp]),
    "pcgx_dev_alloc": (_i32, [_sz, C.POINTER(_vp)]),
    "pcgx_dev_free": (_i32, [_vp]),
    "pcgx_dev_upload": (_i32, [_vp, _vp, _sz]),
    "pcgx_dev_download": (_i32, [_vp, _vp, _sz]),
    "pcgx_kdtree_build": (_i32, [_vp, _i64, _i32, _i32, C.POINTER(_vp)]),
    "pcgx_kdtree_free": (_i32, [_vp]),
    "pcgx_kdtree_len": (_i32, [_vp, C.POINTER(_i64)]),
    "pcgx_kdtree_max_depth": (_i32, [_vp, C.POINTER(_i32)]),
    "pcgx_kdtree_inorder": (_i32, [_vp, _vp]),
    "pcgx_kdtree_points": (_i32, [_vp, _vp, _i64, _vp]),
    "pcgx_debug_icp_grid_stats": (_i32, [_vp, _vp, C.POINTER(_i64)]),
    "pcgx_debug_grid_stats": (_i32, [_vp, _vp, _i64, C.c_float, C.POINTER(_i64)]),
    "pcgx_debug_grid_cert": (_i32, [_vp, _vp, _i64]),
    "pcgx_kdtree_dump": (_i32, [_vp, _vp, _i64, C.POINTER(_i64)]),
    "pcgx_kdtree_delete_points": (_i32, [_vp, _vp, _i64]),
    "pcgx_kdtree_live_count": (_i32, [_vp, C.POINTER(_i64)]),
    "pcgx_kdtree_nearest_batch": (_i32, [_vp, _vp, _i64, _f32, _f32, _vp, _vp]),
    "pcgx_kdtree_range_count": (_i32, [_vp, _vp, _i64, _f32, _vp]),
    "pcgx_kdtree_range_fill": (_i32, [_vp, _vp, _i64, _f32, _vp, _vp, _vp]),
    "pcgx_kdtree_nearest_batch_dev": (_i32, [_vp, _vp, _i64, _f32, _f32, _u32, _vp, _vp, _vp]),
    "pcgx_minmax": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "pcgx_voxel_filter": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, C.POINTER(_i64)]),
    "pcgx_voxel_filter_dev": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, C.POINTER(_i64), _vp]),
    "pcgx_voxel_filter_sharded_dev": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, C.POINTER(_i64), _vp]),
    "pcgx_voxel_filter_sharded": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, C.POINTER(_i64)]),
    "pcgx_icp_pairs": (_i32, [_vp, _vp, _i64, _f32, _f32, _vp, _vp, _vp, C.POINTER(_i64)]),
    "pcgx_icp_evaluate": (_i32, [_vp, _vp, _i64, _f32, _f32, _i32, C.POINTER(IcpEvaluated)]),
    "pcgx_icp_evaluate_params": (_i32, [_vp, _vp, _i64, C.POINTER(IcpParams), C.POINTER(IcpEvaluated)]),
    "pcgx_icp_finish_evaluate": (_i32, [_vp, _i32, C.POINTER(IcpEvaluated)]),
    "pcgx_icp_update": (_i32, [C.POINTER(IcpParams), C.POINTER(_i32), _vp, _vp, C.POINTER(_i32)]),
    "pcgx_rodrigues": (_i32, [_vp, _vp]),
    "pcgx_mat4_mul": (_i32, [_vp, _vp, _vp]),
    "pcgx_mat4_transform": (_i32, [_vp, _vp, _i64, _vp]),
    "pcgx_icp_fit": (_i32, [_vp, _vp, _i64, C.POINTER(IcpParams), _vp, C.POINTER(IcpStat)]),
    "pcgx_icp_session_create": (_i32, [_vp, _vp, _i64, _i32, C.POINTER(IcpParams), _vp, C.POINTER(_vp)]),
    "pcgx_icp_session_free": (_i32, [_vp]),
    "pcgx_icp_session_reset": (_i32, [_vp, _vp]),
    "pcgx_icp_session_set_pose": (_i32, [_vp, _vp, _i32, _vp]),
    "pcgx_icp_session_read_sums": (_i32, [_vp, _vp, _vp]),
    "pcgx_icp_session_partials": (_i32, [_vp, _vp]),
    "pcgx_icp_session_update": (_i32, [_vp, _vp]),
    "pcgx_icp_session_step": (_i32, [_vp, _vp]),
    "pcgx_icp_session_set_strict": (_i32, [_vp, _i32]),
    "pcgx_debug_call_stats": (_i32, [_vp, _i32]),
    "pcgx_debug_voxel_stats": (_i32, [_vp, _i32]),
    "pcgx_debug_shard_stats": (_i32, [_vp, _i32]),
    "pcgx_debug_ring_kinds": (_i32, [_vp, _i32]),
    "pcgx_debug_host_walks": (_i32, [_vp, _i32]),
    "pcgx_debug_icp_one_launch": (_i32, [_vp, _i32]),
    "pcgx_comm_unique_id": (_i32, [_vp]),
    "pcgx_comm_init": (_i32, [_i32, _i32, _vp, C.POINTER(_vp)]),
    "pcgx_comm_init_callback": (_i32, [_i32, _i32, _vp, _vp, C.POINTER(_vp)]),
    "pcgx_comm_free": (_i32, [_vp]),
    "pcgx_comm_rank": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "pcgx_comm_allreduce_f64": (_i32, [_vp, _vp, _i32, _vp]),
    "pcgx_comm_allreduce_host_f64": (_i32, [_vp, _vp, _i32]),
    "pcgx_icp_fit_multi": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcgx_init_devices": (_i32, [_i32, _vp]),
    "pcgx_set_device": (_i32, [_i32]),
    "pcgx_get_device": (_i32, [_vp, _vp]),
    "pcgx_icp_session_step_sharded": (_i32, [_vp, _vp, _vp]),
    "pcgx_icp_fit_sharded": (_i32, [_vp, _vp, _i64, C.POINTER(IcpParams), _vp, _vp, C.POINTER(IcpStat)]),
    "pcgx_debug_icp_strict_stats": (_i32, [_vp, _vp, _vp]),
    "pcgx_debug_strict_sum_host": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "pcgx_debug_strict_sum_dev": (_i32, [_vp, _i64, _vp, _vp]),
    "pcgx_icp_session_result": (_i32, [_vp, _vp, _vp, C.POINTER(IcpStat), C.POINTER(_i32)]),
    "pcgx_bucket_grid_build": (_i32, [_vp, _i64, _i32, _i32, _f32, _vp, _vp, C.POINTER(_vp)]),
    "pcgx_bucket_grid_free": (_i32, [_vp]),
    "pcgx_bucket_grid_counts": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "pcgx_bucket_grid_addr": (_i32, [_vp, _vp, C.POINTER(_i64), C.POINTER(_i32)]),
    "pcgx_bucket_grid_point_addrs": (_i32, [_vp, _vp]),
    "pcgx_bucket_grid_get_by_addr": (_i32, [_vp, _i64, _vp, _i64, C.POINTER(_i64)]),
    "pcgx_bucket_grid_get": (_i32, [_vp, _vp, _vp, _i64, C.POINTER(_i64)]),
    "pcgx_bucket_grid_indice": (_i32, [_vp, _vp]),
    "pcgx_bucket_grid_components": (_i32, [_vp, _vp]),
    "pcgx_bucket_grid_segment": (_i32, [_vp, _vp, _vp, _i64, C.POINTER(_i64)]),
    "pcgx_bucket_grid_segment_bfs": (_i32, [_vp, _vp, _vp, _i64, C.POINTER(_i64)]),
    "pcgx_region_growing_components": (_i32, [_vp, _vp, _f32, _vp]),
    "pcgx_region_growing_segment": (_i32, [_vp, _vp, _vp, _vp, _f32, _vp, _i64, C.POINTER(_i64)]),
    "pcgx_pcd_unmarshal_header": (_i32, [_vp, _sz, C.POINTER(PcdHeader)]),
    "pcgx_pcd_unmarshal": (_i32, [_vp, _sz, C.POINTER(PcdHeader), _vp]),
    "pcgx_pcd_unmarshal_dev": (_i32, [_vp, _sz, C.POINTER(PcdHeader), _vp, _vp]),
    "pcgx_pcd_marshal": (_i32, [C.POINTER(PcdHeader), _vp, _vp, _sz, C.POINTER(_sz)]),
    "pcgx_region_growing_segment_bfs": (_i32, [_vp, _vp, _vp, _f32, _vp, _i64, C.POINTER(_i64)]),
    "pcgx_icp_plane_session_create": (_i32, [_vp, _vp, _vp, _i64, _i32, C.POINTER(IcpParams), _f32, _vp,
                                             C.POINTER(_vp)]),
    "pcgx_icp_session_sums_count": (_i32, [_vp, C.POINTER(_i32)]),
    "pcgx_icp_session_read_sums_n": (_i32, [_vp, _vp, _i32, _vp]),
    "pcgx_icp_session_hessian": (_i32, [_vp, _vp, _vp]),
    "pcgx_icp_plane_fit": (_i32, [_vp, _vp, _vp, _i64, C.POINTER(IcpParams), _f32, _vp, C.POINTER(IcpStat), _vp]),
    "pcgx_icp_plane_finish_evaluate": (_i32, [_vp, _i32, C.POINTER(IcpEvaluated), _vp]),
    "pcgx_icp_gauss_newton_update": (_i32, [C.POINTER(IcpParams), _f32, C.POINTER(_i32), _vp, _vp, _vp,
                                            C.POINTER(_i32)]),
}

_lib = None


def lib():
    """Loads libpcgx.so.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise ImportError("pcgol_amd/libpcgx.so is missing: run `python -c 'import __graft_entry__ as g; "
                              "g.build()'` (hipcc, gfx950). There is no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own HIP runtime (same soname as /opt/rocm's).  One process
        if not os.environ.get("PCGX_LIB"):
            from . import build as _build
            if _build.stale():
                raise ImportError("pcgol_amd/libpcgx.so was not built from the sources beside it (csrc/ changed since): "
                                  "run `python -c 'import __graft_entry__ as g; g.build()'`")
        # must use ONE runtime: when torch is installed, load it first so that libpcgx.so binds to
        # the runtime torch (and RCCL) will use; loading libpcgx first was seen to make a later
        # ProcessGroupNCCL report "no GPUs found".  Without torch the system runtime is used.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(SO)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.pcgx_abi_version() != ABI_VERSION:   # the struct layouts below belong to ONE version of include/pcgx.h
            raise ImportError("libpcgx.so speaks ABI version %d, this binding %d: rebuild (python -c 'import __graft_entry__ as g; g.build()')"
                              % (L.pcgx_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def last_error():
    buf = C.create_string_buffer(1024)
    lib().pcgx_last_error(buf, 1024)
    return buf.value.decode(errors="replace")


def check(rc):
    if rc != PCGX_OK:
        raise _ERR.get(rc, PcgxError)(rc, last_error())


def ptr(a):
    """void* of a numpy array (must be C-contiguous) or a raw integer address."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)
