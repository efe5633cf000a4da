"""ctypes binding of libumihip.so (include/umihip.h).  No fallback: if the HIP
library is missing or no gfx950 device is visible, calls raise UmiHipError."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (UMIHIP_LIB: tuning experiments load another build of the same library)
LIB_PATH = os.environ.get("UMIHIP_LIB") or os.path.join(_HERE, "libumihip.so")

UMI_OK = 0
UMI_ERR_ARG, UMI_ERR_HIP, UMI_ERR_ORDER, UMI_ERR_NOMEM, UMI_ERR_NODEV, UMI_ERR_CHAR = (
    -1, -2, -3, -4, -5, -6)
UMI_ALGO_DIRECTIONAL, UMI_ALGO_ADJACENCY = 0, 1
UMI_MAX_UMI_LEN = 21


class UmiHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("umihip error %d: %s" % (code, msg))
        self.code = code


class Stats(C.Structure):
    _fields_ = [("n_umis", C.c_uint64), ("n_buckets", C.c_uint64), ("max_bucket", C.c_uint64),
                ("n_kept", C.c_uint64), ("n_pairs", C.c_uint64),
                ("n_pairs_evaluated", C.c_uint64), ("n_candidates", C.c_uint64),
                ("n_edges", C.c_uint64), ("n_rounds", C.c_uint32),
                ("n_pair_launches", C.c_uint32), ("ms_total", C.c_float),
                ("ms_prep", C.c_float), ("ms_pairs", C.c_float), ("ms_collapse", C.c_float),
                ("ms_finalize", C.c_float), ("ms_kernel", C.c_float), ("kernel_id", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_u8p, _u32p, _u64p, _i32p = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                             C.POINTER(C.c_uint64), C.POINTER(C.c_int32))

# name -> (restype, argtypes); every symbol include/umihip.h declares
SIGNATURES = {
    "umi_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "umi_ctx_create_multi": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "umi_ctx_device_count": (C.c_int, [C.c_void_p]),
    "umi_partition_buckets": (C.c_int, [_u64p, C.c_uint64, C.c_uint32, _u32p]),
    "umi_ctx_destroy": (None, [C.c_void_p]),
    "umi_last_error": (C.c_char_p, []),
    "umi_ctx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "umi_abi_version": (C.c_int, []),
    "umi_encode_umis": (C.c_int, [_u8p, C.c_uint64, C.c_int, _u64p, _u64p]),
    "umi_encode_umis_wide": (C.c_int, [_u8p, C.c_uint64, C.c_int, C.c_int, _u64p, _u64p]),
    "umi_dedup_batch_wide": (C.c_int, [C.c_void_p, _u64p, _u64p, C.c_int, _i32p, _u64p, C.c_uint64, C.c_int,
                                       C.c_int, C.c_float, C.c_int, C.c_int32, _u8p, _u32p, C.POINTER(Stats)]),
    "umi_dedup_batch_wide_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, _u64p,
                                              C.c_uint64, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    "umi_stage_reads_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64,
                                         C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, _u64p, _u64p, C.c_void_p]),
    "umi_stage_reads": (C.c_int, [C.c_void_p, _u64p, C.c_int, _u8p, _i32p, C.c_uint64, C.c_int, C.c_int,
                                  _u64p, _u64p, _i32p, _u64p, _u64p, _u64p, _u64p]),
    "umi_stage_reads_wide_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64,
                                              C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, _u64p, _u64p, C.c_void_p]),
    "umi_stage_reads_wide": (C.c_int, [C.c_void_p, _u64p, C.c_int, _u8p, _i32p, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                       _u64p, _u64p, _i32p, _u64p, _u64p, _u64p, _u64p]),
    "umi_dedup_batch": (C.c_int, [C.c_void_p, _u64p, _u64p, _i32p, _u64p, C.c_uint64, C.c_int,
                                  C.c_int, C.c_float, C.c_int, C.c_int32, _u8p, _u32p,
                                  C.POINTER(Stats)]),
    "umi_dedup_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _u64p,
                                         C.c_uint64, C.c_int, C.c_int, C.c_float, C.c_int,
                                         C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(Stats)]),
    "umi_dedup_batch_device_table": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _u64p, C.c_void_p,
                                               C.c_uint64, C.c_int, C.c_int, C.c_float, C.c_int,
                                               C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.POINTER(Stats)]),
    "umi_dedup_batch_device_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _u64p, C.c_void_p,
                                               C.c_uint64, C.c_int, C.c_int, C.c_float, C.c_int,
                                               C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "umi_dedup_batch_end": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "umi_pack_mask_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "umi_dedup_batch_device_multi": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_void_p), C.POINTER(_u64p), _u64p, C.c_int, C.c_int,
                                               C.c_float, C.c_int, C.c_int32, C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_uint64,
                                               C.POINTER(Stats)]),
    "umi_pairs_partial_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _u64p,
                                           C.c_uint64, C.c_int, C.c_int, C.c_float, C.c_int,
                                           C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p,
                                           C.c_uint64, _u64p, C.c_void_p, C.POINTER(Stats)]),
    "umi_collapse_edges_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    "umi_data_new": (C.c_int, [C.c_void_p, _u64p, _u64p, _i32p, C.c_uint32, C.c_int, C.c_int,
                               C.POINTER(C.c_void_p)]),
    "umi_data_new_wide": (C.c_int, [C.c_void_p, _u64p, _u64p, C.c_int, _i32p, C.c_uint32, C.c_int, C.c_int,
                                    C.POINTER(C.c_void_p)]),
    "umi_data_remove_near": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_int32, _u32p, _u32p]),
    "umi_data_contains": (C.c_int, [C.c_void_p, C.c_uint32]),
    "umi_data_free": (None, [C.c_void_p]),
}

_lib = None


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (same SONAME as /opt/rocm's).
    Whichever copy is mapped first serves the whole process, and a process that starts on
    the system copy cannot initialise torch.cuda afterwards.  So when torch is installed,
    map its copy first; without torch the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def load():
    """Load libumihip.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UmiHipError(UMI_ERR_NODEV, "libumihip.so not built (run `make` or "
                              "__graft_entry__.build()); there is no CPU fallback")
        _preload_hip_runtime()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != UMI_OK:
        raise UmiHipError(rc, load().umi_last_error().decode(errors="replace"))


def ptr(a, ty):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None
