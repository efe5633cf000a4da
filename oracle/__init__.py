"""ctypes bindings of the CPU oracle (oracle/umi_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product path (umi_collapse_rs_amd) never does.
Parity unpinned by the reference (it has no tests) -- see umi_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libumi_oracle.so")
ORC_MAXW = 4


class Bitset(C.Structure):
    _fields_ = [("nwords", C.c_int), ("has_n", C.c_int),
                ("bits", C.c_uint64 * ORC_MAXW), ("nbits", C.c_uint64 * ORC_MAXW)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u8p, u32p, u64p, i32p = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_uint64), C.POINTER(C.c_int32))
        bsp = C.POINTER(Bitset)
        L.orc_to_bitset.argtypes = [C.c_char_p, C.c_int, bsp]
        L.orc_to_bitset.restype = C.c_int
        L.orc_bitset_hash.argtypes = [bsp]
        L.orc_bitset_hash.restype = C.c_int32
        L.orc_bit_count_xor.argtypes = [bsp, bsp]
        L.orc_bit_count_xor.restype = C.c_int32
        L.orc_umi_dist.argtypes = [bsp, bsp]
        L.orc_umi_dist.restype = C.c_int32
        L.orc_threshold.argtypes = [C.c_float, C.c_int32]
        L.orc_threshold.restype = C.c_int32
        L.orc_avg_qual.argtypes = [u8p, C.c_int]
        L.orc_avg_qual.restype = C.c_int32
        L.orc_naive_new.argtypes = [bsp, i32p, C.c_uint32]
        L.orc_naive_new.restype = C.c_void_p
        L.orc_naive_free.argtypes = [C.c_void_p]
        L.orc_naive_remove_near.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_int32, u32p]
        L.orc_naive_remove_near.restype = C.c_uint32
        L.orc_naive_contains.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_naive_contains.restype = C.c_int
        L.orc_directional_apply.argtypes = [bsp, i32p, C.c_uint32, C.c_int32, C.c_float, u32p,
                                            u32p, u64p]
        L.orc_directional_apply.restype = C.c_uint32
        L.orc_adjacency_apply.argtypes = [bsp, i32p, C.c_uint32, C.c_int32, C.c_int32, u32p,
                                          u32p, u64p]
        L.orc_adjacency_apply.restype = C.c_uint32
        L.orc_dedup_batch.argtypes = [u64p, u64p, i32p, u64p, C.c_uint64, C.c_int, C.c_int32,
                                      C.c_float, C.c_int, C.c_int32, u8p, u32p, u64p]
        L.orc_dedup_batch.restype = C.c_int
        L.orc_dedup_batch_wide.argtypes = [u64p, u64p, C.c_int, i32p, u64p, C.c_uint64, C.c_int, C.c_int32,
                                           C.c_float, C.c_int, C.c_int32, u8p, u32p, C.POINTER(C.c_uint64)]
        L.orc_dedup_batch_wide.restype = C.c_int
        L.orc_stage_reads.argtypes = [u32p, u8p, i32p, C.c_uint64, C.c_int, C.c_int, u64p, u64p,
                                      i32p, u64p, u64p, u64p, u64p]
        L.orc_stage_reads.restype = C.c_int
        _lib = L
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


def to_bitset(s):
    """ASCII UMI -> Bitset; raises ValueError where the reference panics."""
    if isinstance(s, str):
        s = s.encode()
    b = Bitset()
    if lib().orc_to_bitset(s, len(s), C.byref(b)) != 0:
        raise ValueError("Unknown character in UMI sequence")
    return b


def bits_of(b):
    return [int(b.bits[i]) for i in range(b.nwords)]


def nbits_of(b):
    return [int(b.nbits[i]) for i in range(b.nwords)] if b.has_n else None


def bitset_hash(b):
    return int(lib().orc_bitset_hash(C.byref(b)))


def bit_count_xor(a, b):
    return int(lib().orc_bit_count_xor(C.byref(a), C.byref(b)))


def umi_dist(a, b):
    return int(lib().orc_umi_dist(C.byref(a), C.byref(b)))


def threshold(p, f):
    return int(lib().orc_threshold(p, f))


def avg_qual(quals):
    q = np.ascontiguousarray(quals, dtype=np.uint8)
    return int(lib().orc_avg_qual(_p(q, C.c_uint8), len(q)))


def encode_keys(umis):
    """list of ASCII UMIs (all <= 21 bp) -> (keys u64, nmask u64)."""
    keys = np.zeros(len(umis), dtype=np.uint64)
    nm = np.zeros(len(umis), dtype=np.uint64)
    for i, u in enumerate(umis):
        b = to_bitset(u)
        assert b.nwords <= 1
        keys[i] = b.bits[0]
        nm[i] = b.nbits[0] if b.has_n else 0
    return keys, nm


def _bitset_array(umis):
    arr = (Bitset * max(1, len(umis)))()
    for i, u in enumerate(umis):
        arr[i] = to_bitset(u)
    return arr


def apply_strings(umis, freq, k, algo="dir", percentage=0.5, adj_max_freq=0):
    """Run the collapse on UMIs given in first-appearance order.
    Returns (survivor input indices in output order, root_of per input, dist_calls)."""
    n = len(umis)
    arr = _bitset_array(umis)
    f = np.ascontiguousarray(freq, dtype=np.int32)
    out = np.zeros(max(1, n), dtype=np.uint32)
    rof = np.arange(max(1, n), dtype=np.uint32)
    calls = C.c_uint64(0)
    if algo == "dir":
        ns = lib().orc_directional_apply(arr, _p(f, C.c_int32), n, k, percentage,
                                         _p(out, C.c_uint32), _p(rof, C.c_uint32),
                                         C.byref(calls))
    else:
        ns = lib().orc_adjacency_apply(arr, _p(f, C.c_int32), n, k, adj_max_freq,
                                       _p(out, C.c_uint32), _p(rof, C.c_uint32), C.byref(calls))
    return out[:ns].tolist(), rof[:n].tolist(), int(calls.value)


class Naive:
    """src/data/naive.rs over ASCII UMIs (test helper)."""

    def __init__(self, umis, freq):
        self._arr = _bitset_array(umis)
        self._f = np.ascontiguousarray(freq, dtype=np.int32)
        self.n = len(umis)
        self._h = lib().orc_naive_new(self._arr, _p(self._f, C.c_int32), self.n)

    def remove_near(self, idx, k, max_freq):
        out = np.zeros(max(1, self.n), dtype=np.uint32)
        c = lib().orc_naive_remove_near(self._h, idx, k, max_freq, _p(out, C.c_uint32))
        return out[:c].tolist()

    def contains(self, idx):
        return bool(lib().orc_naive_contains(self._h, idx))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_naive_free(self._h)
            self._h = None


def dedup_batch(keys, nmask, freq, bucket_off, umi_len, k, percentage=0.5, algo=0,
                adj_max_freq=0):
    """Batched oracle: returns (kept u8[N], root u32[N], dist_calls)."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    freq = np.ascontiguousarray(freq, dtype=np.int32)
    bucket_off = np.ascontiguousarray(bucket_off, dtype=np.uint64)
    nm = None if nmask is None else np.ascontiguousarray(nmask, dtype=np.uint64)
    n = len(keys)
    kept = np.ones(max(1, n), dtype=np.uint8)
    root = np.arange(max(1, n), dtype=np.uint32)
    calls = C.c_uint64(0)
    rc = lib().orc_dedup_batch(_p(keys, C.c_uint64), _p(nm, C.c_uint64), _p(freq, C.c_int32),
                               _p(bucket_off, C.c_uint64), len(bucket_off) - 1, umi_len, k,
                               percentage, algo, adj_max_freq, _p(kept, C.c_uint8),
                               _p(root, C.c_uint32), C.byref(calls))
    if rc != 0:
        raise ValueError("orc_dedup_batch failed: %d" % rc)
    return kept[:n], root[:n], int(calls.value)


def encode_keys_wide(umis):
    """list of ASCII UMIs of one length (any, up to 85 bp) -> (keys, nmask) uint64 [n, n_words]
    through the restated to_bitset."""
    first = to_bitset(umis[0])
    w = first.nwords
    keys = np.zeros((len(umis), w), dtype=np.uint64)
    nm = np.zeros((len(umis), w), dtype=np.uint64)
    for i, u in enumerate(umis):
        b = to_bitset(u)
        assert b.nwords == w
        for j in range(w):
            keys[i, j] = b.bits[j]
            nm[i, j] = b.nbits[j] if b.has_n else 0
    return keys, nm


def dedup_batch_wide(keys, nmask, freq, bucket_off, umi_len, k, percentage=0.5, algo=0, adj_max_freq=0):
    """Batched oracle for keys of several words: keys / nmask uint64 [N, n_words]."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    n, w = keys.shape
    freq = np.ascontiguousarray(freq, dtype=np.int32)
    bucket_off = np.ascontiguousarray(bucket_off, dtype=np.uint64)
    nm = None if nmask is None else np.ascontiguousarray(nmask, dtype=np.uint64)
    kept = np.ones(max(1, n), dtype=np.uint8)
    root = np.arange(max(1, n), dtype=np.uint32)
    calls = C.c_uint64(0)
    rc = lib().orc_dedup_batch_wide(_p(keys, C.c_uint64), _p(nm, C.c_uint64), w, _p(freq, C.c_int32),
                                    _p(bucket_off, C.c_uint64), len(bucket_off) - 1, umi_len, k,
                                    percentage, algo, adj_max_freq, _p(kept, C.c_uint8),
                                    _p(root, C.c_uint32), C.byref(calls))
    if rc != 0:
        raise ValueError("orc_dedup_batch_wide failed: %d" % rc)
    return kept[:n], root[:n], int(calls.value)


def stage_reads(bucket_id, umi_bytes, score, umi_len, merge=1):
    """Staging oracle.  umi_bytes: uint8[n_reads*umi_len].
    Returns dict(keys,nmask,freq,rep,bucket_off)."""
    bucket_id = np.ascontiguousarray(bucket_id, dtype=np.uint32)
    umi_bytes = np.ascontiguousarray(umi_bytes, dtype=np.uint8)
    sc = None if score is None else np.ascontiguousarray(score, dtype=np.int32)
    n = len(bucket_id)
    keys = np.zeros(max(1, n), dtype=np.uint64)
    nm = np.zeros(max(1, n), dtype=np.uint64)
    freq = np.zeros(max(1, n), dtype=np.int32)
    rep = np.zeros(max(1, n), dtype=np.uint64)
    boff = np.zeros(n + 2, dtype=np.uint64)
    n_out, b_out = C.c_uint64(0), C.c_uint64(0)
    rc = lib().orc_stage_reads(_p(bucket_id, C.c_uint32), _p(umi_bytes, C.c_uint8),
                               _p(sc, C.c_int32), n, umi_len, merge, _p(keys, C.c_uint64),
                               _p(nm, C.c_uint64), _p(freq, C.c_int32), _p(rep, C.c_uint64),
                               _p(boff, C.c_uint64), C.byref(n_out), C.byref(b_out))
    if rc != 0:
        raise ValueError("Unknown character in UMI sequence")
    m, b = int(n_out.value), int(b_out.value)
    return dict(keys=keys[:m], nmask=nm[:m], freq=freq[:m], rep=rep[:m], bucket_off=boff[:b + 1])
