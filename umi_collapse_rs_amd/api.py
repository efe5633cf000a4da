"""Host-side mirror of the reference's plugin interface for the hot path, on top
of the C ABI (include/umihip.h):

    trait DataStruct  (src/data/mod.rs:11-17)   -> HipNaive
    trait Algorithm   (src/algo/mod.rs:13-20)    -> Directional, Adjacency
    bucket loop       (src/deduplicate_sam.rs:207-233) -> Context.dedup_batch

Same names, argument meaning and error behaviour as the reference (where the
reference panics, these raise).  Everything computes on the GPU through
libumihip.so; there is no CPU path in this package."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import (UMI_ALGO_ADJACENCY, UMI_ALGO_DIRECTIONAL, Stats, UmiHipError, check, load, ptr)


def to_bitset(umis, umi_len=None):
    """utils::to_bitset (src/utils/mod.rs:63-83) for a batch: list of str/bytes or a
    uint8 array [n*umi_len] -> (keys u64[n], nmask u64[n])."""
    if isinstance(umis, np.ndarray):
        raw = np.ascontiguousarray(umis, dtype=np.uint8)
        assert umi_len, "umi_len is required for a raw byte array"
        n = raw.size // umi_len
    else:
        bs = [u.encode() if isinstance(u, str) else bytes(u) for u in umis]
        n = len(bs)
        umi_len = umi_len or (len(bs[0]) if bs else 1)
        if any(len(b) != umi_len for b in bs):
            raise ValueError("all UMIs of a run have the same length (umi_length)")
        raw = np.frombuffer(b"".join(bs), dtype=np.uint8)
    keys = np.zeros(n, dtype=np.uint64)
    nmask = np.zeros(n, dtype=np.uint64)
    check(load().umi_encode_umis(ptr(raw, C.c_uint8), n, umi_len, ptr(keys, C.c_uint64),
                                 ptr(nmask, C.c_uint64)))
    return keys, nmask


def to_bitset_wide(umis, umi_len):
    """to_bitset (src/utils/mod.rs:63-83) for UMIs of any length up to 85: (keys, nmask) uint64 [n, n_words]."""
    w = (3 * umi_len + 63) // 64
    buf = np.frombuffer("".join(umis).encode(), dtype=np.uint8)
    keys = np.zeros((len(umis), w), dtype=np.uint64)
    nm = np.zeros((len(umis), w), dtype=np.uint64)
    check(load().umi_encode_umis_wide(ptr(buf, C.c_uint8), len(umis), umi_len, w, ptr(keys, C.c_uint64),
                                      ptr(nm, C.c_uint64)))
    return keys, nm


def partition_buckets(bucket_off, n_ranks):
    """umi_partition_buckets: owner rank of every bucket (uint32[n_buckets]), the assignment the
    multi-device context uses -- for hosts that run one process per GPU."""
    bucket_off = np.ascontiguousarray(bucket_off, dtype=np.uint64)
    nb = max(0, len(bucket_off) - 1)
    owner = np.zeros(nb, dtype=np.uint32)
    check(load().umi_partition_buckets(ptr(bucket_off, C.c_uint64), nb, n_ranks, ptr(owner, C.c_uint32)))
    return owner


class Context:
    """One GPU context (umi_ctx): one per process in a one-process-per-GPU job; or, with a list
    of device ids, one context over several GPUs of the node (umi_ctx_create_multi: dedup_batch
    shards its buckets over them)."""

    def __init__(self, device_id=0, profile=False):
        self._h = C.c_void_p()
        if isinstance(device_id, (list, tuple)):
            ids = (C.c_int * len(device_id))(*[int(d) for d in device_id])
            check(load().umi_ctx_create_multi(ids, len(device_id), C.byref(self._h)))
        else:
            check(load().umi_ctx_create(device_id, C.byref(self._h)))
        self.device_id = device_id
        if profile:
            self.set_option("profile", 1)

    def set_option(self, name, value):
        check(load().umi_ctx_set_option(self._h, name.encode(), int(value)))

    def close(self):
        if getattr(self, "_h", None):
            load().umi_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dedup_batch(self, keys, nmask, freq, bucket_off, umi_len, k=1, percentage=0.5,
                    algo=UMI_ALGO_DIRECTIONAL, adj_max_freq=0, want_root=True):
        """Host-buffer batched call.  Returns (kept u8[N], root u32[N] or None, stats dict)."""
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        freq = np.ascontiguousarray(freq, dtype=np.int32)
        bucket_off = np.ascontiguousarray(bucket_off, dtype=np.uint64)
        nm = None if nmask is None else np.ascontiguousarray(nmask, dtype=np.uint64)
        n = len(keys)
        if len(freq) != n or (nm is not None and len(nm) != n):
            raise ValueError("keys/nmask/freq lengths differ")
        if len(bucket_off) < 1 or (len(bucket_off) > 1 and int(bucket_off[-1]) != n):
            raise ValueError("bucket_off[-1] must equal len(keys)")
        kept = np.zeros(n, dtype=np.uint8)
        root = np.zeros(n, dtype=np.uint32) if want_root else None
        st = Stats()
        check(load().umi_dedup_batch(self._h, ptr(keys, C.c_uint64), ptr(nm, C.c_uint64),
                                     ptr(freq, C.c_int32), ptr(bucket_off, C.c_uint64),
                                     len(bucket_off) - 1, umi_len, k, percentage, algo,
                                     adj_max_freq, ptr(kept, C.c_uint8), ptr(root, C.c_uint32),
                                     C.byref(st)))
        return kept, root, st.as_dict()

    def dedup_batch_wide(self, keys, nmask, freq, bucket_off, umi_len, k=1, percentage=0.5,
                         algo=UMI_ALGO_DIRECTIONAL, adj_max_freq=0, want_root=True):
        """Batched call for keys of several words (umi_len > 21): keys / nmask uint64 [N, n_words]."""
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        n, w = keys.shape
        nm = None if nmask is None else np.ascontiguousarray(nmask, dtype=np.uint64)
        freq = np.ascontiguousarray(freq, dtype=np.int32)
        bucket_off = np.ascontiguousarray(bucket_off, dtype=np.uint64)
        kept = np.zeros(max(1, n), dtype=np.uint8)
        root = np.zeros(max(1, n), dtype=np.uint32) if want_root else None
        st = Stats()
        check(load().umi_dedup_batch_wide(self._h, ptr(keys, C.c_uint64), ptr(nm, C.c_uint64), w,
                                          ptr(freq, C.c_int32), ptr(bucket_off, C.c_uint64), len(bucket_off) - 1,
                                          umi_len, k, percentage, algo, adj_max_freq, ptr(kept, C.c_uint8),
                                          ptr(root, C.c_uint32), C.byref(st)))
        return kept[:n], (root[:n] if want_root else None), st.as_dict()

    def stage_reads(self, align_key, umi_bytes, score, umi_len, merge=1, align_key_bits=64):
        """Read staging on the device (host arrays in and out): reads in file order ->
        dict(keys, nmask, freq, rep, bucket_off) in canonical order, the batched path's input
        (src/deduplicate_sam.rs:148-176 + the rank order of src/algo/directional.rs:67-72)."""
        align_key = np.ascontiguousarray(align_key, dtype=np.uint64)
        umi_bytes = np.ascontiguousarray(umi_bytes, dtype=np.uint8)
        sc = None if score is None else np.ascontiguousarray(score, dtype=np.int32)
        n = len(align_key)
        assert len(umi_bytes) == n * umi_len
        m = max(1, n)
        keys, nm, rep = (np.zeros(m, np.uint64) for _ in range(3))
        freq = np.zeros(m, np.int32)
        boff = np.zeros(m + 1, np.uint64)
        ne, nb = C.c_uint64(0), C.c_uint64(0)
        check(load().umi_stage_reads(self._h, ptr(align_key, C.c_uint64), align_key_bits,
                                        ptr(umi_bytes, C.c_uint8), ptr(sc, C.c_int32), n, umi_len, merge,
                                        ptr(keys, C.c_uint64), ptr(nm, C.c_uint64), ptr(freq, C.c_int32),
                                        ptr(rep, C.c_uint64), ptr(boff, C.c_uint64), C.byref(ne), C.byref(nb)))
        e, b = int(ne.value), int(nb.value)
        return dict(keys=keys[:e], nmask=nm[:e], freq=freq[:e], rep=rep[:e], bucket_off=boff[:b + 1])

    def stage_reads_wide(self, align_key, umi_bytes, score, umi_len, merge=1, align_key_bits=64):
        """stage_reads for UMIs of any length up to 85 bases: keys / nmask come back as uint64
        [n_entries, n_words] (the input of dedup_batch_wide; one column for umi_len <= 21)."""
        align_key = np.ascontiguousarray(align_key, dtype=np.uint64)
        umi_bytes = np.ascontiguousarray(umi_bytes, dtype=np.uint8)
        sc = None if score is None else np.ascontiguousarray(score, dtype=np.int32)
        n, w = len(align_key), (3 * umi_len + 63) // 64
        assert len(umi_bytes) == n * umi_len
        m = max(1, n)
        keys, nm = np.zeros((m, w), np.uint64), np.zeros((m, w), np.uint64)
        rep, freq, boff = np.zeros(m, np.uint64), np.zeros(m, np.int32), np.zeros(m + 1, np.uint64)
        ne, nb = C.c_uint64(0), C.c_uint64(0)
        check(load().umi_stage_reads_wide(self._h, ptr(align_key, C.c_uint64), align_key_bits, ptr(umi_bytes, C.c_uint8),
                                          ptr(sc, C.c_int32), n, umi_len, w, merge, ptr(keys, C.c_uint64),
                                          ptr(nm, C.c_uint64), ptr(freq, C.c_int32), ptr(rep, C.c_uint64),
                                          ptr(boff, C.c_uint64), C.byref(ne), C.byref(nb)))
        e, b = int(ne.value), int(nb.value)
        return dict(keys=keys[:e], nmask=nm[:e], freq=freq[:e], rep=rep[:e], bucket_off=boff[:b + 1])

    def stage_reads_device(self, d_align_key, d_umi, d_score, n_reads, umi_len, d_keys, d_nmask, d_freq,
                           d_rep, d_bucket_off, merge=1, align_key_bits=64, stream=0):
        """The same with everything in device memory (raw pointers); returns (n_entries, n_buckets)."""
        ne, nb = C.c_uint64(0), C.c_uint64(0)
        check(load().umi_stage_reads_device(self._h, d_align_key, align_key_bits, d_umi, d_score or None,
                                               n_reads, umi_len, merge, d_keys, d_nmask or None, d_freq, d_rep,
                                               d_bucket_off, C.byref(ne), C.byref(nb), stream or None))
        return int(ne.value), int(nb.value)

    def dedup_batch_device(self, d_keys, d_nmask, d_freq, bucket_off, umi_len, d_kept, d_root=0,
                           k=1, percentage=0.5, algo=UMI_ALGO_DIRECTIONAL, adj_max_freq=0,
                           stream=0, d_bucket_off=0):
        """Device-pointer batched call (integers = device addresses, e.g. tensor.data_ptr()).
        d_bucket_off: device copy of bucket_off, if the caller keeps one (no upload inside)."""
        bucket_off = np.ascontiguousarray(bucket_off, dtype=np.uint64)
        st = Stats()
        check(load().umi_dedup_batch_device_table(self._h, d_keys, d_nmask or None, d_freq,
                                                  ptr(bucket_off, C.c_uint64), d_bucket_off or None,
                                                  len(bucket_off) - 1, umi_len, k, percentage, algo,
                                                  adj_max_freq, d_kept, d_root or None, stream or None,
                                                  C.byref(st)))
        return st.as_dict()


    def dedup_batch_device_begin(self, d_keys, d_nmask, d_freq, bucket_off, umi_len, d_kept, d_root=0,
                                 k=1, percentage=0.5, algo=UMI_ALGO_DIRECTIONAL, adj_max_freq=0,
                                 stream=0, d_bucket_off=0):
        """umi_dedup_batch_device_begin: the device-pointer call enqueued on `stream`; where every position is
        the fused kernel's it returns without waiting (d_kept / d_root final in stream order), else it
        runs to its end.  dedup_batch_end() waits and returns the stats."""
        bucket_off = np.ascontiguousarray(bucket_off, dtype=np.uint64)
        check(load().umi_dedup_batch_device_begin(self._h, d_keys, d_nmask or None, d_freq,
                                                  ptr(bucket_off, C.c_uint64), d_bucket_off or None,
                                                  len(bucket_off) - 1, umi_len, k, percentage, algo,
                                                  adj_max_freq, d_kept, d_root or None, stream or None))

    def dedup_batch_end(self):
        st = Stats()
        check(load().umi_dedup_batch_end(self._h, C.byref(st)))
        return st.as_dict()

    def dedup_batch_device_multi(self, shards, umi_len, slice_bytes, k=1, percentage=0.5, algo=UMI_ALGO_DIRECTIONAL,
                                 adj_max_freq=0, gather=True):
        """umi_dedup_batch_device_multi on a multi-device context: shards = one dict per device with the
        device addresses d_keys, d_freq, d_kept (and optionally d_nmask, d_root), the host array
        bucket_off, and d_bits_all (n_devices * slice_bytes bytes on that device) for the RCCL
        all-gather of the packed kept masks.  Returns the merged stats."""
        nd = len(shards)
        boffs = [np.ascontiguousarray(sh["bucket_off"], dtype=np.uint64) for sh in shards]

        def ptrs(name, required=True):
            vals = [sh.get(name) or None for sh in shards]
            if not required and all(v is None for v in vals):
                return None
            return (C.c_void_p * nd)(*vals)
        st = Stats()
        off_arr = (_lib._u64p * nd)(*[ptr(b, C.c_uint64) for b in boffs])
        nb_arr = (C.c_uint64 * nd)(*[len(b) - 1 for b in boffs])
        check(load().umi_dedup_batch_device_multi(self._h, ptrs("d_keys"), ptrs("d_nmask", False), ptrs("d_freq"), off_arr,
                                                  nb_arr, umi_len, k, percentage, algo, adj_max_freq, ptrs("d_kept"),
                                                  ptrs("d_root", False), ptrs("d_bits_all") if gather else None,
                                                  slice_bytes, C.byref(st)))
        return st.as_dict()

    def dedup_batch_wide_device(self, d_keys, d_nmask, n_words, d_freq, bucket_off, umi_len, d_kept, d_root=0, k=1,
                                percentage=0.5, algo=UMI_ALGO_DIRECTIONAL, adj_max_freq=0, stream=0):
        """Device-pointer batched call for keys of n_words words per entry (umi_len > 21)."""
        bucket_off = np.ascontiguousarray(bucket_off, dtype=np.uint64)
        st = Stats()
        check(load().umi_dedup_batch_wide_device(self._h, d_keys, d_nmask or None, n_words, d_freq, ptr(bucket_off, C.c_uint64),
                                                 len(bucket_off) - 1, umi_len, k, percentage, algo, adj_max_freq, d_kept,
                                                 d_root or None, stream or None, C.byref(st)))
        return st.as_dict()

    def pack_mask_device(self, d_kept, n, d_bits, stream=0):
        """kept bytes -> bits on the device (umi_pack_mask_device), enqueued on `stream`."""
        check(load().umi_pack_mask_device(self._h, d_kept, n, d_bits, stream or None))

    def pairs_partial_device(self, d_keys, d_nmask, d_freq, bucket_off, umi_len, part, n_parts,
                             d_edges, edge_capacity, k=1, percentage=0.5,
                             algo=UMI_ALGO_DIRECTIONAL, adj_max_freq=0, stream=0):
        """This rank's share (every n_parts-th tile task) of the permitted-edge list, written
        to the caller's device buffer d_edges (uint64 entries).  Returns (n_edges, stats);
        raises UmiHipError(UMI_ERR_NOMEM) if edge_capacity is too small (n_edges in the text)."""
        bucket_off = np.ascontiguousarray(bucket_off, dtype=np.uint64)
        st = Stats()
        n_edges = C.c_uint64(0)
        check(load().umi_pairs_partial_device(self._h, d_keys, d_nmask or None, d_freq,
                                              ptr(bucket_off, C.c_uint64), len(bucket_off) - 1,
                                              umi_len, k, percentage, algo, adj_max_freq, part,
                                              n_parts, d_edges, edge_capacity, C.byref(n_edges),
                                              stream or None, C.byref(st)))
        return int(n_edges.value), st.as_dict()

    def collapse_edges_device(self, n, d_edges, n_edges, d_kept, d_root=0,
                              algo=UMI_ALGO_DIRECTIONAL, stream=0):
        """Collapse of a gathered edge list over n entries; fills d_kept / d_root."""
        st = Stats()
        check(load().umi_collapse_edges_device(self._h, n, d_edges or None, n_edges, algo, d_kept,
                                               d_root or None, stream or None, C.byref(st)))
        return st.as_dict()


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class HipNaive:
    """DataStruct (src/data/mod.rs:11-17) with Naive's semantics (src/data/naive.rs),
    neighbour lists built on the GPU at construction.  UMIs are ASCII strings here
    (the reference's &BitSet keys); insertion order of `umi_freq` is kept."""

    def __init__(self):  # Default
        self._h = None
        self._index = {}
        self._umis = []

    @classmethod
    def new(cls, umi_freq, umi_length, max_edits, ctx=None):
        """DataStruct::new(umi_freq: HashMap<&BitSet,i32>, umi_length, max_edits)"""
        self = cls()
        ctx = ctx or default_context()
        self._ctx = ctx
        self._umis = list(umi_freq.keys())
        self._index = {u: i for i, u in enumerate(self._umis)}
        freq = np.array([umi_freq[u] for u in self._umis], dtype=np.int32)
        self._n = len(self._umis)
        h = C.c_void_p()
        if umi_length > _lib.UMI_MAX_UMI_LEN:  # keys of several words (bitset.rs:17-27)
            words = (3 * umi_length + 63) // 64
            keys, nmask = to_bitset_wide([u if isinstance(u, str) else u.decode() for u in self._umis], umi_length) \
                if self._umis else (np.zeros((0, words), np.uint64), np.zeros((0, words), np.uint64))
            check(load().umi_data_new_wide(ctx._h, ptr(keys, C.c_uint64), ptr(nmask, C.c_uint64) if nmask.any() else None,
                                           words, ptr(freq, C.c_int32), self._n, umi_length, max_edits, C.byref(h)))
            self._h = h
            return self
        keys, nmask = to_bitset(self._umis, umi_length) if self._umis else (
            np.zeros(0, np.uint64), np.zeros(0, np.uint64))
        check(load().umi_data_new(ctx._h, ptr(keys, C.c_uint64),
                                  ptr(nmask, C.c_uint64) if nmask.any() else None,
                                  ptr(freq, C.c_int32), self._n, umi_length, max_edits,
                                  C.byref(h)))
        self._h = h
        return self

    def remove_near(self, umi, k, max_freq):
        """-> set of removed UMIs (HashSet<&BitSet>)"""
        out = np.zeros(max(1, self._n), dtype=np.uint32)
        cnt = C.c_uint32(0)
        check(load().umi_data_remove_near(self._h, self._index[umi], k, max_freq,
                                          ptr(out, C.c_uint32), C.byref(cnt)))
        return {self._umis[i] for i in out[:cnt.value]}

    def contains(self, umi):
        i = self._index.get(umi)
        if i is None:
            return False
        rc = load().umi_data_contains(self._h, i)
        if rc < 0:
            check(rc)
        return rc == 1

    def stats(self):
        return {}

    def __del__(self):
        if getattr(self, "_h", None):
            load().umi_data_free(self._h)
            self._h = None


class ReadFreq:
    """src/utils/read_freq.rs:4-13"""
    __slots__ = ("read", "freq")

    def __init__(self, read, freq):
        self.read, self.freq = read, freq


def _f32_as_i32(x):
    x = np.float32(x)
    if np.isnan(x):
        return 0
    return int(max(-2 ** 31, min(2 ** 31 - 1, int(x))))


class Directional:
    """src/algo/directional.rs:15-91.  apply() follows the reference's control flow
    over any DataStruct (default HipNaive): stable freq-descending sort, root loop,
    neighbour visit (explicit stack instead of the reference's recursion)."""

    def __init__(self, k=1, percentage=0.5, track_cluster=False):
        self.k, self.percentage, self.track_cluster = k, np.float32(percentage), track_cluster

    def _visit_and_remove(self, start_umi, reads, data, cluster):
        stack = [start_umi]
        while stack:
            u = stack.pop()
            f1 = (reads[u].freq + 1 + 2 ** 31) % 2 ** 32 - 2 ** 31  # i32 wrap, as a release build
            threshold = _f32_as_i32(self.percentage * np.float32(f1))
            near = data.remove_near(u, self.k, threshold)
            if cluster is not None:
                cluster.extend(near)
            stack.extend(v for v in near if v != u)

    def apply(self, reads, tracker, umi_length, data_struct=HipNaive):
        """reads: dict umi -> ReadFreq in first-appearance order.  Returns list of reads."""
        data_member = {umi: rf.freq for umi, rf in reads.items()}
        umi_freqs = sorted(reads.items(), key=lambda kv: -kv[1].freq)  # stable
        data = data_struct.new(data_member, umi_length, self.k)
        res = []
        for umi, rf in umi_freqs:
            if data.contains(umi):
                cluster = [] if (self.track_cluster and tracker is not None) else None
                self._visit_and_remove(umi, reads, data, cluster)
                if cluster is not None:
                    tracker[umi] = cluster
                res.append(rf.read)
        return res


class Adjacency:
    """src/algo/adjacency.rs:15-63 (remove_near(umi, k, 0): reference behaviour)."""

    def __init__(self, k=1, percentage=0.5, track_cluster=False, max_freq=0):
        self.k, self.percentage, self.track_cluster = k, percentage, track_cluster
        self.max_freq = max_freq

    def apply(self, reads, tracker, umi_length, data_struct=HipNaive):
        freq = sorted(reads.items(), key=lambda kv: -kv[1].freq)
        m = {umi: rf.freq for umi, rf in reads.items()}
        data = data_struct.new(m, umi_length, self.k)
        res = []
        for umi, rf in freq:
            if data.contains(umi):
                near = data.remove_near(umi, self.k, self.max_freq)
                if tracker is not None:
                    tracker[umi] = sorted(near)
                res.append(rf.read)
        return res
