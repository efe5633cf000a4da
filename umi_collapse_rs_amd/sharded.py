"""Multi-GPU driver: one process per GPU, alignment-position buckets sharded across
ranks, kept-read mask reassembled with an all-gatherv (RCCL over xGMI when the
process group is "nccl"; "gloo" in the CPU tests).

Buckets are independent units (src/deduplicate_sam.rs:207-233 shares nothing between
iterations but additive counters), so there is no data-path collective: the only
exchange is the final mask gather."""
import heapq

import numpy as np

from ._lib import UMI_ERR_NOMEM, UmiHipError


def partition_buckets_py(sizes, world_size):
    """Longest-processing-time assignment on cost n_b^2 + n_b (ties -> lower bucket index,
    then lower rank), in exact integer arithmetic: the reference implementation the library's
    umi_partition_buckets is checked against.  Returns the owner rank of every bucket."""
    sizes = [int(x) for x in sizes]
    cost = [x * x + x for x in sizes]
    order = sorted(range(len(sizes)), key=lambda b: (-sizes[b], b))
    heap = [(0, r) for r in range(world_size)]
    heapq.heapify(heap)
    owner = np.zeros(len(sizes), dtype=np.int64)
    for b in order:
        load, r = heapq.heappop(heap)
        owner[b] = r
        heapq.heappush(heap, (min(load + cost[b], 2 ** 64 - 1), r))
    return owner


def partition_buckets(sizes, world_size):
    """The library's bucket -> rank assignment (umi_partition_buckets: longest processing time
    first on n_b^2 + n_b, deterministic: every rank computes the same answer from the same
    sizes).  Returns a list of ascending bucket-index arrays, one per rank."""
    from .api import partition_buckets as lib_partition
    sizes = np.asarray(sizes, dtype=np.int64)
    off = np.zeros(len(sizes) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)
    owner = lib_partition(off, world_size)
    return [np.nonzero(owner == r)[0] for r in range(world_size)]


def shard_arrays(keys, nmask, freq, bucket_off, buckets):
    """Entries of the given buckets, concatenated in ascending bucket order.
    Returns (keys, nmask, freq, local bucket_off, global entry index of each local entry)."""
    bucket_off = np.asarray(bucket_off, dtype=np.int64)
    s, e = bucket_off[buckets], bucket_off[buckets + 1]
    n = e - s
    loff = np.zeros(len(buckets) + 1, dtype=np.uint64)
    loff[1:] = np.cumsum(n)
    total = int(loff[-1])
    if total:
        gidx = np.repeat(s - loff[:-1].astype(np.int64), n) + np.arange(total)
    else:
        gidx = np.zeros(0, dtype=np.int64)
    return (keys[gidx], None if nmask is None else nmask[gidx], freq[gidx], loff, gidx)


def allgatherv_mask(local_bits, counts, dist, device=None):
    """all-gatherv of bit-packed kept masks: every rank ends with all ranks' bytes.
    local_bits: uint8 tensor (packed bits of this rank's slice), counts: bytes per rank.
    Implemented as one all_gather on max-padded slices (RCCL has no native gatherv; the
    payload is <= 1 bit per unique UMI, latency-bound: SURVEY.md 8e)."""
    import torch
    world = dist.get_world_size()
    mx = max(1, int(max(counts)))
    buf = torch.zeros(mx, dtype=torch.uint8, device=local_bits.device)
    buf[: local_bits.numel()] = local_bits
    out = torch.empty(world * mx, dtype=torch.uint8, device=local_bits.device)
    dist.all_gather_into_tensor(out, buf)
    return [out[r * mx: r * mx + int(counts[r])] for r in range(world)]


class ShardedDedup:
    """Runs the batched hot path on this rank's buckets and gathers the global mask.

    Default (compute=None): the HIP path.  This rank's shard is uploaded once and stays on the
    GPU; a run is Context.dedup_batch_device on it, the kept mask packed to bits on the device
    (umi_pack_mask_device) and all-gathered as padded slices (RCCL over xGMI under "nccl").
    compute(keys, nmask, freq, bucket_off) -> kept uint8[n_local] replaces the per-GPU hot
    path with host code: the CPU tests (gloo, no GPU) pass the oracle here -- the product
    default never does."""

    def __init__(self, dist, compute=None, ctx=None, umi_len=None, k=1, percentage=0.5, algo=0,
                 adj_max_freq=0):
        self.dist = dist
        self.compute = compute
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.ctx = ctx
        self.params = dict(umi_len=umi_len, k=k, percentage=percentage, algo=algo, adj_max_freq=adj_max_freq)
        self._shard = None

    def shard(self, keys, nmask, freq, bucket_off):
        """Select this rank's buckets; on the HIP path, upload them (once) to the GPU."""
        bucket_off = np.asarray(bucket_off, dtype=np.uint64)
        sizes = np.diff(bucket_off.astype(np.int64))
        parts = partition_buckets(sizes, self.world)
        lk, lnm, lf, loff, _ = shard_arrays(keys, nmask, freq, bucket_off, parts[self.rank])
        sh = dict(parts=parts, bucket_off=bucket_off, n_local=[int(sizes[p].sum()) for p in parts], loff=loff)
        if self.compute is None:
            import torch
            if self.ctx is None:
                from .api import Context
                self.ctx = Context(torch.cuda.current_device())
            dev = torch.device("cuda", torch.cuda.current_device())
            sh["d_keys"] = torch.from_numpy(lk.view(np.int64)).to(dev)
            sh["d_nmask"] = None if lnm is None else torch.from_numpy(lnm.view(np.int64)).to(dev)
            sh["d_freq"] = torch.from_numpy(lf).to(dev)
            sh["d_kept"] = torch.zeros(max(1, len(lk)), dtype=torch.uint8, device=dev)
            mx = max(1, (max(sh["n_local"]) + 7) // 8)
            sh["d_bits"] = torch.zeros(mx, dtype=torch.uint8, device=dev)
            sh["d_all"] = torch.zeros(mx * self.world, dtype=torch.uint8, device=dev)
        else:
            sh["host"] = (lk, lnm, lf)
        self._shard = sh
        return sh

    def run_resident(self):
        """One pass over the resident shard.  Returns the packed masks of all ranks: a list of
        uint8 tensors (bits, little-endian per byte), slice r holding rank r's entries in its own
        shard order.  On the HIP path nothing leaves the GPU but the gathered bits."""
        import torch
        sh = self._shard
        counts = [(n + 7) // 8 for n in sh["n_local"]]
        if self.compute is None:
            n = sh["n_local"][self.rank]
            stream = torch.cuda.current_stream().cuda_stream
            p = self.params
            if n:
                # the call in two halves (umi_dedup_batch_device_begin / _end): packing and gathering are
                # enqueued behind the call's kernels while those run
                self.ctx.dedup_batch_device_begin(sh["d_keys"].data_ptr(),
                                                  0 if sh["d_nmask"] is None else sh["d_nmask"].data_ptr(),
                                                  sh["d_freq"].data_ptr(), sh["loff"], p["umi_len"], sh["d_kept"].data_ptr(), 0,
                                                  k=p["k"], percentage=p["percentage"], algo=p["algo"],
                                                  adj_max_freq=p["adj_max_freq"], stream=stream)
                self.ctx.pack_mask_device(sh["d_kept"].data_ptr(), n, sh["d_bits"].data_ptr(), stream=stream)
            mx = sh["d_bits"].numel()
            if self.dist.get_backend() == "nccl":  # RCCL over xGMI: device to device
                work = self.dist.all_gather_into_tensor(sh["d_all"], sh["d_bits"], async_op=True)
                if n:
                    self.ctx.dedup_batch_end()  # (a contract violation of this rank's shard surfaces here)
                work.wait()
                out = sh["d_all"]
            else:  # a rehearsal over gloo (several ranks on one GPU): the packed slice via the host
                if n:
                    self.ctx.dedup_batch_end()
                out = torch.empty(mx * self.world, dtype=torch.uint8)
                self.dist.all_gather_into_tensor(out, sh["d_bits"].cpu())
            return [out[r * mx: r * mx + counts[r]] for r in range(self.world)]
        lk, lnm, lf = sh["host"]
        kept_local = np.asarray(self.compute(lk, lnm, lf, sh["loff"]), dtype=np.uint8)
        backend = self.dist.get_backend()
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else "cpu"
        bits = torch.from_numpy(np.packbits(kept_local, bitorder="little")).to(dev)
        return allgatherv_mask(bits, counts, self.dist)

    def run(self, keys, nmask, freq, bucket_off):
        """shard + one pass + the global kept mask as a host array (uint8[N])."""
        import torch
        sh = self.shard(keys, nmask, freq, bucket_off)
        gathered = self.run_resident()
        kept = np.zeros(int(sh["bucket_off"][-1]), dtype=np.uint8)
        for r in range(self.world):
            if sh["n_local"][r] == 0:
                continue
            kr = np.unpackbits(gathered[r].cpu().numpy(), bitorder="little")[: sh["n_local"][r]]
            kept[_global_index(sh["bucket_off"], sh["parts"][r])] = kr
        # additive counters of deduplicate_sam.rs:217-219
        mine = kept[_global_index(sh["bucket_off"], sh["parts"][self.rank])] if sh["n_local"][self.rank] else kept[:0]
        dev = gathered[0].device
        tot = torch.tensor([int(mine.sum())], dtype=torch.int64, device=dev)
        self.dist.all_reduce(tot)
        assert int(tot.item()) == int(kept.sum())
        return kept


def _global_index(bucket_off, buckets):
    bucket_off = np.asarray(bucket_off, dtype=np.int64)
    s, e = bucket_off[buckets], bucket_off[buckets + 1]
    n = e - s
    loff = np.zeros(len(buckets) + 1, dtype=np.int64)
    loff[1:] = np.cumsum(n)
    total = int(loff[-1])
    if total == 0:
        return np.zeros(0, dtype=np.int64)
    return np.repeat(s - loff[:-1], n) + np.arange(total)


def allgatherv_edges(local_edges, dist):
    """all-gatherv of per-rank edge lists (int64 device tensors of different lengths): one
    all_gather of the lengths, one padded all_gather of the payload (RCCL over xGMI when the
    group is nccl).  Returns the concatenated list."""
    import torch
    world = dist.get_world_size()
    dev = local_edges.device
    cnt = torch.tensor([local_edges.numel()], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    mx = max(1, max(counts))
    buf = torch.zeros(mx, dtype=torch.int64, device=dev)
    buf[: local_edges.numel()] = local_edges
    out = torch.empty(world * mx, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, buf)
    return torch.cat([out[r * mx: r * mx + counts[r]] for r in range(world)])


def split_dedup_device(ctx, dist, d_keys, d_nmask, d_freq, bucket_off, umi_len, d_kept, d_root=None,
                       k=1, percentage=0.5, algo=0, adj_max_freq=0, edge_capacity=None):
    """One call's all-pairs work split over the ranks of `dist` (every rank holds the same
    inputs on its GPU): rank r evaluates every world-th tile task, the permitted-edge lists are
    all-gathered, every rank collapses the union.  For inputs that do not shard by buckets
    (one giant alignment position).  d_* are torch device tensors."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    n = d_keys.numel()
    stream = torch.cuda.current_stream().cuda_stream
    cap = int(edge_capacity or max(1 << 20, 4 * n))
    while True:
        buf = torch.empty(cap, dtype=torch.int64, device=d_keys.device)
        try:
            ne, st = ctx.pairs_partial_device(
                d_keys.data_ptr(), d_nmask.data_ptr() if d_nmask is not None else 0,
                d_freq.data_ptr(), bucket_off, umi_len, rank, world, buf.data_ptr(), cap, k=k,
                percentage=percentage, algo=algo, adj_max_freq=adj_max_freq, stream=stream)
            break
        except UmiHipError as e:  # UMI_ERR_NOMEM: the list did not fit; grow and redo
            if e.code != UMI_ERR_NOMEM:
                raise
            cap *= 4
    edges = allgatherv_edges(buf[:ne], dist)
    st2 = ctx.collapse_edges_device(n, edges.data_ptr() if edges.numel() else 0, edges.numel(),
                                    d_kept.data_ptr(), d_root.data_ptr() if d_root is not None else 0,
                                    algo=algo, stream=stream)
    st2["n_pairs"] = st["n_pairs"]
    st2["n_pairs_evaluated"] = st["n_pairs_evaluated"]
    st2["ms_pairs"] = st.get("ms_pairs", 0.0)
    return st2
