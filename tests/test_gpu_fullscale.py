"""BASELINE configs 4 and 5 at their STATED size on the one GPU of the test box: the whole
100 M-read / 10^6-position job (and 50 M reads, 20-bp, k = 2) through the 8-way multi-device
context, `umi.Context([0] * 8)` -- eight workers, each with its own context, stream and workspace,
all on this card -- i.e. the bucket loop src/deduplicate_sam.rs:207-233 over a whole node's input.

The oracle is O(n_b^2) scalar code per bucket and there are 10^6 buckets: the checks are
  P1  structure of kept / root over all ~59 M entries (tests/test_gpu_fullsize.py),
  P5  the oracle, bit for bit, on every ~1000th bucket,
  P4  the stitched mask equals what a single-device context gives for each rank's share on its own
      (the shares are what bench.py's ranks generate: seed + 1000 * rank).
Generation dominates the run time (numpy, ~10 s per 12.5 M reads): shares are made on a thread pool."""
import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle as orc
from test_gpu_fullsize import check_structure

pytestmark = pytest.mark.gpu
N_RANKS = 8


def _job(cfg_seed, reads_per_rank, umi_len):
    from umi_collapse_rs_amd import synth

    def share(r):
        return synth.config3(seed=cfg_seed + 1000 * r, n_reads=reads_per_rank, n_positions=reads_per_rank // 100,
                             umi_len=umi_len)
    with ThreadPoolExecutor(min(N_RANKS, max(1, len(os.sched_getaffinity(0))))) as ex:
        shares = list(ex.map(share, range(N_RANKS)))
    keys = np.concatenate([s["keys"] for s in shares])
    freq = np.concatenate([s["freq"] for s in shares])
    offs, base, bounds = [np.zeros(1, np.uint64)], np.uint64(0), [0]
    for s in shares:
        offs.append(s["bucket_off"][1:] + base)
        base = base + s["bucket_off"][-1]
        bounds.append(bounds[-1] + len(s["bucket_off"]) - 1)
    return keys, freq, np.concatenate(offs), bounds, shares


@pytest.mark.parametrize("name,cfg_seed,reads_per_rank,umi_len,k", [
    ("config 4: 100 M reads, 12-bp, 10^6 positions", 4, 12_500_000, 12, 1),
    ("config 5: 50 M reads, 20-bp, k = 2", 5, 6_250_000, 20, 2)])
def test_eight_gpu_configs_at_full_size_through_the_multi_device_context(name, cfg_seed, reads_per_rank, umi_len, k,
                                                                         capfd):
    import umi_collapse_rs_amd as umi
    t0 = time.time()
    keys, freq, off, bounds, shares = _job(cfg_seed, reads_per_rank, umi_len)
    t_gen = time.time() - t0
    n_buckets = len(off) - 1
    assert n_buckets == N_RANKS * (reads_per_rank // 100)
    os.environ["UMIHIP_TIMING"] = "1"  # the library prints every worker's wall split of a sharded call
    multi = umi.Context([0] * N_RANKS)
    try:
        t0 = time.time()
        kept, root, st = multi.dedup_batch(keys, None, freq, off, umi_len, k=k, percentage=0.5)
        t_call = time.time() - t0
        t0 = time.time()
        kept2, root2, st2 = multi.dedup_batch(keys, None, freq, off, umi_len, k=k, percentage=0.5)  # buffers warm
        t_call2 = time.time() - t0
    finally:
        multi.close()
        os.environ.pop("UMIHIP_TIMING", None)
    split = [l for l in capfd.readouterr().err.splitlines() if l.startswith("umihip multi:")]
    print("\n%s: %d entries in %d buckets; generated in %.1f s; sharded call %.2f s cold, %.2f s warm" % (
        name, len(keys), n_buckets, t_gen, t_call, t_call2))
    for l in split[-N_RANKS:]:
        print("   ", l)
    assert (kept2 == kept).all() and (root2 == root).all()
    assert st["n_umis"] == len(keys) and st["n_buckets"] == n_buckets and st["n_kept"] == int(kept.sum())
    sizes = np.diff(off.astype(np.int64))
    assert st["n_pairs"] == int((sizes * (sizes - 1) // 2).sum()) and st["max_bucket"] == int(sizes.max())
    check_structure(kept, root, off)  # P1
    for b in range(0, n_buckets, 997):  # P5
        s, e = int(off[b]), int(off[b + 1])
        ok, oroot, _ = orc.dedup_batch(keys[s:e], None, freq[s:e], [0, e - s], umi_len, k)
        assert (kept[s:e] == ok).all() and (root[s:e] - s == oroot).all(), b
    single = umi.Context(0)  # P4
    try:
        for r in range(N_RANKS):
            sh = shares[r]
            skept, sroot, sst = single.dedup_batch(sh["keys"], None, sh["freq"], sh["bucket_off"], umi_len, k=k)
            e0, e1 = int(off[bounds[r]]), int(off[bounds[r + 1]])
            assert (kept[e0:e1] == skept).all(), r
            assert (root[e0:e1] - np.uint32(e0) == sroot).all(), r
    finally:
        single.close()
