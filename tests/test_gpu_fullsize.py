"""BASELINE.json's full sizes on the GPU, checked through size-independent properties
(the oracle is O(n^2) scalar code: hours at these sizes).

Properties of the reference semantics (src/algo/directional.rs:30-91 over
src/data/naive.rs:26-40) that hold for any input:
  P1  kept[v] == (root[v] == v); root[root[v]] == root[v]; root[v] <= v; same bucket.
  P2  fixed point of the min-rank recurrence: root[v] == min(v, min over permitted in-edges
      u->v of root[u]), checked exactly on a random sample of v against ALL entries of
      the bucket (numpy popcount over the whole key array per sampled v).
  P3  k = 0 keeps every entry; adjacency (reference max_freq 0) keeps every entry.
  P4  bucket independence: a bucket deduplicated alone gives the same mask as inside the
      batch; the result does not depend on where the bucket sits in the batch.
  P5  an oracle run on a subsample of whole buckets agrees bit for bit."""
import numpy as np
import pytest

import oracle as orc
from helpers import is_dev_build, usable

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import umi_collapse_rs_amd as umi
    c = umi.Context(0)
    yield c
    c.close()


def popcount64(x):
    x = x - ((x >> np.uint64(1)) & np.uint64(0x5555555555555555))
    x = (x & np.uint64(0x3333333333333333)) + ((x >> np.uint64(2)) & np.uint64(0x3333333333333333))
    x = (x + (x >> np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
    return (x * np.uint64(0x0101010101010101)) >> np.uint64(56)


def thr_of(freq, p):
    return (np.float32(p) * (freq + 1).astype(np.float32)).astype(np.int32)


def check_structure(kept, root, off):
    n = len(kept)
    idx = np.arange(n, dtype=np.uint32)
    assert ((kept == 1) == (root == idx)).all()
    assert (root[root] == root).all()
    assert (root <= idx).all()
    bucket = np.searchsorted(off.astype(np.int64), idx.astype(np.int64), side="right")
    assert (bucket[root] == bucket).all()


def check_fixed_point(keys, freq, root, off, k, p, sample, rng):
    """Exact recurrence check for sampled entries (N-free keys: dist = popcount(xor)/2)."""
    thr = thr_of(freq, p)
    for v in sample:
        b = np.searchsorted(off.astype(np.int64), v, side="right") - 1
        s, e = int(off[b]), int(off[b + 1])
        d = popcount64(keys[s:e] ^ keys[v]) // np.uint64(2)
        ok = (d <= k) & (freq[v] <= thr[s:e])
        ok[v - s] = False
        cand = root[s:e][ok]
        want = min(int(v), int(cand.min())) if len(cand) else int(v)
        assert int(root[v]) == want, (v, int(root[v]), want)


def test_config2_one_million_reads_single_position(ctx):
    from umi_collapse_rs_amd import synth
    st = synth.config2(seed=2, n_reads=1_000_000, umi_len=12)
    keys, freq, off = st["keys"], st["freq"], st["bucket_off"]
    assert len(keys) > 950_000
    kept, root, stats = ctx.dedup_batch(keys, None, freq, off, 12, k=1, percentage=0.5)
    assert stats["n_pairs"] == len(keys) * (len(keys) - 1) // 2
    assert stats["n_kept"] == int(kept.sum())
    check_structure(kept, root, off)
    rng = np.random.default_rng(0)
    sample = np.concatenate([rng.choice(len(keys), 150, replace=False),
                             np.nonzero(kept == 0)[0][:50], np.nonzero(kept == 1)[0][-50:]])
    check_fixed_point(keys, freq, root, off, 1, 0.5, sample, rng)
    # P3
    kept0, root0, _ = ctx.dedup_batch(keys, None, freq, off, 12, k=0)
    assert kept0.all() and (root0 == np.arange(len(keys))).all()
    kept_adj, _, _ = ctx.dedup_batch(keys, None, freq, off, 12, k=1, algo=1)
    assert kept_adj.all()
    # the true all-pairs popcount tile kernel on the same input (every one of the W pairs is
    # evaluated) gives the same answer as the segment index
    import umi_collapse_rs_amd as umi
    c2 = umi.Context(0)
    c2.set_option("seg_index", 0)
    if is_dev_build():  # (the shipped library's all-pairs path is the popcount tile kernel anyway)
        c2.set_option("bitslice", 0)
    try:
        kept2, root2, _ = c2.dedup_batch(keys, None, freq, off, 12, k=1, percentage=0.5)
    finally:
        c2.close()
    assert (kept2 == kept).all() and (root2 == root).all()
    # the default path evaluates the pairs inside the sub-buckets of its two 6-base parts:
    # 2 x 4096 sub-buckets of ~237 entries
    assert stats["n_pairs_evaluated"] < stats["n_pairs"] // 1000
    if not is_dev_build():
        return
    # key-sorted tiles with range pruning: same answer, a fraction of the comparisons
    c3 = umi.Context(0)
    c3.set_option("prune", 1)
    try:
        kept3, root3, st3 = c3.dedup_batch(keys, None, freq, off, 12, k=1, percentage=0.5)
    finally:
        c3.close()
    assert (kept3 == kept).all() and (root3 == root).all()
    assert st3["n_pairs_evaluated"] < stats["n_pairs"] // 2
    # the key-sorted scan + walk of the earlier versions: only the column tiles whose high bases
    # leave a row within k
    c4 = umi.Context(0)
    c4.set_option("seg_index", 0)
    try:
        kept4, root4, st4 = c4.dedup_batch(keys, None, freq, off, 12, k=1, percentage=0.5)
    finally:
        c4.close()
    assert (kept4 == kept).all() and (root4 == root).all() and st4["n_edges"] == stats["n_edges"]
    assert st4["n_pairs_evaluated"] < stats["n_pairs"] // 3


@pytest.mark.parametrize("L,k,n_reads", [(13, 1, 1_300_000), (12, 2, 600_000), (11, 0, 500_000)])
def test_table_kernel_against_the_other_tile_kernels(L, k, n_reads):
    """Deep positions the oracle cannot walk in test time: the two item walks over the key-sorted
    bucket (columns of a run across the lanes; register tables -- also in their 16-base shape,
    which needs > 10^6 entries to be chosen), the key-sorted mask kernel and the unsorted one must
    agree bit for bit, and the result must be a fixed point (P2)."""
    import umi_collapse_rs_amd as umi
    from umi_collapse_rs_amd import synth
    st = synth.config2(seed=40 + L, n_reads=n_reads, umi_len=L)
    keys, freq, off = st["keys"], st["freq"], st["bucket_off"]
    outs = []
    for opts in ({}, {"seg_index": 0}, {"seg_index": 0, "bs_transposed": 0}, {"seg_index": 0, "bs_tables": 0},
                 {"seg_index": 0, "bs_sorted": 0}):
        if not usable(opts):
            continue
        c = umi.Context(0)
        try:
            for name, v in opts.items():
                c.set_option(name, v)
            outs.append(c.dedup_batch(keys, None, freq, off, L, k=k, percentage=0.5))
        finally:
            c.close()
    kept, root, stats = outs[0]
    check_structure(kept, root, off)
    for kept_o, root_o, st_o in outs[1:]:
        assert (kept_o == kept).all() and (root_o == root).all()
        assert st_o["n_edges"] == stats["n_edges"]
    if k > 0:
        rng = np.random.default_rng(1)
        sample = np.concatenate([rng.choice(len(keys), 100, replace=False), np.nonzero(kept == 0)[0][:50]])
        check_fixed_point(keys, freq, root, off, k, 0.5, sample, rng)


def test_config3_many_small_buckets(ctx):
    from umi_collapse_rs_amd import synth
    st = synth.config3(seed=3, n_reads=10_000_000, n_positions=100_000, umi_len=12)
    keys, freq, off = st["keys"], st["freq"], st["bucket_off"]
    assert len(off) - 1 == 100_000
    kept, root, stats = ctx.dedup_batch(keys, None, freq, off, 12, k=1, percentage=0.5)
    check_structure(kept, root, off)
    # P5: oracle on every 97th bucket
    pick = np.arange(0, 100_000, 97)
    for b in pick:
        s, e = int(off[b]), int(off[b + 1])
        ok, oroot, _ = orc.dedup_batch(keys[s:e], None, freq[s:e], [0, e - s], 12, 1)
        assert (kept[s:e] == ok).all() and (root[s:e] - s == oroot).all(), b
    # P4: reversed bucket order gives the same per-bucket masks
    sizes = np.diff(off.astype(np.int64))
    order = np.arange(len(sizes))[::-1]
    roff = np.zeros(len(off), np.uint64)
    roff[1:] = np.cumsum(sizes[order])
    gidx = np.concatenate([np.arange(off[b], off[b + 1], dtype=np.int64) for b in order[:2000]])
    n_sub = len(gidx)
    sub_off = roff[:2001]
    kept_r, _, _ = ctx.dedup_batch(keys[gidx], None, freq[gidx], sub_off, 12, k=1)
    assert (kept_r == kept[gidx]).all()
    assert n_sub == int(sub_off[-1])


def test_config5_shape_20bp_k2(ctx):
    from umi_collapse_rs_amd import synth
    st = synth.config3(seed=5, n_reads=2_000_000, n_positions=20_000, umi_len=20)
    keys, freq, off = st["keys"], st["freq"], st["bucket_off"]
    kept, root, stats = ctx.dedup_batch(keys, None, freq, off, 20, k=2, percentage=0.5)
    check_structure(kept, root, off)
    for b in range(0, 20_000, 53):
        s, e = int(off[b]), int(off[b + 1])
        ok, oroot, _ = orc.dedup_batch(keys[s:e], None, freq[s:e], [0, e - s], 20, 2)
        assert (kept[s:e] == ok).all() and (root[s:e] - s == oroot).all(), b
    # one large 20-bp bucket (bit-sliced 64-bit-key path) with planted clusters
    rng = np.random.default_rng(9)
    base = rng.integers(0, 4, (3000, 20), dtype=np.uint8)
    reps = rng.integers(1, 12, 3000)
    reads = np.repeat(base, reps, axis=0)
    mut = rng.random(reads.shape) < 0.02
    reads = np.where(mut, (reads + rng.integers(1, 4, reads.shape)) & 3, reads).astype(np.uint8)
    st2 = synth.stage(np.zeros(len(reads), np.int64), synth.bases_to_keys(reads))
    k2, f2, o2 = st2["keys"], st2["freq"], st2["bucket_off"]
    assert len(k2) > 4096
    kept2, root2, _ = ctx.dedup_batch(k2, None, f2, o2, 20, k=2)
    ok, oroot, _ = orc.dedup_batch(k2, None, f2, o2, 20, 2)
    assert (kept2 == ok).all() and (root2 == oroot).all()


def _oracle_on_every_nth_bucket(keys, freq, off, kept, root, umi_len, k, step):
    for b in range(0, len(off) - 1, step):
        s, e = int(off[b]), int(off[b + 1])
        ok, oroot, _ = orc.dedup_batch(keys[s:e], None, freq[s:e], [0, e - s], umi_len, k)
        assert (kept[s:e] == ok).all() and (root[s:e] - s == oroot).all(), b


@pytest.mark.parametrize("name,n_reads,umi_len,k,seed", [
    ("config 4: one GPU's share of 100M reads in 10^6 positions", 12_500_000, 12, 1, 4),
    ("config 5: one GPU's share of 50M reads, 20-bp, k = 2", 6_250_000, 20, 2, 5)])
def test_per_gpu_share_of_the_8_gpu_configs(ctx, name, n_reads, umi_len, k, seed):
    """BASELINE configs 4 and 5 are 8-GPU jobs; what one GPU gets of them (bucket sharding: an
    eighth of the positions) at full size: structure (P1), the oracle on every ~100th bucket (P5),
    and the same call through a two-device context (both workers on this GPU) bit for bit."""
    import umi_collapse_rs_amd as umi
    from umi_collapse_rs_amd import synth
    st = synth.config3(seed=seed, n_reads=n_reads, n_positions=n_reads // 100, umi_len=umi_len)
    keys, freq, off = st["keys"], st["freq"], st["bucket_off"]
    assert len(off) - 1 == n_reads // 100
    kept, root, stats = ctx.dedup_batch(keys, None, freq, off, umi_len, k=k, percentage=0.5)
    assert stats["n_kept"] == int(kept.sum()) and stats["n_buckets"] == n_reads // 100
    check_structure(kept, root, off)
    _oracle_on_every_nth_bucket(keys, freq, off, kept, root, umi_len, k, 101)
    multi = umi.Context([0, 0])
    try:
        mkept, mroot, mst = multi.dedup_batch(keys, None, freq, off, umi_len, k=k, percentage=0.5)
    finally:
        multi.close()
    assert (mkept == kept).all() and (mroot == root).all()
    assert mst["n_kept"] == stats["n_kept"] and mst["n_pairs"] == stats["n_pairs"]


def test_config2m_molecule_model_deep_position(ctx):
    """One deep position from the molecule model (clusters: a true UMI of high freq with its freq-1
    error copies around it): fixed point (P2) on a sample, cross-kernel equality with the key-sorted
    scan + walk and the unsorted all-pairs mask kernel, oracle on a 40k-entry cut of the same data."""
    import umi_collapse_rs_amd as umi
    from umi_collapse_rs_amd import synth
    st = synth.config2m(seed=22, n_reads=1_000_000, umi_len=12)
    keys, freq, off = st["keys"], st["freq"], st["bucket_off"]
    assert 150_000 < len(keys) < 400_000 and freq[0] > 50
    kept, root, stats = ctx.dedup_batch(keys, None, freq, off, 12, k=1, percentage=0.5)
    check_structure(kept, root, off)
    rng = np.random.default_rng(5)
    sample = np.concatenate([rng.choice(len(keys), 150, replace=False), np.nonzero(kept == 0)[0][:50],
                             np.arange(20)])
    check_fixed_point(keys, freq, root, off, 1, 0.5, sample, rng)
    for opts in ({"seg_index": 0}, {"seg_index": 0, "bs_sorted": 0}):
        if not usable(opts):
            continue
        c = umi.Context(0)
        try:
            for name, v in opts.items():
                c.set_option(name, v)
            k2, r2, st2 = c.dedup_batch(keys, None, freq, off, 12, k=1, percentage=0.5)
        finally:
            c.close()
        assert (k2 == kept).all() and (r2 == root).all() and st2["n_edges"] == stats["n_edges"]
    # the first 40,000 entries in rank order (the high-freq true UMIs and a share of the rest)
    m = 40_000
    ok, oroot, _ = orc.dedup_batch(keys[:m], None, freq[:m], [0, m], 12, 1)
    kk, rr, _ = ctx.dedup_batch(keys[:m], None, freq[:m], [0, m], 12, k=1)
    assert (kk == ok).all() and (rr == oroot).all()


def test_large_and_small_calls_alternate_on_one_context():
    """Regression for the staging buffers of the plan (round 1 saw a host segfault in a
    development build when a small single-bucket call followed a large all-fused one on the same
    context: the first call that reached the pinned task staging after calls that never had).
    Calls of very different shapes alternate on ONE context, each checked against the oracle:
    every grow-only device and pinned buffer is reused at a smaller size, regrown, reused."""
    import umi_collapse_rs_amd as umi
    from umi_collapse_rs_amd import synth
    rng = np.random.default_rng(77)
    big20 = synth.config3(seed=5, n_reads=1_270_000, n_positions=20_000, umi_len=20)  # all buckets fused
    raw = rng.integers(0, 4, (11_000, 20), dtype=np.uint8)
    raw[:, 8:] = raw[:1, 8:]  # 8 free bases: one dense 20-bp bucket of ~9,000 entries
    one20 = synth.stage(np.zeros(len(raw), np.int64), synth.bases_to_keys(raw))
    deep12 = synth.config2(seed=9, n_reads=60_000, umi_len=9)
    many12 = synth.config3(seed=3, n_reads=300_000, n_positions=3_000, umi_len=12)
    assert 7_000 < len(one20["keys"]) < 11_000
    calls = [(big20, 20, 2, None), (one20, 20, 2, "full"), (deep12, 9, 1, "full"), (big20, 20, 2, None),
             (many12, 12, 1, "sample"), (one20, 20, 2, "full"), (deep12, 9, 1, "full"), (many12, 12, 1, "sample")]
    for opts in ({}, {"seg_index": 0}):
        c = umi.Context(0)
        try:
            for name, v in opts.items():
                c.set_option(name, v)
            for st, L, k, check in calls:
                keys, freq, off = st["keys"], st["freq"], st["bucket_off"]
                kept, root, stats = c.dedup_batch(keys, None, freq, off, L, k=k)
                check_structure(kept, root, off)
                if check == "full":
                    ok, oroot, _ = orc.dedup_batch(keys, None, freq, off, L, k)
                    assert (kept == ok).all() and (root == oroot).all()
                elif check == "sample":
                    _oracle_on_every_nth_bucket(keys, freq, off, kept, root, L, k, 37)
        finally:
            c.close()
