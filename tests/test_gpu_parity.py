"""Parity of the HIP path (through the C ABI) with the CPU oracle.  Needs an MI355X."""
import numpy as np
import pytest

import oracle as orc
from helpers import canonical, is_dev_build, legacy_mark, random_bucket

legacy = legacy_mark()

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["default", "nofuse", pytest.param("prune", marks=legacy_mark()), "fused_walk",
                                        pytest.param("label_prop", marks=legacy_mark()), "tiles",
                                        pytest.param("hook_rounds", marks=legacy_mark()), "seg_small"])
def ctx(request):
    """default: fused one-wave kernel for buckets <= 128, popcount chunks to the segment index's
    lower bound (512), the n-gram partition (segment index) above, union-find collapse, one host
    synchronisation.  nofuse: buckets <= 1024 all go through the chunk kernel + edge list.
    tiles: no segment index -- popcount chunks to 1024, bit-sliced tiles above (the kernels of the
    earlier versions).  hook_rounds: the symmetric components by hook/jump rounds instead of
    union-find.  seg_small: the segment index from 130 entries up, no fused kernel.
    prune: every bucket > 128 through key-sorted bit-sliced tiles with range pruning.
    fused_walk: the fused kernel's column-walking body instead of its bit-sliced one.
    label_prop: no fused kernel, and plain label propagation instead of the two-phase collapse."""
    import umi_collapse_rs_amd as umi
    c = umi.Context(0)
    if request.param == "nofuse":
        c.set_option("fused_max", 0)
    if request.param == "label_prop":
        c.set_option("fused_max", 0)
        c.set_option("two_phase", 0)
    if request.param == "fused_walk":
        c.set_option("fused_sliced", 0)
    if request.param == "prune":
        c.set_option("prune", 1)
        c.set_option("small_max", 128)
    if request.param == "tiles":
        c.set_option("seg_index", 0)
    if request.param == "hook_rounds":
        c.set_option("two_phase", 1)
    if request.param == "seg_small":
        c.set_option("seg_min", 130)
        c.set_option("fused_max", 0)
    yield c
    c.close()


def make_batch(rng, n_buckets, L, n_mol_max, err=0.05, n_frac=0.0, mean_copies=3.0,
               exact=False):
    keys, nm, fr, off = [], [], [], [0]
    for _ in range(n_buckets):
        n_mol = n_mol_max if exact else int(rng.integers(0, n_mol_max + 1))
        umis, freq = random_bucket(rng, n_mol, L, err=err, n_frac=n_frac,
                                   mean_copies=mean_copies)
        umis, freq, _ = canonical(umis, freq)
        k, m = orc.encode_keys(umis)
        keys.append(k); nm.append(m); fr.extend(freq)
        off.append(off[-1] + len(umis))
    keys = np.concatenate(keys) if keys else np.zeros(0, np.uint64)
    nm = np.concatenate(nm) if nm else np.zeros(0, np.uint64)
    return keys, nm, np.array(fr, np.int32), np.array(off, np.uint64)


def check_against_oracle(ctx, keys, nm, fr, off, L, k, p=0.5, algo=0, amf=0):
    kept, root, st = ctx.dedup_batch(keys, nm if nm.any() else None, fr, off, L, k, p, algo, amf)
    okept, oroot, _ = orc.dedup_batch(keys, nm, fr, off, L, k, p, algo, amf)
    assert (kept == okept).all(), "kept mask differs at %s" % np.nonzero(kept != okept)[0][:10]
    assert (root == oroot).all(), "root differs at %s" % np.nonzero(root != oroot)[0][:10]
    assert st["n_kept"] == int(okept.sum())
    assert st["n_umis"] == len(keys)
    n_b = np.diff(off.astype(np.int64))
    assert st["n_pairs"] == int((n_b * (n_b - 1) // 2).sum())
    assert st["max_bucket"] == (int(n_b.max()) if len(n_b) else 0)
    return st


def test_kat_g8_g9_batched(ctx, kat):
    g = kat["G8_bucket"]
    umis, freq, order = canonical(g["umis"], g["freq"])
    assert order == g["rank"]
    keys, nm = orc.encode_keys(umis)
    kept, root, _ = ctx.dedup_batch(keys, None, freq, [0, len(umis)], 12, g["k"], g["p"])
    assert [order[i] for i in np.nonzero(kept)[0]] == g["dir"]
    kept, _, _ = ctx.dedup_batch(keys, None, freq, [0, len(umis)], 12, g["k"], g["p"], algo=1)
    assert [order[i] for i in np.nonzero(kept)[0]] == g["adj"]
    g9 = kat["G9_tie"]
    keys, nm = orc.encode_keys(g9["umis"])
    for k, exp in ((1, g9["dir_k1"]), (0, g9["dir_k0"])):
        kept, _, _ = ctx.dedup_batch(keys, None, g9["freq"], [0, 2], 12, k, g9["p"])
        assert np.nonzero(kept)[0].tolist() == exp


@pytest.mark.parametrize("L,k,p,n_frac", [
    (12, 1, 0.5, 0.0), (12, 0, 0.5, 0.0), (12, 2, 0.5, 0.0), (12, 3, 0.5, 0.01),
    (12, 1, 0.3, 0.0), (12, 1, 1.0, 0.0), (6, 1, 0.5, 0.05), (16, 2, 0.5, 0.02),
    (17, 1, 0.5, 0.0), (20, 2, 0.5, 0.0), (20, 2, 0.75, 0.03), (21, 3, 0.5, 0.01),
    (1, 1, 0.5, 0.0), (3, 1, 0.5, 0.2),
])
def test_random_buckets_directional(ctx, L, k, p, n_frac):
    rng = np.random.default_rng(100 * L + 7 * k + int(10 * p))
    keys, nm, fr, off = make_batch(rng, 60, L, 40, err=0.06, n_frac=n_frac)
    check_against_oracle(ctx, keys, nm, fr, off, L, k, p)


@pytest.mark.parametrize("p", [-0.5, -0.0, 2.0, 1e9, -1e9, float("inf"), float("-inf"), float("nan")])
def test_threshold_edge_cases(ctx, p):
    """directional.rs:100-102 on odd inputs: a negative percentage makes the thresholds rise
    with rank, freq = i32::MAX wraps freq + 1 and the cast saturates / maps NaN to 0.  The
    fused kernel's prefix search must notice that the thresholds are out of order."""
    rng = np.random.default_rng(4242)
    keys, nm, fr, off = make_batch(rng, 40, 8, 45, err=0.08, n_frac=0.01)
    keys2, nm2, fr2, off2 = make_batch(rng, 3, 9, 500, err=0.05, exact=True)
    for kk, mm, ff, oo, L in ((keys, nm, fr, off, 8), (keys2, nm2, fr2, off2, 9)):
        for k in (1, 2):
            check_against_oracle(ctx, kk, mm, ff, oo, L, k, p)
        big = ff.copy() # the top entries of every bucket at and next to i32::MAX
        starts = oo[:-1][np.diff(oo.astype(np.int64)) > 3].astype(np.int64)
        big[starts] = 0x7FFFFFFF
        big[starts + 1] = 0x7FFFFFFF
        big[starts + 2] = 0x7FFFFFFE
        check_against_oracle(ctx, kk, mm, big, oo, L, 1, p)


@pytest.mark.parametrize("amf", [0, 1, 2, 1 << 30])
def test_random_buckets_adjacency(ctx, amf):
    rng = np.random.default_rng(900 + amf % 13)
    keys, nm, fr, off = make_batch(rng, 50, 10, 40, err=0.08)
    check_against_oracle(ctx, keys, nm, fr, off, 10, 1, algo=1, amf=amf)
    check_against_oracle(ctx, keys, nm, fr, off, 10, 2, algo=1, amf=amf)


def test_bucket_sizes_around_kernel_boundaries(ctx):
    """Buckets of exactly 1, 2, 63, 64, 65, 127, 128, 129 entries (fused 1- and 2-row
    variants, then the chunk kernel), with N and with p > 0.5."""
    rng = np.random.default_rng(77)
    for L, k, p, n_frac in ((9, 1, 0.5, 0.0), (12, 2, 1.0, 0.02), (20, 1, 0.5, 0.01)):
        keys, nm, fr, off = [], [], [], [0]
        for n in (1, 2, 63, 64, 65, 127, 128, 129, 64, 128):
            umis, freq = [], []
            while len(umis) < n:
                u, f = random_bucket(rng, 40, L, err=0.15, n_frac=n_frac)
                for a, b in zip(u, f):
                    if a not in umis and len(umis) < n:
                        umis.append(a); freq.append(b)
            umis, freq, _ = canonical(umis, freq)
            kk, mm = orc.encode_keys(umis)
            keys.append(kk); nm.append(mm); fr.extend(freq); off.append(off[-1] + n)
        check_against_oracle(ctx, np.concatenate(keys), np.concatenate(nm), np.array(fr, np.int32),
                             np.array(off, np.uint64), L, k, p)
        check_against_oracle(ctx, np.concatenate(keys), np.concatenate(nm), np.array(fr, np.int32),
                             np.array(off, np.uint64), L, k, algo=1, amf=2)


def test_empty_and_degenerate_inputs(ctx):
    kept, root, st = ctx.dedup_batch(np.zeros(0, np.uint64), None, np.zeros(0, np.int32), [0], 12)
    assert len(kept) == 0 and st["n_umis"] == 0
    kept, _, _ = ctx.dedup_batch(np.zeros(0, np.uint64), None, np.zeros(0, np.int32),
                                 [0, 0, 0], 12)
    assert len(kept) == 0
    # singleton buckets and empty buckets mixed
    keys, nm = orc.encode_keys(["ACGTACGTACGT", "ACGTACGTACGA", "TTTTTTTTTTTT"])
    kept, root, st = ctx.dedup_batch(keys, None, [1, 1, 4], [0, 0, 1, 2, 2, 3], 12)
    assert kept.tolist() == [1, 1, 1] and root.tolist() == [0, 1, 2]
    assert st["n_pairs"] == 0 and st["n_kept"] == 3


def test_medium_bucket_crosses_tile_boundaries(ctx):
    # one bucket larger than a 64-row chunk and a 1024-column LDS tile, dense in dist-1 pairs
    rng = np.random.default_rng(42)
    keys, nm, fr, off = make_batch(rng, 1, 7, 1500, err=0.1, mean_copies=2.0, exact=True)
    assert 1024 < len(keys) < 20000
    check_against_oracle(ctx, keys, nm, fr, off, 7, 1)
    # several such buckets next to small ones
    keys, nm, fr, off = make_batch(rng, 6, 8, 900, err=0.05, exact=True)
    check_against_oracle(ctx, keys, nm, fr, off, 8, 2)


@pytest.mark.parametrize("bitslice,unit", [pytest.param(1, 2, marks=legacy), pytest.param(1, 1, marks=legacy),
                                           pytest.param(1, 3, marks=legacy), (0, 2)])
def test_tile_kernels_on_all_sizes(bitslice, unit):
    """small_max=0, fused_max=0 send every bucket through the tile kernels: the bit-sliced
    one (k<=3; counting units of 2 bases or single bases) or the popcount one."""
    import umi_collapse_rs_amd as umi
    c = umi.Context(0)
    c.set_option("small_max", 0)
    c.set_option("fused_max", 0)
    c.set_option("seg_index", 0)
    if is_dev_build():  # (the shipped library has the popcount tiles only)
        c.set_option("bitslice", bitslice)
        c.set_option("bs_unit", unit)
    try:
        rng = np.random.default_rng(43)
        keys, nm, fr, off = make_batch(rng, 40, 12, 60, err=0.05, n_frac=0.01)
        for k in (0, 1, 2, 3, 4):
            check_against_oracle(c, keys, nm, fr, off, 12, k)
        keys, nm, fr, off = make_batch(rng, 2, 7, 2500, err=0.08, mean_copies=2.0, exact=True)
        assert np.diff(off.astype(np.int64)).max() > 2048
        check_against_oracle(c, keys, nm, fr, off, 7, 1)
        check_against_oracle(c, keys, nm, fr, off, 7, 2, p=1.0)
        keys, nm, fr, off = make_batch(rng, 2, 20, 1500, err=0.03, n_frac=0.01, exact=True)
        check_against_oracle(c, keys, nm, fr, off, 20, 2)
        keys, nm, fr, off = make_batch(rng, 3, 21, 700, err=0.03, exact=True)
        check_against_oracle(c, keys, nm, fr, off, 21, 3)
        keys, nm, fr, off = make_batch(rng, 3, 16, 700, err=0.05, n_frac=0.02, exact=True)
        check_against_oracle(c, keys, nm, fr, off, 16, 1)
        keys, nm, fr, off = make_batch(rng, 5, 5, 300, err=0.1, exact=True)
        check_against_oracle(c, keys, nm, fr, off, 5, 1)
        check_against_oracle(c, keys, nm, fr, off, 5, 1, algo=1, amf=3)
    finally:
        c.close()


def test_bitsliced_rows_beyond_one_tile_and_column_chunks():
    """One bucket > 4096 rows (several 64-thread row tiles, diagonal and off-diagonal
    tasks, more than one 4096-column chunk) against the oracle."""
    import umi_collapse_rs_amd as umi
    rng = np.random.default_rng(47)
    L = 8
    raw = rng.integers(0, 4, (12000, L))
    umis = sorted({"".join("ACGT"[c] for c in r) for r in raw})
    rng.shuffle(umis)
    freq = np.minimum(rng.geometric(0.5, len(umis)), 30).tolist()
    umis, freq, _ = canonical(umis, freq)
    assert len(umis) > 9000
    keys, nm = orc.encode_keys(umis)
    off = np.array([0, len(umis)], np.uint64)
    c = umi.Context(0)
    try:
        st_seg = check_against_oracle(c, keys, nm, np.array(freq, np.int32), off, L, 1)
        c.set_option("seg_index", 0)  # the bit-sliced tiles
        st = check_against_oracle(c, keys, nm, np.array(freq, np.int32), off, L, 1)
        assert st["n_pair_launches"] >= 1 and st["n_edges"] == st_seg["n_edges"]
        assert st_seg["n_pairs_evaluated"] < st["n_pairs"] < st["n_pairs_evaluated"]
        if not is_dev_build():
            return
        c.set_option("prune", 1)
        st2 = check_against_oracle(c, keys, nm, np.array(freq, np.int32), off, L, 1)
        assert st2["n_pairs_evaluated"] <= st["n_pairs_evaluated"]  # tile tasks may be skipped
        assert st2["n_edges"] == st["n_edges"]
    finally:
        c.close()


def test_large_single_bucket_both_kernels(ctx):
    """~20k unique 9-mers at one position (7.6% of the 4^9 space: chains and a giant
    component).  Oracle needs a few seconds."""
    rng = np.random.default_rng(44)
    L = 9
    raw = rng.integers(0, 4, (22000, L))
    uniq = {"".join("ACGT"[c] for c in r) for r in raw}
    umis = sorted(uniq)
    rng.shuffle(umis)
    freq = np.minimum(rng.geometric(0.6, len(umis)), 50).tolist()
    umis, freq, _ = canonical(umis, freq)
    keys, nm = orc.encode_keys(umis)
    off = np.array([0, len(umis)], np.uint64)
    st = check_against_oracle(ctx, keys, nm, np.array(freq, np.int32), off, L, 1)
    assert st["n_edges"] > 0 and st["n_rounds"] >= 2
    import umi_collapse_rs_amd as umi
    c = umi.Context(0)
    c.set_option("small_max", 1 << 20)  # same bucket through the wave-per-chunk kernel
    try:
        check_against_oracle(c, keys, nm, np.array(freq, np.int32), off, L, 1)
    finally:
        c.close()


def _wide_bucket(rng, n_raw, L, n_frac=0.0):
    raw = rng.integers(0, 4, (n_raw, L))
    if n_frac:
        raw = np.where(rng.random(raw.shape) < n_frac, 4, raw)
    umis = sorted({"".join("ACGTN"[c] for c in r) for r in raw})
    rng.shuffle(umis)
    freq = np.minimum(rng.geometric(0.5, len(umis)), 40).tolist()
    umis, freq, _ = canonical(umis, freq)
    keys, nm = orc.encode_keys(umis)
    return keys, nm, np.array(freq, np.int32), np.array([0, len(umis)], np.uint64)


@pytest.mark.parametrize("L,k,n_raw,n_frac,algo,amf", [
    (10, 1, 42000, 0.0, 0, 0),    # 32-bit keys, table kernel (2 live units + cached prefix)
    (9, 2, 40000, 0.002, 0, 0),   # the same with N bases and k = 2
    (8, 0, 60000, 0.0, 0, 0),     # LP = 8: two prefix units only
    (10, 1, 40000, 0.0, 1, 2),    # adjacency with a real max_freq
    (12, 1, 36000, 0.0, 0, 0),    # too few entries per 8-base prefix: mask kernel, 3 cached units
    (18, 2, 34000, 0.001, 0, 0),  # 64-bit keys: mask kernel with cached prefix, one group per lane
    (14, 3, 34000, 0.0, 0, 0),    # LP = 16
])
def test_wide_sorted_buckets(L, k, n_raw, n_frac, algo, amf):
    """Buckets of >= 32768 entries: sorted by filter key on the device; the table kernel or the
    mask kernel with cached prefix state, checked against the oracle and against the same
    context with sorting / tables switched off (identical edges, not only identical output)."""
    import umi_collapse_rs_amd as umi
    rng = np.random.default_rng(1000 * L + k)
    keys, nm, fr, off = _wide_bucket(rng, n_raw, L, n_frac)
    assert len(keys) >= 32768
    c = umi.Context(0)
    try:
        st_seg = check_against_oracle(c, keys, nm, fr, off, L, k, algo=algo, amf=amf)
        c.set_option("seg_index", 0)
        st = check_against_oracle(c, keys, nm, fr, off, L, k, algo=algo, amf=amf)
        assert st_seg["n_edges"] == st["n_edges"]
        if not is_dev_build():
            return
        c.set_option("bs_tables", 0)
        st1 = check_against_oracle(c, keys, nm, fr, off, L, k, algo=algo, amf=amf)
        c.set_option("bs_sorted", 0)
        st0 = check_against_oracle(c, keys, nm, fr, off, L, k, algo=algo, amf=amf)
        assert st["n_edges"] == st1["n_edges"] == st0["n_edges"]
    finally:
        c.close()


@pytest.mark.parametrize("n_extra", [0, 1])
def test_bucket_at_the_sort_merge_boundary(n_extra):
    """A saturated 8-bp position holds all 4^8 = 65536 UMIs: exactly the size up to which the
    library sort takes its merge path (size <= merge_sort_limit); one entry more goes through the
    onesweep passes with the restricted bit range.  Both against the oracle, and with the scan +
    walk forced (bs_tab_min_run = 0) so that the key-sorted path is the one that runs."""
    import umi_collapse_rs_amd as umi
    rng = np.random.default_rng(4 ** 8 + n_extra)
    L = 8 + n_extra
    ids = np.arange(4 ** 8, dtype=np.int64)
    if n_extra:  # 9-mers: every 8-mer with a fixed ninth base, and one more
        ids = np.concatenate([ids, [4 ** 8 + 12345]])
    umis = ["".join("ACGT"[(int(i) >> (2 * b)) & 3] for b in range(L)) for i in ids]
    rng.shuffle(umis)
    freq = np.minimum(rng.geometric(0.5, len(umis)), 30).tolist()
    umis, freq, _ = canonical(umis, freq)
    keys, nm = orc.encode_keys(umis)
    fr, off = np.array(freq, np.int32), np.array([0, len(umis)], np.uint64)
    assert len(keys) == 65536 + n_extra
    c = umi.Context(0)
    try:
        st_seg = check_against_oracle(c, keys, nm, fr, off, L, 1)
        c.set_option("seg_index", 0)
        if not is_dev_build():
            st = check_against_oracle(c, keys, nm, fr, off, L, 1)
            assert st["n_edges"] == st_seg["n_edges"]
            return
        c.set_option("bs_tab_min_run", 0)
        st = check_against_oracle(c, keys, nm, fr, off, L, 1)
        c.set_option("bs_sorted", 0)
        st0 = check_against_oracle(c, keys, nm, fr, off, L, 1)
        assert st["n_edges"] == st0["n_edges"] == st_seg["n_edges"]
    finally:
        c.close()


@legacy
@pytest.mark.parametrize("L,k,algo", [(12, 1, 0), (13, 2, 0), (16, 3, 0), (11, 0, 0), (12, 1, 1)])
def test_table_kernel_forced_on_several_buckets(L, k, algo):
    """bs_tab_min_run = 0 sends every bucket of >= 32768 entries with 32-bit keys through the scan +
    table kernels, whatever the run length: three such buckets of unlike sizes (ragged last row
    tile, last column tile of a few columns) and small ones between them in one call, against the
    oracle and against the mask kernel."""
    import umi_collapse_rs_amd as umi
    rng = np.random.default_rng(500 + 10 * L + k)
    parts = []
    for n_raw in (33500, 7, 41000, 300, 36001, 1):
        parts.append(_wide_bucket(rng, n_raw, L, 0.001 if L == 13 else 0.0))
    keys = np.concatenate([p[0] for p in parts])
    nm = np.concatenate([p[1] for p in parts])
    fr = np.concatenate([p[2] for p in parts])
    off = np.concatenate([[0], np.cumsum([len(p[0]) for p in parts])]).astype(np.uint64)
    c = umi.Context(0)
    try:
        c.set_option("seg_index", 0)
        c.set_option("bs_tab_min_run", 0)
        c.set_option("edge_capacity", 64)  # both lists run over on the first attempt: the pair
        c.set_option("ovf_capacity", 16)   # stage is redone, scan and item counters included
        st = check_against_oracle(c, keys, nm, fr, off, L, k, algo=algo, amf=1 if algo else 0)
        assert st["n_pairs_evaluated"] > 0
        nmask = nm if nm.any() else None
        kept, root, _ = c.dedup_batch(keys, nmask, fr, off, L, k, 0.5, algo, 1 if algo else 0)
        for name in ("bs_transposed", "bs_tables"): # the table walk of the same items; the mask kernel
            c.set_option(name, 0)
            kept1, root1, st1 = c.dedup_batch(keys, nmask, fr, off, L, k, 0.5, algo, 1 if algo else 0)
            assert (kept1 == kept).all() and (root1 == root).all(), name
            assert st["n_edges"] == st1["n_edges"], name
    finally:
        c.close()


@pytest.mark.parametrize("L,k", [(12, 2), (11, 1)])
def test_wide_sorted_bucket_dense_in_neighbours(L, k):
    """A wide bucket made of 1,000 centres with all their single-substitution variants: in
    key-sorted order the filter hits crowd into the tiles next to the diagonal, and with k = 2
    (every pair of a cluster within reach) the hit queue and the edge stage of a block run over:
    the hits beyond them are checked and written directly."""
    import umi_collapse_rs_amd as umi
    rng = np.random.default_rng(77 + L)
    seen = {}
    for c in rng.integers(0, 4, (1100, L)):
        members = [c]
        for pos in range(L):
            for b in range(4):
                if b != c[pos]:
                    u = c.copy()
                    u[pos] = b
                    members.append(u)
        for u in members:
            s_ = "".join("ACGT"[x] for x in u)
            seen[s_] = seen.get(s_, 0) + int(rng.geometric(0.5))
    umis = list(seen.keys())
    rng.shuffle(umis)
    freq = [seen[u] for u in umis]
    umis, freq, _ = canonical(umis, freq)
    assert len(umis) >= 32768
    keys, nm = orc.encode_keys(umis)
    fr, off = np.array(freq, np.int32), np.array([0, len(umis)], np.uint64)
    c = umi.Context(0)
    try:
        st_seg = check_against_oracle(c, keys, nm, fr, off, L, k)  # segment index: dense sub-buckets
        assert st_seg["n_candidates"] >= st_seg["n_edges"] > (8 if k == 2 else 1) * len(umis)
        c.set_option("seg_index", 0)
        st = check_against_oracle(c, keys, nm, fr, off, L, k)
        assert st["n_candidates"] > (8 if k == 2 else 1) * len(umis)
        assert st["n_edges"] == st_seg["n_edges"]
        if is_dev_build():
            c.set_option("bs_sorted", 0)
            st0 = check_against_oracle(c, keys, nm, fr, off, L, k)
            assert st0["n_edges"] == st["n_edges"]
    finally:
        c.close()
    # the global overflow list itself too short at first: everything again with a longer one
    for seg_index in (0, 1):
        c = umi.Context(0)
        c.set_option("seg_index", seg_index)
        if is_dev_build():
            c.set_option("ovf_capacity", 16)
        c.set_option("edge_capacity", 64)
        try:
            st1 = check_against_oracle(c, keys, nm, fr, off, L, k)
            assert st1["n_edges"] == st["n_edges"]
        finally:
            c.close()


def test_edge_list_overflow_is_transparent():
    import umi_collapse_rs_amd as umi
    c = umi.Context(0)
    c.set_option("edge_capacity", 8)
    try:
        rng = np.random.default_rng(45)
        keys, nm, fr, off = make_batch(rng, 30, 8, 60, err=0.1)
        st = check_against_oracle(c, keys, nm, fr, off, 8, 2)
        assert st["n_edges"] > 8
    finally:
        c.close()


@pytest.mark.parametrize("spin", [0, 1])
def test_end_of_call_seen_by_the_runtime_or_by_watching_pinned_memory(spin):
    """option spin_wait: the stream's last kernel writes a sequence word behind the control block in pinned
    host memory (1, default) / hipStreamSynchronize (0) -- same results, also when the device-side outputs are
    read straight after the call returns, and call after call on one context (the sequence goes on)."""
    import torch
    import umi_collapse_rs_amd as umi
    rng = np.random.default_rng(460 + spin)
    c = umi.Context(0)
    try:
        c.set_option("spin_wait", spin)
        dev = torch.device("cuda:0")
        for rep in range(4):
            keys, nm, fr, off = make_batch(rng, 200 + 50 * rep, 12, 50, err=0.05)
            kb, nb_, fb, ob = make_batch(rng, 1, 12, 900, err=0.05, exact=True)  # one position for the segment index
            keys = np.concatenate([keys, kb]); nm = np.concatenate([nm, nb_]); fr = np.concatenate([fr, fb])
            off = np.concatenate([off, off[-1] + ob[1:]]).astype(np.uint64)
            t_keys = torch.from_numpy(keys.view(np.int64)).to(dev)
            t_fr = torch.from_numpy(fr).to(dev)
            t_kept = torch.zeros(len(keys), dtype=torch.uint8, device=dev)
            t_root = torch.zeros(len(keys), dtype=torch.int32, device=dev)
            st = c.dedup_batch_device(t_keys.data_ptr(), 0, t_fr.data_ptr(), off, 12, t_kept.data_ptr(), t_root.data_ptr(),
                                      k=1, stream=torch.cuda.current_stream().cuda_stream)
            got_kept, got_root = t_kept.cpu().numpy(), t_root.cpu().numpy().view(np.uint32)
            okept, oroot, _ = orc.dedup_batch(keys, None, fr, off, 12, 1)
            assert (got_kept == okept).all() and (got_root == oroot).all() and st["n_kept"] == int(okept.sum())
    finally:
        c.close()


def test_device_call_in_two_halves():
    """umi_dedup_batch_device_begin / umi_dedup_batch_end: a batch of small positions is left on the stream
    (work enqueued behind it on the same stream -- here the packing of the mask -- reads the final outputs),
    a batch with a deep position runs to its end inside begin; both give the plain call's result.  A second
    begin replaces the first; another call on the context lets a pending one end, its result keeps waiting; a
    contract violation surfaces at end; end with nothing begun is an error."""
    import torch
    import umi_collapse_rs_amd as umi
    rng = np.random.default_rng(4700)
    dev = torch.device("cuda:0")
    c = umi.Context(0)
    try:
        with pytest.raises(umi.UmiHipError):
            c.dedup_batch_end()
        small = make_batch(rng, 3000, 12, 25, err=0.05)
        assert np.diff(small[3].astype(np.int64)).max() <= 128  # every position the fused kernel's: the end is deferred
        kb, nb_, fb, ob = make_batch(rng, 1, 12, 900, err=0.05, exact=True)
        deep = (np.concatenate([small[0], kb]), np.concatenate([small[1], nb_]), np.concatenate([small[2], fb]),
                np.concatenate([small[3], small[3][-1] + ob[1:]]).astype(np.uint64))
        stream = torch.cuda.current_stream().cuda_stream
        for keys, nm, fr, off in (small, deep, small):
            t_keys = torch.from_numpy(keys.view(np.int64)).to(dev)
            t_fr = torch.from_numpy(fr).to(dev)
            t_kept = torch.zeros(len(keys), dtype=torch.uint8, device=dev)
            t_root = torch.zeros(len(keys), dtype=torch.int32, device=dev)
            t_bits = torch.zeros((len(keys) + 7) // 8, dtype=torch.uint8, device=dev)
            t_off = torch.from_numpy(off.view(np.int64)).to(dev)
            c.dedup_batch_device_begin(t_keys.data_ptr(), 0, t_fr.data_ptr(), off, 12, t_kept.data_ptr(), t_root.data_ptr(),
                                       k=1, stream=stream, d_bucket_off=t_off.data_ptr())
            c.pack_mask_device(t_kept.data_ptr(), len(keys), t_bits.data_ptr(), stream=stream)  # behind the call, same stream
            st = c.dedup_batch_end()
            torch.cuda.synchronize()
            okept, oroot, _ = orc.dedup_batch(keys, None, fr, off, 12, 1)
            assert (t_kept.cpu().numpy() == okept).all() and (t_root.cpu().numpy().view(np.uint32) == oroot).all()
            assert st["n_kept"] == int(okept.sum()) and st["n_umis"] == len(keys)
            assert (np.unpackbits(t_bits.cpu().numpy(), bitorder="little")[:len(keys)] == okept).all()
        # two begins in a row: the first ends by itself; then a plain call; then the pending end
        keys, nm, fr, off = small
        t_keys = torch.from_numpy(keys.view(np.int64)).to(dev)
        t_fr = torch.from_numpy(fr).to(dev)
        t_kept = torch.zeros(len(keys), dtype=torch.uint8, device=dev)
        for _ in range(2):
            c.dedup_batch_device_begin(t_keys.data_ptr(), 0, t_fr.data_ptr(), off, 12, t_kept.data_ptr(), 0, k=1, stream=stream)
        st_plain = c.dedup_batch_device(t_keys.data_ptr(), 0, t_fr.data_ptr(), off, 12, t_kept.data_ptr(), 0, k=1, stream=stream)
        okept, _, _ = orc.dedup_batch(keys, None, fr, off, 12, 1)
        assert st_plain["n_kept"] == int(okept.sum()) and (t_kept.cpu().numpy() == okept).all()
        assert c.dedup_batch_end()["n_kept"] == int(okept.sum())  # (the plain call let the pending one end; its result waited)
        with pytest.raises(umi.UmiHipError):
            c.dedup_batch_end()  # (handed out: nothing is out now)
        # a contract violation (rank order) is reported by end
        bad = fr.copy()
        first = int(off[0]); n0 = int(off[1] - off[0])
        if n0 >= 2:
            bad[first], bad[first + 1] = 1, 5
            t_bad = torch.from_numpy(bad).to(dev)
            c.dedup_batch_device_begin(t_keys.data_ptr(), 0, t_bad.data_ptr(), off, 12, t_kept.data_ptr(), 0, k=1, stream=stream)
            with pytest.raises(umi.UmiHipError):
                c.dedup_batch_end()
            # ... and by begin where the call runs to its end there
            kd, _, fd, od = deep
            fd = fd.copy()
            fd[int(od[-2])], fd[int(od[-2]) + 1] = 1, 7
            with pytest.raises(umi.UmiHipError):
                c.dedup_batch_device_begin(torch.from_numpy(kd.view(np.int64)).to(dev).data_ptr(), 0,
                                           torch.from_numpy(fd).to(dev).data_ptr(), od, 12,
                                           torch.zeros(len(kd), dtype=torch.uint8, device=dev).data_ptr(), 0, k=1, stream=stream)
    finally:
        c.close()


def test_contract_violations_are_reported(ctx):
    import umi_collapse_rs_amd as umi
    from umi_collapse_rs_amd import _lib
    keys, nm = orc.encode_keys(["AAAA", "AAAT", "CCCC"])
    with pytest.raises(umi.UmiHipError) as e:  # not freq-descending inside the bucket
        ctx.dedup_batch(keys, None, [1, 2, 1], [0, 3], 4)
    assert e.value.code == _lib.UMI_ERR_ORDER
    with pytest.raises(umi.UmiHipError) as e:  # freq 0
        ctx.dedup_batch(keys, None, [1, 1, 0], [0, 3], 4)
    assert e.value.code == _lib.UMI_ERR_ORDER
    # a rise in freq exactly at a bucket boundary is fine
    kept, _, _ = ctx.dedup_batch(keys, None, [1, 2, 1], [0, 1, 3], 4)
    assert kept.tolist() == [1, 1, 1]
    for bad in (dict(umi_len=0), dict(umi_len=22), dict(k=-1), dict(algo=7)):
        args = dict(umi_len=4, k=1, algo=0)
        args.update(bad)
        with pytest.raises(umi.UmiHipError) as e:
            ctx.dedup_batch(keys, None, [1, 1, 1], [0, 3], args["umi_len"], args["k"], 0.5,
                            args["algo"])
        assert e.value.code == _lib.UMI_ERR_ARG


def test_device_pointer_entry_point(ctx):
    import torch
    rng = np.random.default_rng(46)
    keys, nm, fr, off = make_batch(rng, 40, 12, 50, err=0.05, n_frac=0.01)
    dev = torch.device("cuda:0")
    t_keys = torch.from_numpy(keys.view(np.int64)).to(dev)
    t_nm = torch.from_numpy(nm.view(np.int64)).to(dev)
    t_fr = torch.from_numpy(fr).to(dev)
    t_kept = torch.zeros(len(keys), dtype=torch.uint8, device=dev)
    t_root = torch.zeros(len(keys), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    st = ctx.dedup_batch_device(t_keys.data_ptr(), t_nm.data_ptr(), t_fr.data_ptr(), off, 12,
                                t_kept.data_ptr(), t_root.data_ptr(), k=1, stream=stream)
    torch.cuda.synchronize()
    okept, oroot, _ = orc.dedup_batch(keys, nm, fr, off, 12, 1)
    assert (t_kept.cpu().numpy() == okept).all()
    assert (t_root.cpu().numpy().view(np.uint32) == oroot).all()
    assert st["n_kept"] == int(okept.sum())
    # the bucket table resident on the device as well (umi_dedup_batch_device_table): same result,
    # small buckets (fused kernel) and one beyond its reach (prep / rise check read the table too)
    import umi_collapse_rs_amd as umi
    keys2, nm2, fr2, off2 = make_batch(rng, 3000, 12, 40, err=0.05)
    kb, nb_, fb, ob = make_batch(rng, 1, 12, 700, err=0.05, exact=True)
    keys2 = np.concatenate([keys2, kb]); nm2 = np.concatenate([nm2, nb_]); fr2 = np.concatenate([fr2, fb])
    off2 = np.concatenate([off2, off2[-1] + ob[1:]]).astype(np.uint64)
    t_keys = torch.from_numpy(keys2.view(np.int64)).to(dev)
    t_fr = torch.from_numpy(fr2).to(dev)
    t_off = torch.from_numpy(off2.view(np.int64)).to(dev)
    t_kept = torch.zeros(len(keys2), dtype=torch.uint8, device=dev)
    t_root = torch.zeros(len(keys2), dtype=torch.int32, device=dev)
    okept, oroot, _ = orc.dedup_batch(keys2, nm2, fr2, off2, 12, 1)
    for table in (t_off.data_ptr(), 0):
        t_kept.zero_()
        st = ctx.dedup_batch_device(t_keys.data_ptr(), 0, t_fr.data_ptr(), off2, 12, t_kept.data_ptr(),
                                    t_root.data_ptr(), k=1, stream=stream, d_bucket_off=table)
        torch.cuda.synchronize()
        assert (t_kept.cpu().numpy() == okept).all() and st["n_kept"] == int(okept.sum())
        assert (t_root.cpu().numpy().view(np.uint32) == oroot).all()
    # a device table that is not the host's: entries that would lead outside the arrays are refused
    bad = torch.from_numpy((off2 + np.uint64(len(keys2))).view(np.int64)).to(dev)
    with pytest.raises(umi.UmiHipError):
        ctx.dedup_batch_device(t_keys.data_ptr(), 0, t_fr.data_ptr(), off2, 12, t_kept.data_ptr(),
                               t_root.data_ptr(), k=1, stream=stream, d_bucket_off=bad.data_ptr())
    torch.cuda.synchronize()


@pytest.mark.parametrize("L,k,n_raw,n_frac", [
    (10, 1, 42000, 0.0),    # one segment of ~40k entries: three 16k-entry blocks of the LDS counting sort
    (9, 2, 40000, 0.002),   # three parts, N bases
    (18, 2, 34000, 0.001),  # 64-bit keys
])
def test_segment_index_counting_sort_and_union_variants(L, k, n_raw, n_frac):
    """The segment index with its counting sort through per-block LDS histograms or per-entry
    atomics (seg_lds), symmetric pairs united where they are found or through the edge list
    (seg_unite): every combination against the oracle, same edge count; a deep position next to
    mid-size ones (several segments in one call, one of them spanning several blocks)."""
    import umi_collapse_rs_amd as umi
    rng = np.random.default_rng(77 * L + k)
    keys, nm, fr, off = _wide_bucket(rng, n_raw, L, n_frac)
    k2, n2, f2, o2 = make_batch(rng, 6, L, 900, err=0.05, n_frac=n_frac, exact=True)  # segments of ~600-2000 entries
    keys = np.concatenate([k2, keys]); nm = np.concatenate([n2, nm]); fr = np.concatenate([f2, fr])
    off = np.concatenate([o2, o2[-1] + off[1:]]).astype(np.uint64)
    c = umi.Context(0)
    try:
        ref = None
        for lds, unite, ckey, sliced in ((1, 1, 1, 1), (1, 1, 1, 0), (1, 1, 0, 1), (1, 0, 1, 1), (0, 1, 1, 0), (0, 0, 0, 0)):
            if True:
                c.set_option("seg_lds", lds)
                c.set_option("seg_unite", unite)
                c.set_option("seg_ckey", ckey)  # compare keys (3 bits per base outside the bin) or filter keys
                c.set_option("seg_sliced", sliced)  # 64 columns at a time from ballots, or a broadcast per column
                st = check_against_oracle(c, keys, nm, fr, off, L, k)
                assert st["n_edges"] > 0
                ref = ref or st
                assert st["n_edges"] == ref["n_edges"] and st["n_pairs_evaluated"] == ref["n_pairs_evaluated"]
    finally:
        c.close()
