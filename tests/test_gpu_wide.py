"""Keys of more than one word (umi_len 22..85, umi_dedup_batch_wide): the per-word arithmetic of
src/utils/bitset.rs:77-91 on the device against the oracle's restatement, the base that straddles
words 0 and 1 (KAT G5) included."""
import numpy as np
import pytest

import oracle as orc
from helpers import canonical

pytestmark = pytest.mark.gpu
ALPHA = "ACGT"


@pytest.fixture(scope="module")
def ctx():
    import umi_collapse_rs_amd as umi
    c = umi.Context(0)
    yield c
    c.close()


def random_bucket(rng, n_mol, L, err, n_frac):
    """Molecules with error copies; N bases sprinkled in (also at base 21, the straddler)."""
    out = {}
    for _ in range(n_mol):
        true = rng.integers(0, 4, L)
        for _ in range(int(rng.geometric(0.4))):
            u = true.copy()
            flip = rng.random(L) < err
            u[flip] = rng.integers(0, 4, int(flip.sum()))
            s = [ALPHA[c] for c in u]
            for b in np.nonzero(rng.random(L) < n_frac)[0]:
                s[b] = "N"
            if L > 21 and rng.random() < 0.3 * (n_frac > 0):
                s[21] = "N"
            s = "".join(s)
            out[s] = out.get(s, 0) + 1
    umis = list(out)
    freq = [out[u] for u in umis]
    umis, freq, _ = canonical(umis, freq)
    return umis, freq


def make_batch(rng, n_buckets, L, n_mol_max, err=0.02, n_frac=0.0):
    keys, nm, fr, off = [], [], [], [0]
    for _ in range(n_buckets):
        umis, freq = random_bucket(rng, int(rng.integers(1, n_mol_max + 1)), L, err, n_frac)
        k, m = orc.encode_keys_wide(umis)
        keys.append(k); nm.append(m); fr.extend(freq)
        off.append(off[-1] + len(umis))
    return np.concatenate(keys), np.concatenate(nm), np.array(fr, np.int32), np.array(off, np.uint64)


def test_encode_wide_is_to_bitset():
    import umi_collapse_rs_amd.api as api
    rng = np.random.default_rng(3)
    for L in (22, 25, 42, 43, 64, 85):
        umis = ["".join("ACGTN"[c] for c in rng.integers(0, 5, L)) for _ in range(50)]
        k, m = api.to_bitset_wide(umis, L)
        ok, om = orc.encode_keys_wide(umis)
        assert (k == ok).all() and (m == om).all(), L
    # KAT G5: the 22-bp UMI whose last base straddles words 0 and 1
    k, m = api.to_bitset_wide(["ACGTACGTACGTACGTACGTAN"], 22)
    assert k[0].tolist() == [0xaf0af0af0af0af0, 0x2] and m[0].tolist() == [0x8000000000000000, 0x3]


@pytest.mark.parametrize("L,k,n_frac,p,algo,amf", [
    (22, 1, 0.0, 0.5, 0, 0),
    (22, 1, 0.01, 0.5, 0, 0),    # N bases, also at base 21: the straddle quirk of bit_count_xor
    (30, 2, 0.005, 0.5, 0, 0),
    (42, 1, 0.0, 0.3, 0, 0),     # two full words
    (43, 3, 0.003, 1.0, 0, 0),   # three words
    (85, 2, 0.002, 0.5, 0, 0),   # four words
    (25, 1, 0.0, 0.5, 1, 0),     # adjacency, reference behaviour
    (25, 1, 0.01, 0.5, 1, 2),    # adjacency with a real max_freq
    (22, 0, 0.01, 0.5, 0, 0),
])
def test_wide_keys_against_the_oracle(ctx, L, k, n_frac, p, algo, amf):
    rng = np.random.default_rng(100 * L + k + algo)
    keys, nm, fr, off = make_batch(rng, 40, L, 60, err=0.03, n_frac=n_frac)
    # one deeper position (several 64-row chunks) and empty / single-entry ones
    kb, nb_, fb, ob = make_batch(rng, 1, L, 400, err=0.03, n_frac=n_frac)
    keys = np.concatenate([keys, kb]); nm = np.concatenate([nm, nb_]); fr = np.concatenate([fr, fb])
    off = np.concatenate([off, [off[-1]], off[-1] + ob[1:]]).astype(np.uint64)
    okept, oroot, _ = orc.dedup_batch_wide(keys, nm, fr, off, L, k, p, algo, amf)
    kept, root, st = ctx.dedup_batch_wide(keys, nm if nm.any() else None, fr, off, L, k, p, algo, amf)
    assert (kept == okept).all(), np.nonzero(kept != okept)[0][:10]
    assert (root == oroot).all()
    assert st["n_kept"] == int(okept.sum()) and st["n_umis"] == len(keys)


def test_wide_contract_violations(ctx):
    import umi_collapse_rs_amd as umi
    keys, nm = orc.encode_keys_wide(["ACGTACGTACGTACGTACGTAN", "ACGTACGTACGTACGTACGTAA"])
    off = np.array([0, 2], np.uint64)
    with pytest.raises(umi.UmiHipError):  # an N (here the straddling base) without nmask
        ctx.dedup_batch_wide(keys, None, np.array([1, 1], np.int32), off, 22)
    with pytest.raises(umi.UmiHipError):  # rank order
        ctx.dedup_batch_wide(keys, nm, np.array([1, 2], np.int32), off, 22)
    with pytest.raises(umi.UmiHipError):  # one word too few for the length
        ctx.dedup_batch_wide(keys[:, :1].copy(), None, np.array([1, 1], np.int32), off, 22)


@pytest.mark.parametrize("L,k,n_frac,words", [(24, 1, 0.0, 2), (24, 2, 0.004, 2), (45, 1, 0.002, 3), (70, 3, 0.0, 4)])
def test_wide_keys_deep_positions_through_the_segment_index(ctx, L, k, n_frac, words):
    """Positions of a few thousand UMIs of more than 21 bases: the segment index works on the first
    word's 21 bases (a pair within k overall is within k there) and every filter hit is decided on all
    words; next to them buckets for the fused kernel (<= 128) and for the exact chunk kernel.  A dual
    12 + 12 UMI is the 24-bp case.  Against the oracle, and against the all-pairs path (seg_index = 0)."""
    import umi_collapse_rs_amd as umi
    rng = np.random.default_rng(7 * L + k)
    parts = [make_batch(rng, 30, L, 50, err=0.03, n_frac=n_frac)]
    for n_mol in (1500, 300, 700):  # ~3,500 / ~700 / ~1,600 entries: segment index; chunk kernel; segment index
        # clustered: few centres with many error copies, so that neighbours within k exist at every freq
        centres = rng.integers(0, 4, (max(2, n_mol // 12), L))
        out = {}
        for _ in range(n_mol * 3):
            u = centres[rng.integers(len(centres))].copy()
            flip = rng.random(L) < 0.04
            u[flip] = rng.integers(0, 4, int(flip.sum()))
            t = [ALPHA[c] for c in u]
            for b in np.nonzero(rng.random(L) < n_frac)[0]:
                t[b] = "N"
            t = "".join(t)
            out[t] = out.get(t, 0) + 1
        umis = list(out)
        umis, freq, _ = canonical(umis, [out[u] for u in umis])
        kk, mm = orc.encode_keys_wide(umis)
        parts.append((kk, mm, np.array(freq, np.int32), np.array([0, len(umis)], np.uint64)))
    keys = np.concatenate([p_[0] for p_ in parts]); nm = np.concatenate([p_[1] for p_ in parts])
    fr = np.concatenate([p_[2] for p_ in parts])
    offs, base = [np.zeros(1, np.uint64)], np.uint64(0)
    for p_ in parts:
        offs.append(p_[3][1:] + base)
        base = base + p_[3][-1]
    off = np.concatenate(offs)
    assert keys.shape[1] == words and np.diff(off.astype(np.int64)).max() > 2000
    okept, oroot, _ = orc.dedup_batch_wide(keys, nm, fr, off, L, k, 0.5)
    nmask = nm if nm.any() else None
    kept, root, st = ctx.dedup_batch_wide(keys, nmask, fr, off, L, k, 0.5)
    assert (kept == okept).all(), np.nonzero(kept != okept)[0][:10]
    assert (root == oroot).all()
    assert st["n_pairs_evaluated"] < st["n_pairs"]  # the partition did prune
    c2 = umi.Context(0)
    try:
        c2.set_option("seg_index", 0)
        c2.set_option("fused_max", 0)
        kept2, root2, st2 = c2.dedup_batch_wide(keys, nmask, fr, off, L, k, 0.5)
        assert (kept2 == okept).all() and (root2 == oroot).all() and st2["n_pairs_evaluated"] > st["n_pairs_evaluated"]
        multi = umi.Context([0, 0, 0])
        try:
            mk, mr, mst = multi.dedup_batch_wide(keys, nmask, fr, off, L, k, 0.5)  # sharded over three workers
            assert (mk == okept).all() and (mr == oroot).all() and mst["n_kept"] == st["n_kept"]
        finally:
            multi.close()
    finally:
        c2.close()
