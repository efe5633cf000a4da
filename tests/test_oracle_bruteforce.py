"""Oracle vs an independent brute-force model on random buckets (CPU only)."""
import numpy as np
import pytest

import oracle as orc
from helpers import brute_adjacency, brute_directional, canonical, random_bucket


@pytest.mark.parametrize("L,k,p,n_frac", [
    (6, 1, 0.5, 0.0), (6, 2, 0.5, 0.0), (12, 1, 0.5, 0.0), (12, 0, 0.5, 0.0),
    (12, 1, 0.3, 0.0), (12, 2, 1.0, 0.0), (8, 1, 0.5, 0.05), (20, 2, 0.5, 0.02),
    (21, 3, 0.5, 0.0), (5, 1, 0.75, 0.1),
])
def test_directional_matches_bruteforce(L, k, p, n_frac):
    rng = np.random.default_rng(1000 * L + 10 * k + int(p * 10))
    for trial in range(6):
        umis, freq = random_bucket(rng, n_mol=int(rng.integers(1, 60)), L=L, err=0.08,
                                   n_frac=n_frac)
        surv, root_of, _ = orc.apply_strings(umis, freq, k, "dir", p)
        bs, br = brute_directional(umis, freq, k, p)
        assert surv == bs
        assert root_of == br


@pytest.mark.parametrize("max_freq", [0, 1, 2, 1 << 30])
def test_adjacency_matches_bruteforce(max_freq):
    rng = np.random.default_rng(77 + max_freq % 97)
    for trial in range(6):
        umis, freq = random_bucket(rng, n_mol=int(rng.integers(1, 60)), L=8, err=0.1)
        surv, root_of, _ = orc.apply_strings(umis, freq, 1, "adj", adj_max_freq=max_freq)
        bs, br = brute_adjacency(umis, freq, 1, max_freq)
        assert surv == bs
        assert root_of == br


def test_survivor_is_min_rank_reaching_node():
    """The identity the device collapse relies on: v survives iff no
    node of smaller rank reaches it; root(v) = the smallest rank that reaches v."""
    rng = np.random.default_rng(5)
    for p in (0.5, 1.0, 0.3):
        umis, freq = random_bucket(rng, n_mol=40, L=7, err=0.15)
        umis, freq, _ = canonical(umis, freq)
        n = len(umis)
        from helpers import hamming_matrix, thr_f32
        d = hamming_matrix(umis)
        thr = np.array([thr_f32(p, f) for f in freq])
        adj = (d <= 1) & (np.array(freq)[None, :] <= thr[:, None])
        reach = adj | np.eye(n, dtype=bool)
        for _ in range(n):
            reach = reach | ((reach.astype(np.int32) @ reach.astype(np.int32)) > 0)
        label = np.array([np.nonzero(reach[:, v])[0].min() for v in range(n)])
        surv, root_of, _ = orc.apply_strings(umis, freq, 1, "dir", p)
        assert surv == np.nonzero(label == np.arange(n))[0].tolist()
        assert root_of == label.tolist()


def test_batched_oracle_equals_per_bucket():
    rng = np.random.default_rng(11)
    keys, nm, fr, off = [], [], [], [0]
    expect_kept = []
    for b in range(20):
        umis, freq = random_bucket(rng, n_mol=int(rng.integers(0, 30)), L=12, err=0.05,
                                   n_frac=0.01)
        umis, freq, _ = canonical(umis, freq)
        kk, nn = orc.encode_keys(umis)
        keys.append(kk); nm.append(nn); fr.extend(freq)
        off.append(off[-1] + len(umis))
        surv, _, _ = orc.apply_strings(umis, freq, 1, "dir")
        m = np.zeros(len(umis), np.uint8); m[surv] = 1
        expect_kept.append(m)
    keys = np.concatenate(keys); nm = np.concatenate(nm)
    kept, root, calls = orc.dedup_batch(keys, nm, fr, off, 12, 1)
    assert kept.tolist() == np.concatenate(expect_kept).tolist()
    assert calls > 0
    assert (kept == (root == np.arange(len(kept)))).all()


def test_batched_oracle_rejects_unranked_input():
    keys, nm = orc.encode_keys(["AAAA", "AAAT"])
    with pytest.raises(ValueError):
        orc.dedup_batch(keys, nm, [1, 2], [0, 2], 4, 1)


def test_stage_reads_merge_and_order():
    # bucket 7 appears first; UMI order inside = freq desc then first appearance
    bid = [7, 7, 3, 7, 3, 7, 7]
    umis = ["AAAA", "CCCC", "GGGG", "CCCC", "GGGG", "TTTT", "CCCC"]
    score = [30, 20, 10, 25, 10, 5, 25]
    ub = np.frombuffer("".join(umis).encode(), dtype=np.uint8)
    st = orc.stage_reads(bid, ub, score, 4, merge=1)
    assert st["bucket_off"].tolist() == [0, 3, 4]
    k, _ = orc.encode_keys(["CCCC", "AAAA", "TTTT", "GGGG"])
    assert st["keys"].tolist() == k.tolist()
    assert st["freq"].tolist() == [3, 1, 1, 2]
    # CCCC: read1(20) -> read3(25) replaces -> read6(25) ties keep existing (merge/mod.rs:35)
    assert st["rep"].tolist() == [3, 0, 5, 2]
    st_any = orc.stage_reads(bid, ub, score, 4, merge=0)
    assert st_any["rep"].tolist() == [1, 0, 5, 2]


@pytest.mark.parametrize("L,k,n_frac", [(22, 1, 0.0), (22, 2, 0.03), (30, 1, 0.02), (42, 3, 0.01)])
def test_multi_word_keys_match_bruteforce(L, k, n_frac):
    """UMIs of 22..42 bases: two words per key.  Below 43 bases at most one base straddles two
    words, and bit_count_xor's per-word popcount(x)/3 (bitset.rs:85-87) still gives the
    5-letter Hamming distance after the final /2 -- the character-level model applies.  Both the
    per-bucket apply and the batched multi-word form."""
    rng = np.random.default_rng(50 * L + k)
    keys, nm, fr, off, expect = [], [], [], [0], []
    for trial in range(8):
        umis, freq = random_bucket(rng, n_mol=int(rng.integers(1, 40)), L=L, err=0.06, n_frac=n_frac)
        if n_frac and len(umis) > 2:  # an N exactly at the straddling base
            umis[0] = umis[0][:21] + "N" + umis[0][22:]
            if len(set(umis)) != len(umis):
                continue
        surv, root_of, _ = orc.apply_strings(umis, freq, k, "dir", 0.5)
        bs, br = brute_directional(umis, freq, k, 0.5)
        assert surv == bs and root_of == br
        cu, cf, _ = canonical(umis, freq)
        kk, nn = orc.encode_keys_wide(cu)
        assert kk.shape[1] == 2
        keys.append(kk); nm.append(nn); fr.extend(cf)
        off.append(off[-1] + len(cu))
        s2, _ = brute_directional(cu, cf, k, 0.5)
        m = np.zeros(len(cu), np.uint8); m[s2] = 1
        expect.append(m)
    kept, root, _ = orc.dedup_batch_wide(np.concatenate(keys), np.concatenate(nm), fr, off, L, k)
    assert kept.tolist() == np.concatenate(expect).tolist()
    assert (kept == (root == np.arange(len(kept)))).all()


def test_straddling_bases_quirk_of_the_reference():
    """The two bases that straddle words within 85 bases (21: bits 63|64,65; 42: bits 126,127|128)
    both N-mismatching: the mask bits of word 0 and word 2 (one each) are lost to popcount(x)/3,
    the four of word 1 give 4/3 = 1, so bit_count_xor = 6 - 1 = 5 where whole bases would give 4;
    after the final /2 the distance is still 2 -- with at most these two straddlers the quirk of
    bitset.rs:85-87 never changes umi_dist (SURVEY.md 8a A2 expected it to, from 43 bases on)."""
    a = "A" * 21 + "N" + "A" * 20 + "N" + "A"
    b = "A" * 44
    ba, bb = orc.to_bitset(a), orc.to_bitset(b)
    assert ba.nwords == 3
    assert orc.lib().orc_bit_count_xor(ba, bb) == 5 and orc.lib().orc_umi_dist(ba, bb) == 2
    one = orc.to_bitset("A" * 21 + "N" + "A" * 22)
    assert orc.lib().orc_bit_count_xor(one, bb) == 3 and orc.lib().orc_umi_dist(one, bb) == 1
