"""Oracle vs the known-answer vectors G1..G10 (tests/golden/kat.json).

The reference has no tests of its own: these vectors are
hand-derived from the cited lines, so parity is "unpinned" by the reference."""
import numpy as np
import pytest

import oracle as orc


def _hex(xs):
    return None if xs is None else [int(x, 16) for x in xs]


def test_g1_g5_encode_and_hash(kat):
    for v in kat["G1_G5_encode"]:
        b = orc.to_bitset(v["umi"])
        assert orc.bits_of(b) == _hex(v["bits"]), v["umi"]
        assert orc.nbits_of(b) == _hex(v["nbits"]), v["umi"]
        if v["hash"] is not None:
            assert orc.bitset_hash(b) == v["hash"], v["umi"]


def test_encode_rejects_what_the_reference_panics_on():
    # utils/mod.rs:77-79: anything outside ATCGN (lowercase included) panics
    for bad in ["ACGU", "acgt", "ACG-", "ACG "]:
        with pytest.raises(ValueError):
            orc.to_bitset(bad)


def test_g5_g6_distances(kat):
    for v in kat["G5_straddle"] + kat["G6_dist"]:
        a, b = orc.to_bitset(v["a"]), orc.to_bitset(v["b"])
        assert orc.bit_count_xor(a, b) == v["bit_count_xor"], v
        assert orc.bit_count_xor(b, a) == v["bit_count_xor"], v
        assert orc.umi_dist(a, b) == v["dist"], v


def test_g7_threshold(kat):
    for v in kat["G7_threshold"]:
        got = [orc.threshold(v["p"], f) for f in v["freq"]]
        assert got == v["thr"]


def test_g8_bucket(kat):
    g = kat["G8_bucket"]
    surv, root_of, _ = orc.apply_strings(g["umis"], g["freq"], g["k"], "dir", g["p"])
    assert surv == g["dir"]
    # U1,U5 fall to U0; U2 via U1; U3 via U2; U4 alone
    assert root_of == [0, 0, 0, 0, 4, 0]
    surv, _, _ = orc.apply_strings(g["umis"], g["freq"], g["k"], "adj", g["p"])
    assert surv == g["adj"] == g["rank"]


def test_g9_tie(kat):
    g = kat["G9_tie"]
    assert orc.apply_strings(g["umis"], g["freq"], 1, "dir", g["p"])[0] == g["dir_k1"]
    assert orc.apply_strings(g["umis"], g["freq"], 0, "dir", g["p"])[0] == g["dir_k0"]


def test_g10_avgqual(kat):
    for v in kat["G10_avgqual"]:
        assert orc.avg_qual(v["quals"]) == v["avg"]


def test_naive_remove_near_predicate():
    # naive.rs:31: dist<=k && (dist==0 || f<=max_freq); the query always goes
    umis = ["AAAAAAAAAAAA", "AAAAAAAAAAAT", "AAAAAAAAAATT", "AAAAAAAAAAAC"]
    freq = [5, 2, 1, 4]
    d = orc.Naive(umis, freq)
    assert d.remove_near(0, 1, 3) == [0, 1]          # U3 has freq 4 > 3, U2 is at dist 2
    assert not d.contains(0) and not d.contains(1) and d.contains(2) and d.contains(3)
    assert d.remove_near(1, 1, 1) == [2]             # query already gone: only neighbours
    assert d.remove_near(3, 1, 0) == [3]             # max_freq 0: only itself (adjacency.rs:56)
    assert d.remove_near(3, 1, 100) == []


def test_adjacency_reference_semantics_keep_every_umi():
    # remove_near(umi,k,0) never removes a neighbour (freq>=1)
    rng = np.random.default_rng(7)
    umis = list({"".join(rng.choice(list("ACGT"), 6)) for _ in range(300)})
    freq = rng.integers(1, 6, len(umis)).tolist()
    surv, _, _ = orc.apply_strings(umis, freq, 1, "adj")
    order = sorted(range(len(umis)), key=lambda i: (-freq[i], i))
    assert surv == order
