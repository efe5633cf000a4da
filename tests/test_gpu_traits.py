"""The per-bucket trait path (DataStruct / Algorithm mirror) against the oracle."""
import numpy as np
import pytest

import oracle as orc
from helpers import random_bucket

pytestmark = pytest.mark.gpu


def test_hipnaive_remove_near_sequence_matches_naive():
    import umi_collapse_rs_amd as umi
    rng = np.random.default_rng(8)
    for L, k, n_frac in ((8, 1, 0.0), (8, 2, 0.05), (12, 1, 0.0), (20, 3, 0.02), (24, 2, 0.02), (50, 1, 0.0)):
        # (24 and 50 bases: keys of two and three words, umi_data_new_wide)
        umis, freq = random_bucket(rng, 60, L, err=0.1, n_frac=n_frac)
        d = umi.HipNaive.new(dict(zip(umis, freq)), L, k)
        o = orc.Naive(umis, freq)
        assert d.stats() == {}
        for q in rng.permutation(len(umis))[:40]:
            kk = int(rng.integers(0, k + 1))
            mf = int(rng.integers(0, 6))
            got = d.remove_near(umis[q], kk, mf)
            exp = {umis[i] for i in o.remove_near(int(q), kk, mf)}
            assert got == exp
            assert all(d.contains(u) == o.contains(i) for i, u in enumerate(umis))
        assert not d.contains("A" * L) or ("A" * L) in umis


def test_hipnaive_on_a_wide_bucket():
    """DataStruct path over a bucket large enough for the key-sorted table kernel: the
    neighbour lists (with distances) come back in entry indices through the permutation."""
    import umi_collapse_rs_amd as umi
    rng = np.random.default_rng(18)
    L, k = 10, 2
    raw = rng.integers(0, 4, (42000, L))
    umis = sorted({"".join("ACGT"[c] for c in r) for r in raw})
    rng.shuffle(umis)
    freq = np.minimum(rng.geometric(0.5, len(umis)), 9).tolist()
    assert len(umis) >= 32768
    d = umi.HipNaive.new(dict(zip(umis, freq)), L, k)
    o = orc.Naive(umis, freq)
    for q in rng.permutation(len(umis))[:25]:
        kk = int(rng.integers(0, k + 1))
        mf = int(rng.integers(0, 6))
        got = d.remove_near(umis[q], kk, mf)
        exp = {umis[i] for i in o.remove_near(int(q), kk, mf)}
        assert got == exp
    probe = rng.permutation(len(umis))[:2000]
    assert all(d.contains(umis[i]) == o.contains(int(i)) for i in probe)


def test_algorithm_mirror_over_hipnaive(kat):
    import umi_collapse_rs_amd as umi
    g = kat["G8_bucket"]
    reads = {u: umi.ReadFreq("read_%d" % i, f) for i, (u, f) in enumerate(zip(g["umis"], g["freq"]))}
    out = umi.Directional(k=g["k"], percentage=g["p"]).apply(reads, None, 12)
    assert out == ["read_%d" % i for i in g["dir"]]
    out = umi.Adjacency(k=g["k"]).apply(reads, None, 12)
    assert out == ["read_%d" % i for i in g["adj"]]
    rng = np.random.default_rng(9)
    for p in (0.5, 0.3, 1.0):
        umis, freq = random_bucket(rng, 50, 10, err=0.1)
        reads = {u: umi.ReadFreq(i, f) for i, (u, f) in enumerate(zip(umis, freq))}
        tracker = {}
        out = umi.Directional(k=1, percentage=p, track_cluster=True).apply(reads, tracker, 10)
        surv, root_of, _ = orc.apply_strings(umis, freq, 1, "dir", p)
        assert out == surv
        got_root = {}
        for r, members in tracker.items():
            for m in members:
                got_root[m] = r
        assert [umis.index(got_root[u]) for u in umis] == root_of
        out = umi.Adjacency(k=1, max_freq=2).apply(reads, None, 10)
        assert out == orc.apply_strings(umis, freq, 1, "adj", adj_max_freq=2)[0]


def test_remove_near_rejects_k_above_max_edits():
    import umi_collapse_rs_amd as umi
    from umi_collapse_rs_amd import _lib
    d = umi.HipNaive.new({"AAAA": 1, "AAAT": 1}, 4, 1)
    with pytest.raises(umi.UmiHipError) as e:
        d.remove_near("AAAA", 2, 5)
    assert e.value.code == _lib.UMI_ERR_ARG
