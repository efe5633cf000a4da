"""Seeded differential fuzz of the batched entry point against the oracle: random UMI
lengths, k, percentage, algorithm, N density, and bucket-size mixes that straddle every kernel
boundary (fused 64/128, chunk kernel 1024, column-split tiles, 4096-row tile edges)."""
import numpy as np
import pytest

import oracle as orc
from helpers import usable

pytestmark = pytest.mark.gpu
ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)


def clustered_bucket(rng, n_target, L, n_frac):
    """~n_target distinct UMIs: a few centres with many 1-2 error neighbours + random ones."""
    out = {}
    centres = rng.choice(ALPHA, (max(1, n_target // 40), L))
    while len(out) < n_target:
        if rng.random() < 0.7:
            u = centres[rng.integers(len(centres))].copy()
            for _ in range(int(rng.integers(0, 3))):
                u[rng.integers(L)] = rng.choice(ALPHA)
        else:
            u = rng.choice(ALPHA, L)
        if n_frac and rng.random() < n_frac:
            u[rng.integers(L)] = ord("N")
        s = u.tobytes()
        out[s] = out.get(s, 0) + int(rng.geometric(0.4))
        if len(out) >= 4 ** L - 1 and L < 6:
            break
    umis = list(out.keys())
    freq = np.array([out[u] for u in umis])
    order = np.lexsort((np.arange(len(umis)), -freq))
    return [umis[i].decode() for i in order], freq[order].tolist()


@pytest.mark.parametrize("seed", range(30))
def test_fuzz_against_oracle(seed):
    import umi_collapse_rs_amd as umi
    rng = np.random.default_rng(9000 + seed)
    L = int(rng.choice([4, 6, 8, 9, 11, 12, 13, 16, 17, 20, 21]))
    k = int(rng.choice([0, 1, 1, 1, 2, 2, 3, 4]))
    p = float(rng.choice([0.5, 0.5, 0.3, 0.75, 1.0, 0.0]))
    algo, amf = (0, 0) if rng.random() < 0.75 else (1, int(rng.choice([0, 1, 3])))
    n_frac = float(rng.choice([0.0, 0.0, 0.02, 0.2]))
    sizes = [int(x) for x in rng.choice([0, 1, 2, 31, 63, 64, 65, 127, 128, 129, 300, 1023, 1024,
                                         1025, 2000, 4095, 4097, 6000], size=int(rng.integers(3, 9)))]
    cap = 4 ** L // 2
    keys, nm, fr, off = [], [], [], [0]
    for n_target in sizes:
        umis, freq = clustered_bucket(rng, min(n_target, cap), L, n_frac) if n_target else ([], [])
        kk, mm = orc.encode_keys(umis)
        keys.append(kk); nm.append(mm); fr.extend(freq); off.append(off[-1] + len(umis))
    keys, nm = np.concatenate(keys), np.concatenate(nm)
    fr, off = np.array(fr, np.int32), np.array(off, np.uint64)
    okept, oroot, _ = orc.dedup_batch(keys, nm, fr, off, L, k, p, algo, amf)
    for opts in ({}, {"prune": 1}, {"bitslice": 0, "fused_max": 0}, {"bs_unit": 1, "small_max": 200, "seg_index": 0},
                 {"seg_index": 0}, {"seg_min": 129, "two_phase": 1}, {"seg_min": 129, "fused_max": 0}):
        if not usable(opts):
            continue
        ctx = umi.Context(0)
        try:
            for name, v in opts.items():
                ctx.set_option(name, v)
            kept, root, st = ctx.dedup_batch(keys, nm if nm.any() else None, fr, off, L, k, p, algo, amf)
        finally:
            ctx.close()
        assert (kept == okept).all(), (seed, L, k, p, algo, amf, opts, np.nonzero(kept != okept)[0][:5])
        assert (root == oroot).all(), (seed, L, k, p, algo, amf, opts)


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_wide_buckets(seed):
    """The same differential fuzz on buckets of 33k-50k entries (key-sorted table kernel, mask
    kernel with cached prefix state, 64-bit keys), next to a few small buckets in one call."""
    import umi_collapse_rs_amd as umi
    rng = np.random.default_rng(7000 + seed)
    L = int(rng.choice([8, 9, 10, 11, 12, 13, 17, 20]))
    k = int(rng.choice([0, 1, 1, 2, 3]))
    p = float(rng.choice([0.5, 0.5, 0.3, 1.0]))
    algo, amf = (0, 0) if rng.random() < 0.7 else (1, int(rng.choice([1, 3])))
    n_frac = float(rng.choice([0.0, 0.0, 0.01]))
    cap = 4 ** L // 2
    sizes = [int(rng.integers(33_000, 50_000)), 70, 1, int(rng.integers(2_000, 5_000))]
    rng.shuffle(sizes)
    keys, nm, fr, off = [], [], [], [0]
    for n_target in sizes:
        umis, freq = clustered_bucket(rng, min(n_target, cap), L, n_frac)
        kk, mm = orc.encode_keys(umis)
        keys.append(kk); nm.append(mm); fr.extend(freq); off.append(off[-1] + len(umis))
    keys, nm = np.concatenate(keys), np.concatenate(nm)
    fr, off = np.array(fr, np.int32), np.array(off, np.uint64)
    okept, oroot, _ = orc.dedup_batch(keys, nm, fr, off, L, k, p, algo, amf)
    for opts in ({}, {"seg_index": 0}, {"seg_index": 0, "bs_tables": 0}, {"prune": 1}, {"two_phase": 1}):
        if not usable(opts):
            continue
        ctx = umi.Context(0)
        try:
            for name, v in opts.items():
                ctx.set_option(name, v)
            kept, root, st = ctx.dedup_batch(keys, nm if nm.any() else None, fr, off, L, k, p, algo, amf)
        finally:
            ctx.close()
        assert (kept == okept).all(), (seed, L, k, p, algo, amf, opts, np.nonzero(kept != okept)[0][:5])
        assert (root == oroot).all(), (seed, L, k, p, algo, amf, opts)
