"""Seeded synthetic workloads of BASELINE.json / SURVEY.md 8d and the host staging
that turns reads into the hot path's input arrays (bucket by position, merge equal
UMIs counting freq, canonical rank order: src/deduplicate_sam.rs:148-176 then
src/algo/directional.rs:67-72 with first-appearance tie order).

PRNG = splitmix64 counter mode so that any rank can generate any slice."""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
# 3-bit codes of src/utils/read.rs:23-31 indexed by 2-bit base id (A,C,G,T)
_CODE3 = np.array([0b000, 0b110, 0b011, 0b101], dtype=np.uint64)
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def splitmix64(seed, idx):
    """splitmix64 output number idx (uint64 array) of the stream started at seed."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def bases_to_keys(b2):
    """b2: uint8 [n, L] base ids 0..3 -> 3-bit packed keys (BitSet.bits[0])."""
    n, L = b2.shape
    keys = np.zeros(n, dtype=np.uint64)
    for i in range(L):
        keys |= _CODE3[b2[:, i]] << np.uint64(3 * i)
    return keys


def bases_to_keys_wide(b2):
    """b2: uint8 [n, L] base ids 0..3 -> BitSet.bits as uint64 [n, ceil(3 L / 64)] (base i at bits
    3i .. 3i+2 of the word string, a base may sit across two words: src/utils/bitset.rs:17-27,52-61)."""
    n, L = b2.shape
    w = (3 * L + 63) // 64
    keys = np.zeros((n, w), dtype=np.uint64)
    for i in range(L):
        code = _CODE3[b2[:, i]]
        bit = 3 * i
        wi, sh = bit >> 6, bit & 63
        with np.errstate(over="ignore"):
            keys[:, wi] |= code << np.uint64(sh)
            if sh > 61 and wi + 1 < w:
                keys[:, wi + 1] |= code >> np.uint64(64 - sh)
    return keys


def uniform_reads(seed, n_reads, umi_len, start=0):
    """Uniform-random UMIs (the single-position variant of config 2): base ids [n, L]."""
    idx = np.arange(start, start + n_reads, dtype=np.uint64)
    z = splitmix64(seed, idx)
    out = np.empty((n_reads, umi_len), dtype=np.uint8)
    for i in range(umi_len):
        if i == 21:  # refresh the word (never needed for umi_len <= 21)
            z = splitmix64(seed ^ 0xABCDEF, idx)
        out[:, i] = (z >> np.uint64(2 * (i % 32))) & np.uint64(3)
    return out


def molecule_reads(seed, n_positions, reads_per_position, umi_len, err=0.01, first_position=0):
    """Molecule model of SURVEY.md 8d: per position M true UMIs uniform over ACGT^L,
    copies/molecule = 1+Geometric(0.5), each copy mutated per base with prob err.
    Returns (pos_id int64[n], bases uint8[n, L]) with exactly reads_per_position
    reads at every position, positions in sorted order."""
    rng = np.random.Generator(np.random.PCG64(
        int(splitmix64(seed, np.array([first_position], dtype=np.uint64))[0])))
    n = n_positions * reads_per_position
    # copies ~ 1+Geom(0.5) has mean 2: draw enough molecules, then cut to size
    n_mol = reads_per_position  # upper bound per position
    copies = rng.geometric(0.5, (n_positions, n_mol)).astype(np.int64)
    csum = np.cumsum(copies, axis=1)
    # molecule index of each read slot
    slots = np.arange(reads_per_position)
    mol_of = np.empty((n_positions, reads_per_position), dtype=np.int64)
    for p in range(n_positions):
        mol_of[p] = np.searchsorted(csum[p], slots, side="right")
    true = rng.integers(0, 4, (n_positions, n_mol, umi_len), dtype=np.uint8)
    bases = true[np.arange(n_positions)[:, None], mol_of]  # [P, R, L]
    mut = rng.random(bases.shape) < err
    shift = rng.integers(1, 4, bases.shape, dtype=np.uint8)
    bases = np.where(mut, (bases + shift) & 3, bases).astype(np.uint8)
    # shuffle reads inside a position so first-appearance order is not molecule order
    perm = np.argsort(rng.random((n_positions, reads_per_position)), axis=1)
    bases = np.take_along_axis(bases, perm[:, :, None], axis=1)
    pos = np.repeat(np.arange(first_position, first_position + n_positions, dtype=np.int64),
                    reads_per_position)
    return pos, bases.reshape(n, umi_len)


def stage(pos, keys):
    """Host staging for reads without N, `any` merge: returns dict(keys, freq,
    bucket_off, first) in canonical order -- buckets by first appearance of the
    position, UMIs by freq descending then first appearance (`first` = index of the
    first read carrying that (position, UMI))."""
    n = len(keys)
    if n == 0:
        return dict(keys=np.zeros(0, np.uint64), freq=np.zeros(0, np.int32),
                    bucket_off=np.zeros(1, np.uint64), first=np.zeros(0, np.int64))
    pos = np.asarray(pos, dtype=np.int64)
    # bucket number in first-appearance order
    upos, pfirst, pinv = np.unique(pos, return_index=True, return_inverse=True)
    brank = np.empty(len(upos), dtype=np.int64)
    brank[np.argsort(pfirst, kind="stable")] = np.arange(len(upos))
    bucket = brank[pinv]
    if keys.ndim == 2:  # keys of several words (umi_len > 21): rows compare word by word
        order = np.lexsort(tuple(keys[:, w] for w in range(keys.shape[1])) + (bucket,))
        sb, sk = bucket[order], keys[order]
        new = np.ones(n, dtype=bool)
        new[1:] = (sb[1:] != sb[:-1]) | (sk[1:] != sk[:-1]).any(axis=1)
    else:
        order = np.lexsort((keys, bucket))  # stable: equal (bucket,key) keep read order
        sb, sk = bucket[order], keys[order]
        new = np.ones(n, dtype=bool)
        new[1:] = (sb[1:] != sb[:-1]) | (sk[1:] != sk[:-1])
    starts = np.nonzero(new)[0]
    ukeys, ubucket, first = sk[starts], sb[starts], order[starts]
    freq = np.diff(np.append(starts, n)).astype(np.int64)
    # canonical order: bucket, freq desc, first appearance
    o2 = np.lexsort((first, -freq, ubucket))
    ukeys, ubucket, first, freq = ukeys[o2], ubucket[o2], first[o2], freq[o2]
    nb = int(ubucket.max()) + 1
    counts = np.bincount(ubucket, minlength=nb)
    bucket_off = np.zeros(nb + 1, dtype=np.uint64)
    bucket_off[1:] = np.cumsum(counts)
    return dict(keys=ukeys.astype(np.uint64), freq=freq.astype(np.int32), bucket_off=bucket_off,
                first=first.astype(np.int64))


def config2(seed=2, n_reads=1_000_000, umi_len=12):
    """BASELINE config 2: n_reads reads, one alignment position, uniform UMIs."""
    b2 = uniform_reads(seed, n_reads, umi_len)
    st = stage(np.zeros(n_reads, dtype=np.int64), bases_to_keys(b2))
    st["n_reads"] = n_reads
    return st


def config3(seed=3, n_reads=10_000_000, n_positions=100_000, umi_len=12, chunk=10_000):
    """BASELINE config 3: many small buckets (molecule model)."""
    rpp = n_reads // n_positions
    parts = []
    for p0 in range(0, n_positions, chunk):
        npos = min(chunk, n_positions - p0)
        pos, bases = molecule_reads(seed, npos, rpp, umi_len, first_position=p0)
        parts.append(stage(pos, bases_to_keys(bases)))
    keys = np.concatenate([p["keys"] for p in parts])
    freq = np.concatenate([p["freq"] for p in parts])
    offs, base = [np.zeros(1, np.uint64)], np.uint64(0)
    for p in parts:
        offs.append(p["bucket_off"][1:] + base)
        base = base + p["bucket_off"][-1]
    return dict(keys=keys, freq=freq, bucket_off=np.concatenate(offs), n_reads=rpp * n_positions)


def config2m(seed=22, n_reads=1_000_000, umi_len=12, n_molecules=100_000, err=0.01):
    """One deep alignment position from the molecule model (SURVEY.md 8d's generator at one
    position): n_molecules true UMIs uniform over ACGT^L, copies per molecule ~ 1 + Geometric
    with mean n_reads / n_molecules (cut to exactly n_reads reads), every copy's UMI mutated per
    base with probability err to one of the other three bases.  Unlike config 2's uniform UMIs the
    unique UMIs come in clusters: a true UMI of high freq surrounded by its freq-1 error copies."""
    rng = np.random.Generator(np.random.PCG64(int(splitmix64(seed, np.array([7], dtype=np.uint64))[0])))
    mean = max(1.0, n_reads / n_molecules)
    copies = rng.geometric(1.0 / mean, n_molecules).astype(np.int64)
    csum = np.cumsum(copies)
    if csum[-1] < n_reads:  # top up the last molecules
        copies[-1] += n_reads - csum[-1]
        csum = np.cumsum(copies)
    mol_of = np.searchsorted(csum, np.arange(n_reads), side="right")
    true = rng.integers(0, 4, (n_molecules, umi_len), dtype=np.uint8)
    bases = true[mol_of]
    mut = rng.random(bases.shape) < err
    shift = rng.integers(1, 4, bases.shape, dtype=np.uint8)
    bases = np.where(mut, (bases + shift) & 3, bases).astype(np.uint8)
    bases = bases[rng.permutation(n_reads)]  # first-appearance order is not molecule order
    st = stage(np.zeros(n_reads, dtype=np.int64), bases_to_keys(bases) if umi_len <= 21 else bases_to_keys_wide(bases))
    st["n_reads"] = n_reads
    st["n_molecules"] = int(mol_of.max()) + 1
    return st
