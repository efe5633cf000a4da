"""Shared helpers for the tests: seeded UMI bucket generators and an independent
brute-force model of the collapse (numpy, written from the algorithm's
definition -- not from the oracle's code)."""
import numpy as np

ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)


def hamming_matrix(umis):
    """Hamming distance over the 5-letter alphabet straight from the characters."""
    a = np.array([np.frombuffer(u.encode(), dtype=np.uint8) for u in umis])
    return (a[:, None, :] != a[None, :, :]).sum(-1)


def thr_f32(p, f):
    return int(np.float32(p) * np.float32(f + 1))


def brute_directional(umis, freq, k, p):
    """Survivors (input indices, output order) and root per input index.
    Definition: process UMIs in stable freq-descending order; a UMI still present
    becomes a root and removes everything reachable through edges u->v with
    dist(u,v)<=k and freq[v] <= thr(freq[u])."""
    n = len(umis)
    order = sorted(range(n), key=lambda i: (-freq[i], i))
    if n == 0:
        return [], []
    d = hamming_matrix(umis)
    thr = np.array([thr_f32(p, f) for f in freq])
    fr = np.array(freq)
    adj = (d <= k) & (fr[None, :] <= thr[:, None])
    np.fill_diagonal(adj, False)
    present = np.ones(n, bool)
    root_of = list(range(n))
    surv = []
    for r in order:
        if not present[r]:
            continue
        surv.append(r)
        present[r] = False
        frontier = [r]
        while frontier:
            nxt = []
            for u in frontier:
                vs = np.nonzero(adj[u] & present)[0]
                present[vs] = False
                for v in vs:
                    root_of[v] = r
                nxt.extend(vs.tolist())
            frontier = nxt
    return surv, root_of


def brute_adjacency(umis, freq, k, max_freq):
    n = len(umis)
    order = sorted(range(n), key=lambda i: (-freq[i], i))
    if n == 0:
        return [], []
    d = hamming_matrix(umis)
    fr = np.array(freq)
    present = np.ones(n, bool)
    root_of = list(range(n))
    surv = []
    for r in order:
        if not present[r]:
            continue
        surv.append(r)
        present[r] = False
        vs = np.nonzero((d[r] <= k) & (fr <= max_freq) & present)[0]
        present[vs] = False
        for v in vs:
            root_of[v] = r
    return surv, root_of


def random_bucket(rng, n_mol, L, err=0.05, mean_copies=3.0, n_frac=0.0):
    """Molecule model: n_mol true UMIs, geometric copy counts, per-base errors.
    Returns (umis in first-appearance order, freq)."""
    seen = {}
    for _ in range(n_mol):
        true = rng.choice(ALPHA, L)
        copies = int(rng.geometric(1.0 / mean_copies))
        for _ in range(copies):
            u = true.copy()
            mut = rng.random(L) < err
            for i in np.nonzero(mut)[0]:
                u[i] = rng.choice(ALPHA[ALPHA != u[i]])
            if n_frac:
                u[rng.random(L) < n_frac] = ord("N")
            s = u.tobytes().decode()
            seen[s] = seen.get(s, 0) + 1
    umis = list(seen.keys())
    return umis, [seen[u] for u in umis]


def canonical(umis, freq):
    """Stable freq-descending order (the rank order the batched ABI expects)."""
    order = sorted(range(len(umis)), key=lambda i: (-freq[i], i))
    return [umis[i] for i in order], [freq[i] for i in order], order


# ---- shipped library vs development build ---------------------------------------------------------
# The round-1 tile kernels and their options live in libumihip_dev.so only (make dev, -DUMIHIP_DEV).
# Run the legacy cross-checks with UMIHIP_LIB=<repo>/umi_collapse_rs_amd/libumihip_dev.so; under the
# shipped library they are skipped and option sets that name a legacy option are left out.
LEGACY_OPTS = {"prune", "bs_unit", "bs_sorted", "bs_tables", "two_phase", "bs_col_chunk", "bs_tab_min_run",
               "bs_transposed", "bs_tab_waves", "bitslice", "ovf_capacity"}


def is_dev_build():
    import os
    from umi_collapse_rs_amd import _lib
    return os.path.basename(_lib.LIB_PATH) == "libumihip_dev.so"


def usable(opts):
    """Can this option set be applied to the library under test?"""
    return is_dev_build() or not (set(opts) & LEGACY_OPTS)


def legacy_mark():
    import pytest
    return pytest.mark.skipif(not is_dev_build(), reason="round-1 tile kernels: development build only "
                                                          "(UMIHIP_LIB=.../libumihip_dev.so)")


def stage_model(pos, umis, score, merge):
    """deduplicate_sam.rs:148-176 + the canonical order, straight from the definition (a dict per
    position, plain Python: for a few 10^4 reads): positions by first appearance, a position's UMIs
    by freq descending then first appearance; per UMI its freq and the read that stands for it
    (merge 0: the first; 1: the highest score, the earlier on ties).  Returns (list of UMI strings,
    freq, rep, bucket_off)."""
    buckets = {}
    for i, (p, u) in enumerate(zip(pos, umis)):
        d = buckets.setdefault(int(p), {})
        e = d.get(u)
        sc = 0 if score is None else int(score[i])
        if e is None:
            d[u] = [1, i, sc]  # freq, rep, score of the rep
        else:
            e[0] += 1
            if merge and not (e[2] >= sc):
                e[1], e[2] = i, sc
    out_umis, freq, rep, off = [], [], [], [0]
    for p, d in buckets.items():  # (dicts keep insertion order: first appearance)
        items = sorted(d.items(), key=lambda kv: -kv[1][0])  # stable
        out_umis += [u for u, _ in items]
        freq += [e[0] for _, e in items]
        rep += [e[1] for _, e in items]
        off.append(len(out_umis))
    return out_umis, np.array(freq, np.int32), np.array(rep, np.uint64), np.array(off, np.uint64)
