"""Multi-GPU split of one call's pairs (umi_pairs_partial_device + umi_collapse_edges_device),
rehearsed on one GPU: the parts are evaluated one after the other, their edge lists are
concatenated as an all-gather would, and the collapse of the union must equal the
single-call result and the oracle."""
import numpy as np
import pytest

import oracle as orc
from helpers import canonical, random_bucket

pytestmark = pytest.mark.gpu


def run_split(ctx, keys, nm, fr, off, L, k, p, algo, amf, n_parts, cap):
    import torch
    dev = torch.device("cuda:0")
    n = len(keys)
    t_keys = torch.from_numpy(keys.view(np.int64)).to(dev)
    t_nm = torch.from_numpy(nm.view(np.int64)).to(dev) if nm.any() else None
    t_fr = torch.from_numpy(fr).to(dev)
    parts = []
    for part in range(n_parts):
        buf = torch.zeros(cap, dtype=torch.int64, device=dev)
        ne, _ = ctx.pairs_partial_device(t_keys.data_ptr(), t_nm.data_ptr() if t_nm is not None else 0,
                                         t_fr.data_ptr(), off, L, part, n_parts, buf.data_ptr(), cap,
                                         k=k, percentage=p, algo=algo, adj_max_freq=amf)
        parts.append(buf[:ne].clone())
    edges = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=dev)
    t_kept = torch.zeros(n, dtype=torch.uint8, device=dev)
    t_root = torch.zeros(n, dtype=torch.int32, device=dev)
    st = ctx.collapse_edges_device(n, edges.data_ptr() if len(edges) else 0, len(edges),
                                   t_kept.data_ptr(), t_root.data_ptr(), algo=algo)
    torch.cuda.synchronize()
    return t_kept.cpu().numpy(), t_root.cpu().numpy().view(np.uint32), [len(x) for x in parts], st


@pytest.mark.parametrize("n_parts", [2, 3, 8])
def test_split_equals_single_call_and_oracle(n_parts):
    import umi_collapse_rs_amd as umi
    from umi_collapse_rs_amd import _lib
    ctx = umi.Context(0)
    try:
        rng = np.random.default_rng(60 + n_parts)
        # one large bucket (bit-sliced tiles), a mid one (chunk kernel) and small ones
        L = 8
        raw = rng.integers(0, 4, (9000, L))
        big = sorted({"".join("ACGT"[c] for c in r) for r in raw})
        rng.shuffle(big)
        fbig = np.minimum(rng.geometric(0.5, len(big)), 20).tolist()
        buckets = [canonical(big, fbig)[:2]]
        for n_mol in (400, 30, 1, 70):
            u, f = random_bucket(rng, n_mol, L, err=0.08, n_frac=0.01)
            buckets.append(canonical(u, f)[:2])
        keys, nm, fr, off = [], [], [], [0]
        for u, f in buckets:
            kk, mm = orc.encode_keys(u)
            keys.append(kk); nm.append(mm); fr.extend(f); off.append(off[-1] + len(u))
        keys, nm = np.concatenate(keys), np.concatenate(nm)
        fr, off = np.array(fr, np.int32), np.array(off, np.uint64)
        for k, p, algo, amf in ((1, 0.5, 0, 0), (2, 1.0, 0, 0), (1, 0.5, 1, 2)):
            kept, root, counts, st = run_split(ctx, keys, nm, fr, off, L, k, p, algo, amf, n_parts,
                                               cap=1 << 22)
            okept, oroot, _ = orc.dedup_batch(keys, nm, fr, off, L, k, p, algo, amf)
            assert (kept == okept).all() and (root == oroot).all()
            assert st["n_kept"] == int(okept.sum())
            assert sum(1 for c in counts if c > 0) >= min(2, n_parts)  # the work really was split
        with pytest.raises(umi.UmiHipError) as e:  # buffer too small: reported, nothing copied
            run_split(ctx, keys, nm, fr, off, L, 1, 0.5, 0, 0, 2, cap=4)
        assert e.value.code == _lib.UMI_ERR_NOMEM
    finally:
        ctx.close()
