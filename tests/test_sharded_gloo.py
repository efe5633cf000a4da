"""Multi-rank path on CPU: world_size 2 over gloo.  The per-rank compute callback is the
oracle here (test infrastructure); the product default is the HIP path."""
import os
import socket
import sys

import numpy as np
import pytest

import oracle as orc
from helpers import canonical, random_bucket

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_job(seed, n_buckets):
    rng = np.random.default_rng(seed)
    keys, fr, off = [], [], [0]
    for _ in range(n_buckets):
        umis, freq = random_bucket(rng, int(rng.integers(0, 40)), 10, err=0.08)
        umis, freq, _ = canonical(umis, freq)
        k, _ = orc.encode_keys(umis)
        keys.append(k); fr.extend(freq); off.append(off[-1] + len(umis))
    return (np.concatenate(keys), np.array(fr, np.int32), np.array(off, np.uint64))


def _worker(rank, world, port, seed, n_buckets, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from umi_collapse_rs_amd.sharded import ShardedDedup
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keys, fr, off = _make_job(seed, n_buckets)

    def compute(k, nm, f, o):
        return orc.dedup_batch(k, nm, f, o, 10, 1)[0]

    kept = ShardedDedup(dist, compute).run(keys, None, fr, off)
    q.put((rank, kept.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_buckets", [1, 7, 40])
def test_two_ranks_reassemble_the_global_mask(n_buckets):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 123, n_buckets, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    keys, fr, off = _make_job(123, n_buckets)
    expect = orc.dedup_batch(keys, None, fr, off, 10, 1)[0]
    for r in range(2):
        assert np.frombuffer(got[r], np.uint8).tolist() == expect.tolist()


def test_partition_is_deterministic_balanced_and_complete():
    from umi_collapse_rs_amd.sharded import partition_buckets, shard_arrays
    rng = np.random.default_rng(5)
    sizes = rng.integers(0, 500, 1000)
    sizes[17] = 20000  # one giant bucket dominates the n^2 cost
    for world in (1, 2, 3, 8):
        parts = partition_buckets(sizes, world)
        assert sorted(np.concatenate(parts).tolist()) == list(range(1000))
        again = partition_buckets(sizes, world)
        assert all((a == b).all() for a, b in zip(parts, again))
        cost = [float((sizes[p].astype(np.float64) ** 2).sum()) for p in parts]
        rest = sorted(cost)[:-1] if world > 1 else cost
        if world > 2:
            assert max(rest) / max(1.0, min(rest)) < 1.2  # the non-giant ranks are balanced
    off = np.zeros(1001, np.uint64)
    off[1:] = np.cumsum(sizes)
    keys = np.arange(int(off[-1]), dtype=np.uint64)
    k, _, f, loff, gidx = shard_arrays(keys, None, keys.astype(np.int32), off, parts[1])
    assert (k == gidx).all() and int(loff[-1]) == len(k)


def test_library_partition_matches_the_python_reference():
    """umi_partition_buckets (C++, what the multi-device context and the one-process-per-GPU hosts
    use) against the exact-integer Python restatement of the same rule."""
    from umi_collapse_rs_amd.api import partition_buckets as lib_partition
    from umi_collapse_rs_amd.sharded import partition_buckets_py
    rng = np.random.default_rng(11)
    for trial in range(20):
        nb = int(rng.integers(0, 400))
        sizes = rng.integers(0, 50, nb) if trial % 3 else rng.integers(0, 3, nb) * rng.integers(1, 100000, nb)
        if nb and trial % 4 == 0:
            sizes[rng.integers(nb)] = 2_000_000_000  # n^2 near 2^62: the loads saturate, not wrap
        off = np.zeros(nb + 1, np.uint64)
        off[1:] = np.cumsum(sizes.astype(np.uint64))
        for world in (1, 2, 3, 8, 64):
            assert lib_partition(off, world).tolist() == partition_buckets_py(sizes, world).tolist(), (trial, world)
