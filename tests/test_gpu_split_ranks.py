"""split_dedup_device with two real ranks (gloo process group, both on cuda:0): the
all-gatherv of the edge lists and the replicated collapse, against the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import umi_collapse_rs_amd as umi
    from umi_collapse_rs_amd import synth
    from umi_collapse_rs_amd.sharded import split_dedup_device
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    st = synth.config2(seed=11, n_reads=60_000, umi_len=9)
    dev = torch.device("cuda:0")
    d_keys = torch.from_numpy(st["keys"].view(np.int64)).to(dev)
    d_freq = torch.from_numpy(st["freq"]).to(dev)
    d_kept = torch.zeros(len(st["keys"]), dtype=torch.uint8, device=dev)
    d_root = torch.zeros(len(st["keys"]), dtype=torch.int32, device=dev)
    ctx = umi.Context(0)
    stats = split_dedup_device(ctx, dist, d_keys, None, d_freq, st["bucket_off"], 9, d_kept, d_root,
                               k=1, edge_capacity=1024)  # small: exercises the grow-and-redo path
    torch.cuda.synchronize()
    q.put((rank, d_kept.cpu().numpy().tobytes(), d_root.cpu().numpy().tobytes(), stats["n_kept"]))
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


def test_two_ranks_split_one_giant_bucket():
    import torch.multiprocessing as mp
    import oracle as orc
    from umi_collapse_rs_amd import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    st = synth.config2(seed=11, n_reads=60_000, umi_len=9)
    okept, oroot, _ = orc.dedup_batch(st["keys"], None, st["freq"], st["bucket_off"], 9, 1)
    for rank, kept, root, n_kept in got:
        assert (np.frombuffer(kept, np.uint8) == okept).all()
        assert (np.frombuffer(root, np.int32).view(np.uint32) == oroot).all()
        assert n_kept == int(okept.sum())
