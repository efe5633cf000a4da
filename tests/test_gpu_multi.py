"""Multi-device context (umi_ctx_create_multi) on the one GPU of the test box: the device list
names it several times, so every worker has its own context, stream and workspace on the same
card.  Bucket sharding, the split of one giant bucket, the CLI's --devices, and the
one-process-per-GPU driver (ShardedDedup) on its default HIP path."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import bamio
import oracle as orc
from helpers import canonical, random_bucket
from test_gpu_parity import make_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mixed_batch(seed, L=12, n_frac=0.0):
    rng = np.random.default_rng(seed)
    keys, nm, fr, off = make_batch(rng, 300, L, 60, err=0.05, n_frac=n_frac)
    # a few larger buckets between them (chunk kernel, segment index)
    parts = [(keys, nm, fr, off)]
    for n_raw in (700, 5000, 1500):
        raw = rng.integers(0, 4, (n_raw, L))
        umis = sorted({"".join("ACGT"[c] for c in r) for r in raw})
        rng.shuffle(umis)
        freq = np.minimum(rng.geometric(0.5, len(umis)), 25).tolist()
        umis, freq, _ = canonical(umis, freq)
        k, m = orc.encode_keys(umis)
        parts.append((k, m, np.array(freq, np.int32), np.array([0, len(umis)], np.uint64)))
    keys = np.concatenate([p[0] for p in parts])
    nm = np.concatenate([p[1] for p in parts])
    fr = np.concatenate([p[2] for p in parts])
    offs, base = [np.zeros(1, np.uint64)], np.uint64(0)
    for p in parts:
        offs.append(p[3][1:] + base)
        base = base + p[3][-1]
    return keys, nm, fr, np.concatenate(offs)


@pytest.mark.parametrize("ids,k,p,algo,amf,n_frac", [
    ([0, 0], 1, 0.5, 0, 0, 0.0), ([0, 0, 0], 2, 1.0, 0, 0, 0.02), ([0, 0], 1, 0.5, 1, 2, 0.0), ([0], 1, 0.5, 0, 0, 0.0),
    ([0, 0, 0, 0, 0], 1, 0.5, 1, 0, 0.0)])
def test_bucket_sharded_call_equals_single_device_and_oracle(ids, k, p, algo, amf, n_frac):
    import umi_collapse_rs_amd as umi
    keys, nm, fr, off = _mixed_batch(900 + len(ids) + k, n_frac=n_frac)
    nmask = nm if nm.any() else None
    okept, oroot, _ = orc.dedup_batch(keys, nm, fr, off, 12, k, p, algo, amf)
    single = umi.Context(0)
    multi = umi.Context(ids)
    try:
        assert umi.load().umi_ctx_device_count(multi._h) == len(ids)
        skept, sroot, sst = single.dedup_batch(keys, nmask, fr, off, 12, k, p, algo, amf)
        mkept, mroot, mst = multi.dedup_batch(keys, nmask, fr, off, 12, k, p, algo, amf)
        assert (mkept == okept).all() and (mroot == oroot).all()
        assert (mkept == skept).all() and (mroot == sroot).all()
        for f in ("n_umis", "n_buckets", "max_bucket", "n_kept", "n_pairs", "n_edges"):
            assert mst[f] == sst[f], f
        # kept only (root NULL), and an option set on the multi context reaches every device
        multi.set_option("seg_index", 0)
        k2, r2, st2 = multi.dedup_batch(keys, nmask, fr, off, 12, k, p, algo, amf, want_root=False)
        assert r2 is None and (k2 == okept).all()
        if algo == 0:
            assert st2["n_pairs_evaluated"] > mst["n_pairs_evaluated"]
        # empty call, and the contract check of one device's share comes back as the call's error
        e0 = multi.dedup_batch(np.zeros(0, np.uint64), None, np.zeros(0, np.int32), [0], 12)
        assert len(e0[0]) == 0
        bad = fr.copy()
        bad[int(off[-2]) + 1] = 10 ** 6  # a rise inside the last bucket
        with pytest.raises(umi.UmiHipError) as e:
            multi.dedup_batch(keys, nmask, bad, off, 12, k, p, algo, amf)
        assert e.value.code == -3
    finally:
        single.close()
        multi.close()


@pytest.mark.parametrize("n_dev,algo,amf", [(2, 0, 0), (3, 0, 0), (2, 1, 1)])
def test_giant_bucket_is_split_over_the_devices(n_dev, algo, amf):
    """One bucket that dominates the call: its sub-bucket tasks are split over the devices, the
    edge lists gathered on the first one, one collapse."""
    import umi_collapse_rs_amd as umi
    from umi_collapse_rs_amd import synth
    st = synth.config2(seed=70 + n_dev, n_reads=70_000, umi_len=9)
    rng = np.random.default_rng(n_dev)
    sk, snm, sfr, soff = make_batch(rng, 20, 9, 40)
    keys = np.concatenate([sk, st["keys"]])
    fr = np.concatenate([sfr, st["freq"]])
    off = np.concatenate([soff, soff[-1:] + st["bucket_off"][1:]]).astype(np.uint64)
    okept, oroot, _ = orc.dedup_batch(keys, None, fr, off, 9, 1, 0.5, algo, amf)
    multi = umi.Context([0] * n_dev)
    try:
        multi.set_option("split_min", 10_000)
        kept, root, stm = multi.dedup_batch(keys, None, fr, off, 9, 1, 0.5, algo, amf)
        assert (kept == okept).all() and (root == oroot).all()
        assert stm["n_kept"] == int(okept.sum()) and stm["n_edges"] > 0
        multi.set_option("split_min", 10 ** 9)  # the same call bucket-sharded
        kept2, root2, _ = multi.dedup_batch(keys, None, fr, off, 9, 1, 0.5, algo, amf)
        assert (kept2 == okept).all() and (root2 == oroot).all()
    finally:
        multi.close()


def test_device_pointer_entry_points_want_one_device():
    import torch
    import umi_collapse_rs_amd as umi
    multi = umi.Context([0, 0])
    try:
        t = torch.zeros(8, dtype=torch.int64, device="cuda")
        with pytest.raises(umi.UmiHipError) as e:
            multi.dedup_batch_device(t.data_ptr(), 0, t.data_ptr(), [0, 8], 12, t.data_ptr())
        assert e.value.code == -1
        with pytest.raises(umi.UmiHipError):
            multi.pack_mask_device(t.data_ptr(), 8, t.data_ptr())
    finally:
        multi.close()


def test_pack_mask_device_matches_packbits():
    import torch
    import umi_collapse_rs_amd as umi
    ctx = umi.Context(0)
    try:
        rng = np.random.default_rng(3)
        for n in (1, 7, 8, 63, 64, 65, 1000, 4097, 100_003):
            kept = (rng.random(n) < 0.4).astype(np.uint8) * rng.integers(1, 255, n).astype(np.uint8)
            d_kept = torch.from_numpy(kept).cuda()
            d_bits = torch.full(((n + 7) // 8 + 3,), 0xAA, dtype=torch.uint8, device="cuda")
            ctx.pack_mask_device(d_kept.data_ptr(), n, d_bits.data_ptr())
            torch.cuda.synchronize()
            got = d_bits.cpu().numpy()
            assert (got[: (n + 7) // 8] == np.packbits(kept != 0, bitorder="little")).all(), n
            assert (got[(n + 7) // 8:] == 0xAA).all(), n  # nothing written past the mask
    finally:
        ctx.close()


def test_cli_devices_flag(tmp_path):
    cli = os.path.join(ROOT, "umi_collapse_rs_amd", "bin", "umicollapse")
    header, recs = bamio.synthetic_bam(21, 400, 60, umi_len=12, err=0.03)
    src = str(tmp_path / "in.bam")
    with open(src, "wb") as f:
        f.write(bamio.bgzf_compress(header + b"".join(recs)))
    outs = []
    for flags in (["--device", "0"], ["--devices", "0,0,0"]):
        dst = str(tmp_path / ("out%d.bam" % len(outs)))
        r = subprocess.run([cli, "-i", src, "-o", dst, "--merge", "avgqual", "--num-threads", "4"] + flags,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        outs.append(bamio.bgzf_decompress(open(dst, "rb").read()))
    assert outs[0] == outs[1] and len(outs[0]) > len(header)
    r = subprocess.run([cli, "-i", src, "-o", str(tmp_path / "x.bam"), "--devices", "0,,1"], capture_output=True, text=True)
    assert r.returncode != 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from umi_collapse_rs_amd.sharded import ShardedDedup
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    keys, nm, fr, off = _mixed_batch(4242)
    sd = ShardedDedup(dist, umi_len=12, k=1)  # default compute: the HIP path, shard resident on the GPU
    kept = sd.run(keys, None, fr, off)
    again = sd.run_resident()  # a second pass over the resident shard: same bits
    assert (np.unpackbits(again[rank].cpu().numpy(), bitorder="little")[: sd._shard["n_local"][rank]].sum()
            == kept[np.concatenate([np.arange(int(off[b]), int(off[b + 1])) for b in sd._shard["parts"][rank]] or [np.zeros(0, int)]).astype(int)].sum())
    q.put((rank, kept.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_one_process_per_gpu_driver_on_its_hip_path():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    keys, nm, fr, off = _mixed_batch(4242)
    expect = orc.dedup_batch(keys, None, fr, off, 12, 1)[0]
    for r in range(2):
        assert (np.frombuffer(got[r], np.uint8) == expect).all()


def _nccl_worker(port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from umi_collapse_rs_amd.sharded import ShardedDedup
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    keys, nm, fr, off = _mixed_batch(777)
    sd = ShardedDedup(dist, umi_len=12, k=1)
    kept = sd.run(keys, None, fr, off)  # dedup_batch_device + pack_mask_device + RCCL all_gather_into_tensor
    q.put(kept.tobytes())
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_branch_of_the_sharded_driver_with_one_rank():
    """The "nccl" (= RCCL) branch of ShardedDedup -- device-resident shard, mask packed to bits on the
    device, all_gather_into_tensor on device buffers -- with a communicator of ONE rank: all a 1-GPU
    box can form.  It shows that the collective is issued on buffers RCCL accepts and that the result
    comes back intact; what more ranks do over xGMI is the driver's 8-GPU run to observe (the
    2-rank rehearsal above uses gloo on the same code path up to the collective)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    keys, nm, fr, off = _mixed_batch(777)
    expect = orc.dedup_batch(keys, None, fr, off, 12, 1)[0]
    assert (np.frombuffer(got, np.uint8) == expect).all()


def test_bench_rehearsal_two_ranks_over_gloo_and_one_rank_over_rccl(tmp_path):
    """bench.py's N > 1 path end to end on this box: 2 ranks over gloo (config 4's share shrunk,
    config 5 and config 2 blocks behind it), and the same code with --force-collective over a
    one-rank RCCL communicator: the line parses and carries the blocks the driver's 8-GPU run will."""
    import json
    env = dict(os.environ, BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--steps", "3", "--warmup", "1", "--reads", "500000", "--also", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["baseline_config"] == "4" and line["scaling"] == "weak"
    assert line["configs"]["2"]["ms_per_step"] > 0 and line["value"] > 0
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--config", "4", "--reads", "500000", "--no-extras", "--force-collective"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["collective"] == "nccl" and line["value"] > 0


def _rccl_one_process_worker(q):
    """(its own process: RCCL stays out of the test runner)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import umi_collapse_rs_amd as umi
    keys, nm, fr, off = _mixed_batch(31337)
    dev = torch.device("cuda", 0)
    n = len(keys)
    slice_bytes = (n + 7) // 8 + 5  # (padding: the slices of a real job differ in length)
    t_keys = torch.from_numpy(keys.view(np.int64)).to(dev)
    t_fr = torch.from_numpy(fr).to(dev)
    t_kept = torch.zeros(n, dtype=torch.uint8, device=dev)
    t_root = torch.zeros(n, dtype=torch.int32, device=dev)
    t_bits = torch.full((slice_bytes,), 0xAA, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx = umi.Context([0])  # a multi-device context of one device: all RCCL can be given on this box
    try:
        st = ctx.dedup_batch_device_multi([dict(d_keys=t_keys.data_ptr(), d_freq=t_fr.data_ptr(), bucket_off=off,
                                                d_kept=t_kept.data_ptr(), d_root=t_root.data_ptr(),
                                                d_bits_all=t_bits.data_ptr())], 12, slice_bytes, k=1)
        st2 = ctx.dedup_batch_device_multi([dict(d_keys=t_keys.data_ptr(), d_freq=t_fr.data_ptr(), bucket_off=off,
                                                 d_kept=t_kept.data_ptr())], 12, slice_bytes, k=1, gather=False)
    finally:
        ctx.close()
    torch.cuda.synchronize()
    q.put((t_kept.cpu().numpy().tobytes(), t_root.cpu().numpy().tobytes(), t_bits.cpu().numpy().tobytes(),
           st["n_kept"], st2["n_kept"], slice_bytes))


def test_rccl_all_gather_behind_the_c_abi_with_one_device():
    """umi_dedup_batch_device_multi: per-device resident shards through the ordinary pipeline, masks
    packed to bits and all-gathered by RCCL inside the library (ncclCommInitAll + grouped
    ncclAllGather, librccl opened on first use).  The communicator here has ONE rank -- the box has
    one GPU and RCCL refuses a device named twice -- so what this shows is that the library's own
    collective path runs end to end and leaves the right bytes; N > 1 is the driver's to observe."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_process_worker, args=(q,))
    p.start()
    kept_b, root_b, bits_b, n_kept, n_kept2, slice_bytes = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    keys, nm, fr, off = _mixed_batch(31337)
    okept, oroot, _ = orc.dedup_batch(keys, None, fr, off, 12, 1)
    kept = np.frombuffer(kept_b, np.uint8)
    assert (kept == okept).all() and (np.frombuffer(root_b, np.uint32) == oroot).all()
    assert n_kept == n_kept2 == int(okept.sum())
    bits = np.frombuffer(bits_b, np.uint8)
    want = np.packbits(okept, bitorder="little")
    assert len(bits) == slice_bytes and (bits[:len(want)] == want).all() and (bits[len(want):] == 0).all()
