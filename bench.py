#!/usr/bin/env python3
"""Headline benchmark of the hot path (BASELINE.json): UMI-pair Hamming comparisons/s
on 12-bp UMIs, per-position all-pairs adjacency + directional collapse.

Workload at N=1 = BASELINE config 2: 1,000,000 synthetic reads, 12-bp UMIs, ONE
alignment position (uniform UMIs, ~9.7e5 unique -> W ~ 4.7e11 unordered pairs),
--data naive --algo dir -k 1 -p 0.5.  A step = one pass of the whole hot path
(filter keys, all-pairs, collapse, kept mask) over the batch, inputs resident in HBM.
For N>1 the job is N such positions (weak scaling): position buckets are sharded one
per rank, no data-path collective, and the kept mask is all-gathered over RCCL.

Prints ONE JSON line (rank 0).  `value` = pairs of W processed by all ranks / second."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X VALU peak in 32-bit integer lane-ops/s: 256 CU x 4 SIMD x 32 lanes x 2.4 GHz
# (MI355X_MICROARCH.md: SIMD-32, 2.4 GHz max clock; = half the 157.3 TFLOP/s FP32 FMA peak)
VALU_PEAK_TLANEOPS = 256 * 4 * 32 * 2.4e9 / 1e12
HBM_PEAK_GBS = 8000.0
TRAFFIC_CONFIG2 = (395077.5 + 5265.0 + 8881.2 + 1243.1) * 1024  # bytes per launch of bs_run_kernel +
# tab_scan_kernel at config 2: FETCH_SIZE + WRITE_SIZE (KB) of profiles/r01_config2_pmc_fetch_write_v12.csv
# wave instructions per bs_run_kernel launch at config 2 (profiles/r01_config2_sq_counters_v12.csv):
VALU_INSTS_CONFIG2 = 1.606e8
INSTS_CONFIG2 = 1.606e8 + 0.917e8 + 0.324e8 + 0.077e8 + 0.022e8  # + scalar, branch, LDS, VMEM
ISSUE_PEAK = 256 * 4 * 2.4e9  # one instruction per SIMD and clock
# Algorithmic VALU lane-ops per pair of the dominant kernel (bit-sliced filter): per column
# and 32-row group, 2 full-rate 32-bit ops per base for the unit mismatch masks plus the
# counter over the L'/unit units: 0.5 per unit for K = 0, K+1 per unit for K > 1, and for
# K = 1 the (any, two) tree of bs_pair_kernel (2 ops per triple of units, 2 per merge).
# L = 12, k = 1, unit = 2: (24 + 6) / 32 = 0.94 lane-ops per pair (kernel K1b)
def ops_per_pair(umi_len, k, unit=2):
    lp = 8 if umi_len <= 8 else 12 if umi_len <= 12 else 16 if umi_len <= 16 else 22
    units = lp // unit
    if k == 0:
        counter = 0.5 * units
    elif k == 1:
        full, rem = divmod(units, 3)
        groups = full + (1 if rem else 0)
        counter = 2 * full + (2 if rem == 2 else 0)   # or3 + majority / or + and
        counter += sum(2 if (g < groups - 1 or rem != 1) else 1 for g in range(1, groups))
        counter += max(0, groups - 2)                 # any_acc |= any between merges
    else:
        counter = (k + 1.0) * units
    return (2.0 * lp + counter) / 32.0
BYTES_PER_UMI = 16   # 8 B key + 4 B freq in, 4 B label out


def table_kernel_shape(n_max, umi_len, opts):
    """Mirror of choose_live_units (csrc/umihip_api.cpp): does the largest bucket go through the
    table variant (key-sorted, 32-bit keys, 2 live units)?  Returns (live, prefix_units, run)."""
    if umi_len > 16 or opts.get("bs_sorted", 1) == 0 or opts.get("bs_tables", 1) == 0 \
            or opts.get("bs_unit", 2) != 2 or opts.get("prune", 0) or opts.get("bitslice", 1) == 0:
        return None
    lp = 8 if umi_len <= 8 else 12 if umi_len <= 12 else 16
    units, pad = lp // 2, lp - umi_len
    if units <= 2 or n_max < 32768:
        return None
    bases = max(0, 2 * (units - 2) - pad)
    if (n_max >> (2 * bases)) < 4:
        return None
    return 2, units - 2, n_max / 4.0 ** bases


def table_ops_per_pair(k, shape):
    """Algorithmic lane-ops per pair of bs_tab_kernel: per column and 32-row group the
    register-indexed lookups (1 move for k = 1, `live` otherwise) and the merge (2 bitop3 for
    k = 1, (k+1) per live unit otherwise),
    plus the prefix state (4 mask ops per prefix unit and its counter ops) once per column run."""
    live, pu, run = shape
    merge = 2.0 if k == 1 else (0.5 * live if k == 0 else (k + 1.0) * live)
    if k == 1:
        live -= 1  # the second lookup is the indexed source of the majority op, not a move
    tree = {0: 0.5 * pu, 1: 2.0 * (pu // 3) + (pu % 3) + max(0, (pu + 2) // 3 - 1)}.get(k, (k + 1.0) * pu)
    return (live + merge + (4.0 * pu + tree) / max(run, 1.0)) / 32.0


def cpu_baseline(st, n_sample, k, p, umi_len=12, config=2):
    """The oracle (a scalar C port of the reference path) on a bounded sample of the same
    workload: config 2 -> n_sample unique UMIs drawn in rank order from the staged position;
    configs 3/5 -> a prefix of whole buckets."""
    import oracle as orc
    n = len(st["keys"])
    if config == 2:
        rng = np.random.default_rng(12345)
        idx = np.sort(rng.choice(n, size=min(n_sample, n), replace=False))
        keys, freq = st["keys"][idx], st["freq"][idx]
        boff = np.array([0, len(idx)], np.uint64)
        what = "%d unique UMIs sampled in rank order from the same position" % len(idx)
    else:
        nb = min(len(st["bucket_off"]) - 1, 50_000)
        boff = st["bucket_off"][: nb + 1]
        m = int(boff[-1])
        keys, freq = st["keys"][:m], st["freq"][:m]
        what = "the first %d buckets (%d unique UMIs)" % (nb, m)
    sz = np.diff(boff.astype(np.int64))
    w = int((sz * (sz - 1) // 2).sum())
    t0 = time.perf_counter()
    kept, _, calls = orc.dedup_batch(keys, None, freq, boff, umi_len, k, p)
    dt = time.perf_counter() - t0
    out = {"value": w / dt, "unit": "UMI-pair comparisons/s", "cores": 1, "kind": "port",
           "sample": "%s (W=%d pairs, %d umi_dist calls, %.1f s); the Rust reference cannot be "
                     "built here (no rustc)" % (what, w, calls, dt),
           "dist_calls_per_s": calls / dt, "umis_per_s": len(keys) / dt}
    # The machine's CPU ceiling for a bucket-parallel host (the reference itself is single
    # threaded, deduplicate_sam.rs:207): the same sample once per core, concurrently (ctypes
    # drops the GIL inside the oracle call).  Reported beside `value`, never instead of it.
    import os
    from concurrent.futures import ThreadPoolExecutor
    cores = max(1, min(len(os.sched_getaffinity(0)), 16))  # the box's CPU share for one GPU
    if cores > 1:
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            list(ex.map(lambda _: orc.dedup_batch(keys, None, freq, boff, umi_len, k, p), range(cores)))
        dta = time.perf_counter() - t0
        out["all_cores"] = {"value": cores * w / dta, "cores": cores,
                            "how": "one copy of the sample per core, concurrently (%.1f s)" % dta}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per position")
    ap.add_argument("--umi-len", type=int, default=12)
    ap.add_argument("-k", type=int, default=1)
    ap.add_argument("-p", type=float, default=0.5)
    ap.add_argument("--cpu-sample", type=int, default=80_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed block (profiling runs): no host-buffer call, no CPU baseline")
    ap.add_argument("--opt", action="append", default=[], help="ctx option name=value (tuning)")
    ap.add_argument("--split", action="store_true",
                    help="N>1 only: strong scaling -- ONE giant position, its tile tasks split over "
                         "the ranks, edge lists all-gathered, collapse replicated (default for N>1 "
                         "is weak scaling: one position per rank)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE config per GPU: 2 = one giant position (headline), 3 = 10M reads "
                         "in 100k positions, 4 = one GPU's share of the 8-GPU config (12.5M reads in "
                         "125k positions), 5 = 20-bp UMIs k=2 in many positions (parity-test shapes; "
                         "the judged bench line is config 2)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # one process per GPU; the modulo only matters when the launch is rehearsed with more
    # ranks than GPUs (BENCH_BACKEND=gloo on a 1-GPU box)
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_BACKEND", "nccl")  # nccl = RCCL over xGMI
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import umi_collapse_rs_amd as umi
    from umi_collapse_rs_amd import synth
    from umi_collapse_rs_amd.sharded import partition_buckets

    # ---- workload: position `rank` of the N-position job (config 2 per GPU)
    if args.config == 2:
        st = synth.config2(seed=2 + (0 if args.split else 1000 * rank), n_reads=args.reads,
                           umi_len=args.umi_len)
        workload = ("BASELINE config 2 per GPU: %d reads, %d-bp UMIs, one alignment position "
                    "(uniform UMIs)" % (args.reads, args.umi_len))
    elif args.config in (3, 4):
        per_gpu = 10_000_000 if args.config == 3 else 12_500_000  # config 4: 100M reads over 8 GPUs
        args.reads = per_gpu if args.reads == 1_000_000 else args.reads
        st = synth.config3(seed=args.config + 1000 * rank, n_reads=args.reads,
                           n_positions=args.reads // 100, umi_len=args.umi_len)
        workload = ("BASELINE config %d per GPU: %d reads, %d-bp UMIs, %d alignment positions "
                    "(molecule model)" % (args.config, args.reads, args.umi_len, args.reads // 100))
    else:
        args.umi_len, args.k = 20, 2
        args.reads = 6_250_000 if args.reads == 1_000_000 else args.reads
        st = synth.config3(seed=5 + 1000 * rank, n_reads=args.reads,
                           n_positions=args.reads // 100, umi_len=20)
        workload = ("BASELINE config 5 per GPU: %d reads, 20-bp UMIs, %d alignment positions, "
                    "k=2 (molecule model)" % (args.reads, args.reads // 100))
    n = len(st["keys"])
    nb_sizes = np.diff(st["bucket_off"].astype(np.int64))
    w_local = int((nb_sizes * (nb_sizes - 1) // 2).sum())
    sizes = [n]
    if world > 1:
        t = torch.tensor([n], dtype=torch.int64, device=dev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        sizes = [int(x.item()) for x in allt]
        parts = partition_buckets(sizes, world)
        assert sorted(int(p[0]) for p in parts) == list(range(world))
    w_total = w_local * world if args.config != 2 else sum(s * (s - 1) // 2 for s in sizes)
    reads_total = args.reads * world
    split = args.split and world > 1
    if split:  # every rank holds the same bucket: the job is ONE position
        w_total, reads_total = w_local, args.reads

    d_keys = torch.from_numpy(st["keys"].view(np.int64)).to(dev)
    d_freq = torch.from_numpy(st["freq"]).to(dev)
    d_kept = torch.zeros(n, dtype=torch.uint8, device=dev)
    boff = st["bucket_off"]
    max_n = max(sizes)
    gather_in = torch.zeros(max_n, dtype=torch.uint8, device=dev)
    gather_out = torch.zeros(max_n * world, dtype=torch.uint8, device=dev) if world > 1 else None

    ctx = umi.Context(dev_index, profile=True)
    for o in args.opt:
        name, val = o.split("=")
        ctx.set_option(name, int(val))
    stream = torch.cuda.current_stream().cuda_stream

    from umi_collapse_rs_amd.sharded import split_dedup_device

    def step():
        if split:
            s = split_dedup_device(ctx, dist, d_keys, None, d_freq, boff, args.umi_len, d_kept,
                                   k=args.k, percentage=args.p)
            for f in ("ms_prep", "ms_collapse", "ms_finalize"):
                s.setdefault(f, 0.0)
            s.setdefault("n_candidates", 0)
            return s
        s = ctx.dedup_batch_device(d_keys.data_ptr(), 0, d_freq.data_ptr(), boff, args.umi_len,
                                   d_kept.data_ptr(), 0, k=args.k, percentage=args.p,
                                   stream=stream)
        if world > 1:  # all-gatherv of the kept mask (padded all_gather over RCCL/xGMI)
            gather_in[:n].copy_(d_kept)
            dist.all_gather_into_tensor(gather_out, gather_in)
        return s

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    kept_n = int(d_kept.sum().item())
    assert kept_n == stats[-1]["n_kept"]

    # the same pass through the host-buffer entry point (H2D of keys/freq + D2H of the mask
    # inside the call): the PCIe-inclusive rate, reported beside `value`, never as it
    host_ms = None
    if rank == 0 and world == 1 and not args.no_extras:
        ctx.dedup_batch(st["keys"], None, st["freq"], boff, args.umi_len, k=args.k,
                        percentage=args.p, want_root=False)
        t1 = time.perf_counter()
        hk, _, _ = ctx.dedup_batch(st["keys"], None, st["freq"], boff, args.umi_len, k=args.k,
                                   percentage=args.p, want_root=False)
        host_ms = (time.perf_counter() - t1) * 1e3
        assert int(hk.sum()) == kept_n

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        pair_ms = float(np.mean([s["ms_pairs"] for s in stats]))
        coll_ms = float(np.mean([s["ms_collapse"] for s in stats]))
        s0 = stats[-1]
        opts = {o.split("=")[0]: int(o.split("=")[1]) for o in args.opt}
        n_max = int(np.diff(st["bucket_off"].astype(np.int64)).max())
        shape = table_kernel_shape(n_max, args.umi_len, opts) if args.k <= 3 and not split else None
        walked = None
        std_cfg2 = (args.config == 2 and args.reads == 1_000_000 and args.umi_len == 12 and args.k == 1
                    and not opts)
        if shape:
            # the item walk covers only the (row tile, column tile) pairs its scan keeps (those
            # whose high bases leave a row within k): the fraction is a property of the data and
            # of the algorithm, counted by the run itself
            walked = min(1.0, s0["n_pairs_evaluated"] / max(w_local, 1))
            if opts.get("bs_transposed", 1):
                # bs_run_kernel: its work is per column run and open row lane, not per pair; the
                # lane-ops are the VALU instructions its launch executes (SQ counter pass of the
                # same command in profiles/, x 64 lanes) -- an upper bound of the algorithmic ones
                opp = VALU_INSTS_CONFIG2 * 64 / w_local if std_cfg2 else None
                kernel_name = ("tab_scan_kernel + bs_run_kernel (bit-sliced filter on key-sorted columns: "
                               "early out on the high bases, columns of a run across the lanes)")
            else:
                opp = table_ops_per_pair(args.k, shape) * walked
                kernel_name = ("tab_scan_kernel + bs_tab_kernel (bit-sliced filter, key-sorted columns, "
                               "register tables, early out on the high bases)")
        else:
            opp = ops_per_pair(args.umi_len, args.k, opts.get("bs_unit", 2))
            kernel_name = "bs_pair_kernel (bit-sliced all-pairs filter)"
        # --split: each rank's pair kernels cover 1/world of W
        achieved = None if opp is None else opp * (w_local / world if split else w_local) / (max(pair_ms, 1e-6) * 1e-3) / 1e12
        out = {
            "metric": "UMI-pair Hamming comparisons/s (12-bp UMIs, all-pairs adjacency + "
                      "directional collapse)",
            "value": w_total * args.steps / dt,
            "unit": "UMI-pair comparisons/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong" if split else "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "%s, --data naive --algo dir -k %d -p %g" % (
                           workload, args.k, args.p),
                       "reads_per_position": args.reads, "positions": world,
                       "unique_umis_rank0": n, "pairs_W_total": w_total,
                       "parallelism": ("tile-task split x%d + edge all-gatherv" % world) if split
                       else "bucket-sharded x%d" % world},
            "reads_per_s": reads_total * args.steps / dt,
            "host_buffer_path": None if host_ms is None else {
                "ms_per_call": host_ms, "pairs_per_s": w_local / (host_ms * 1e-3),
                "note": "umi_dedup_batch with pageable host arrays: PCIe copies included"},
            "kept_rank0": kept_n,
            "roofline": {
                "bound": "valu", "kernel": kernel_name,
                "achieved": achieved, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlaneop/s",
                "frac": None if achieved is None else achieved / VALU_PEAK_TLANEOPS,
                # fabric-side bytes of one pair-kernel launch at config 2 from the PMC passes in
                # profiles/ ((FETCH_SIZE+WRITE_SIZE)*1024, uncorrected: 4 B/lane accesses, see
                # profiles/README.md); other shapes: null
                "traffic": TRAFFIC_CONFIG2 if (shape and std_cfg2) else None,
                "ops_per_pair": opp, "pairs_per_launch": w_local,
                "walked_fraction": walked,
                "ops_per_walked_pair": (opp / walked) if (walked and opp is not None) else None,
                "kernel_ms": pair_ms,
                "note": "integer VALU roofline (0 algorithmic HBM bytes per pair; no MFMA).  "
                        "achieved = lane-ops of one launch / time of the pair kernels of one step "
                        "(HIP events); peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz.  The kernel "
                        "decides a pair from its high bases where they already differ in more than "
                        "k units (whole column tiles and column runs at a time) and walks the rest "
                        "(walked_fraction of W, counted by the run) with the columns of a run "
                        "across the lanes; its lane-ops are the VALU instructions of the launch "
                        "(SQ counters, profiles/) x 64.  It is bound by dependent-instruction "
                        "latency and instruction issue at 5 waves per SIMD, not by VALU throughput "
                        "(DESIGN.md section 7); frac fell from version to version while pairs/s "
                        "rose.  HBM view in roofline_hbm."},
            # the resource the table kernel is actually bound by: instructions issued per SIMD
            # (count from the SQ counter pass in profiles/, time from this run's HIP events)
            "roofline_issue": None if not (shape and std_cfg2) else {
                "bound": "instruction issue", "achieved": INSTS_CONFIG2 / (max(pair_ms, 1e-6) * 1e-3) / 1e12,
                "peak": ISSUE_PEAK / 1e12, "unit": "T wave-instructions/s",
                "frac": INSTS_CONFIG2 / (max(pair_ms, 1e-6) * 1e-3) / ISSUE_PEAK,
                "insts_per_launch": INSTS_CONFIG2},
            "roofline_hbm": {
                "bound": "hbm", "achieved": BYTES_PER_UMI * n / (ms_step * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": BYTES_PER_UMI * n / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "bytes_per_umi": BYTES_PER_UMI},
            "phases_ms": {"prep": float(np.mean([s["ms_prep"] for s in stats])),
                          "pairs": pair_ms, "collapse": coll_ms,
                          "finalize": float(np.mean([s["ms_finalize"] for s in stats]))},
            "counters": {"n_edges": s0["n_edges"], "n_candidates": s0["n_candidates"],
                         "n_rounds": s0["n_rounds"], "pairs_evaluated": s0["n_pairs_evaluated"]},
        }
        if world == 1 and not args.no_cpu_baseline and not args.no_extras:
            out["cpu_baseline"] = cpu_baseline(st, args.cpu_sample, args.k, args.p, args.umi_len,
                                               args.config)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
