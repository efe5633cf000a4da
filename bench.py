#!/usr/bin/env python3
"""Headline benchmark of the hot path (BASELINE.json): UMI-pair Hamming comparisons/s
on 12-bp UMIs, per-position all-pairs adjacency + directional collapse.

Workload at N=1 = BASELINE config 2: 1,000,000 synthetic reads, 12-bp UMIs, ONE
alignment position (uniform UMIs, ~9.7e5 unique -> W ~ 4.7e11 unordered pairs),
--data naive --algo dir -k 1 -p 0.5.  A step = one pass of the whole hot path
(filter keys, pair evaluation, collapse, kept mask) over the batch, inputs resident in HBM.
For N>1 the job is N such positions (weak scaling): position buckets are sharded one
per rank, no data-path collective, and the kept mask is all-gathered over RCCL.

`value` is an EFFECTIVE rate: W / t, the pairs of the position per second of hot-path time.
The default path does not look at every pair: a large position is cut into n-gram
sub-buckets (two UMIs within k substitutions agree on one of k+1 base ranges) and only the
pairs inside them are compared -- `walked_fraction` of W, `value_executed` per second; the
result is the one the all-pairs kernels give (`value_bruteforce`: the same step with the
partition and the key sort switched off, every pair evaluated), bit for bit.

Prints ONE JSON line (rank 0)."""
import argparse
import glob
import hashlib
import json
import os
import subprocess
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
BYTES_PER_UMI = 16   # 8 B key + 4 B freq in, 4 B label out
PROFILE_JSON = os.path.join(ROOT, "profiles", "r02_config2_counters.json")


def source_sha256():
    """sha256 over the kernel and host sources of the library (sorted file names, contents)."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "umi_collapse_rs_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def load_profile():
    """The rocprofv3 numbers of the config-2 step (tools/profile_r02.sh -> tools/summarize_r02.py):
    per kernel the average duration, the SQ instruction counts and the fabric-side bytes, stamped
    with the sha256 of the sources they were measured on."""
    if not os.path.exists(PROFILE_JSON):
        return None, "no profiles/r02_config2_counters.json"
    prof = json.load(open(PROFILE_JSON))
    if prof.get("source_sha256") != source_sha256():
        return prof, "profile measured on other sources (sha256 %s...)" % prof.get("source_sha256", "?")[:12]
    return prof, None


def cpu_baseline(st, n_sample, k, p, umi_len=12, one_position=True):
    """The oracle (a scalar C port of the reference path) on a bounded sample of the same
    workload: one deep position -> n_sample unique UMIs drawn in rank order from it;
    many positions -> a prefix of whole buckets."""
    import oracle as orc
    n = len(st["keys"])
    if one_position:
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


def end_to_end(n_reads, n_positions, threads):
    """bin/umicollapse on a generated BAM of this box: whole-file reads/s and its split
    (the program prints it with --timing)."""
    cli = os.path.join(ROOT, "umi_collapse_rs_amd", "bin", "umicollapse")
    gen = os.path.join(ROOT, "tools", "make_bam.py")
    if not (os.path.exists(cli) and os.path.exists(gen)):
        return None
    tmp = os.environ.get("TMPDIR", "/tmp")
    src, dst = os.path.join(tmp, "bench_e2e_in.bam"), os.path.join(tmp, "bench_e2e_out.bam")
    try:
        t0 = time.perf_counter()
        subprocess.check_call([sys.executable, gen, src, "--reads", str(n_reads), "--positions", str(n_positions)],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
        t_gen = time.perf_counter() - t0
        best = None
        for _ in range(2):  # the second run has the file cache and the GPU context warm
            t0 = time.perf_counter()
            r = subprocess.run([cli, "-i", src, "-o", dst, "--merge", "avgqual", "--num-threads",
                                str(threads)], capture_output=True, text=True, timeout=600)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": r.stderr[-300:]}
            split = {}
            for line in r.stderr.splitlines():
                if line.startswith("phases:"):  # the program's own split of its wall time
                    import re
                    for name, val in re.findall(r"([a-z+ ()A-Z0-9]+?) ([0-9.]+) s", line[7:]):
                        split[name.strip(" ,")] = float(val)
            if best is None or dt < best[0]:
                best = (dt, split)
        return {"reads": n_reads, "positions": n_positions, "bam_bytes": os.path.getsize(src),
                "wall_s": best[0], "reads_per_s": n_reads / best[0], "split_s": best[1],
                "threads": threads, "generate_s": t_gen,
                "note": "bin/umicollapse --merge avgqual, process start to exit (HIP context "
                        "creation included), BAM generated on this box by tools/make_bam.py"}
    except Exception as e:  # the hot-path numbers stand without it
        return {"error": repr(e)[:300]}
    finally:
        for f in (src, dst):
            if os.path.exists(f):
                os.unlink(f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per position")
    ap.add_argument("--umi-len", type=int, default=12)
    ap.add_argument("-k", type=int, default=1)
    ap.add_argument("-p", type=float, default=0.5)
    ap.add_argument("--cpu-sample", type=int, default=80_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed block (profiling runs): no sustained block, no brute-force "
                         "step, no host-buffer call, no CPU baseline, no end-to-end run")
    ap.add_argument("--opt", action="append", default=[], help="ctx option name=value (tuning)")
    ap.add_argument("--split", action="store_true",
                    help="N>1 only: strong scaling -- ONE giant position, its tile tasks split over "
                         "the ranks, edge lists all-gathered, collapse replicated (default for N>1 "
                         "is weak scaling: one position per rank)")
    ap.add_argument("--config", default="2", choices=["2", "2m", "3", "4", "5"],
                    help="BASELINE config per GPU: 2 = one giant position of uniform UMIs (headline), "
                         "2m = one deep position from the molecule model (1M reads of ~100k molecules, "
                         "error 0.01 per base), 3 = 10M reads in 100k positions, 4 = one GPU's share of "
                         "the 8-GPU config (12.5M reads in 125k positions), 5 = 20-bp UMIs k=2 in many "
                         "positions (parity-test shapes; the judged bench line is config 2)")
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

    # ---- workload: position `rank` of the N-position job
    cfg = args.config
    if cfg == "2":
        st = synth.config2(seed=2 + (0 if args.split else 1000 * rank), n_reads=args.reads,
                           umi_len=args.umi_len)
        workload = ("BASELINE config 2 per GPU: %d reads, %d-bp UMIs, one alignment position "
                    "(uniform UMIs)" % (args.reads, args.umi_len))
    elif cfg == "2m":
        st = synth.config2m(seed=22 + 1000 * rank, n_reads=args.reads, umi_len=args.umi_len)
        workload = ("one deep alignment position from the molecule model: %d reads of %d molecules, "
                    "%d-bp UMIs, error 0.01 per base" % (args.reads, st["n_molecules"], args.umi_len))
    elif cfg in ("3", "4"):
        per_gpu = 10_000_000 if cfg == "3" else 12_500_000  # config 4: 100M reads over 8 GPUs
        args.reads = per_gpu if args.reads == 1_000_000 else args.reads
        st = synth.config3(seed=int(cfg) + 1000 * rank, n_reads=args.reads,
                           n_positions=args.reads // 100, umi_len=args.umi_len)
        workload = ("BASELINE config %s per GPU: %d reads, %d-bp UMIs, %d alignment positions "
                    "(molecule model)" % (cfg, args.reads, args.umi_len, args.reads // 100))
    else:
        args.umi_len, args.k = 20, 2
        args.reads = 6_250_000 if args.reads == 1_000_000 else args.reads
        st = synth.config3(seed=5 + 1000 * rank, n_reads=args.reads,
                           n_positions=args.reads // 100, umi_len=20)
        workload = ("BASELINE config 5 per GPU: %d reads, 20-bp UMIs, %d alignment positions, "
                    "k=2 (molecule model)" % (args.reads, args.reads // 100))
    one_position = cfg in ("2", "2m")
    n = len(st["keys"])
    nb_sizes = np.diff(st["bucket_off"].astype(np.int64))
    w_local = int((nb_sizes * (nb_sizes - 1) // 2).sum())
    w_all = [w_local]
    sizes = [n]
    if world > 1:
        t = torch.tensor([n, w_local], dtype=torch.int64, device=dev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        sizes = [int(x[0].item()) for x in allt]
        w_all = [int(x[1].item()) for x in allt]
    w_total = sum(w_all)
    reads_total = args.reads * world
    split = args.split and world > 1
    if split:  # every rank holds the same bucket: the job is ONE position
        w_total, reads_total = w_local, args.reads

    d_keys = torch.from_numpy(st["keys"].view(np.int64)).to(dev)
    d_freq = torch.from_numpy(st["freq"]).to(dev)
    d_kept = torch.zeros(n, dtype=torch.uint8, device=dev)
    boff = st["bucket_off"]
    d_boff = torch.from_numpy(np.ascontiguousarray(boff).view(np.int64)).to(dev)  # the table is an input too
    max_n = max(sizes)
    # the kept mask travels as bits: ceil(n / 8) bytes per rank, padded to the largest slice
    # (two sets: the all-gather of one step runs on the collective's own stream while the next step's
    # kernels run on this one; a set is reused two steps later, after its collective has been waited for)
    gather_in = [torch.zeros((max_n + 7) // 8, dtype=torch.uint8, device=dev) for _ in range(2)]
    gather_out = [torch.zeros(gather_in[0].numel() * world, dtype=torch.uint8, device=dev) if world > 1 else None
                  for _ in range(2)]
    gather_work = [None, None]
    step_no = [0]

    ctx = umi.Context(dev_index, profile=True)
    opts = {}
    for o in args.opt:
        name, val = o.split("=")
        ctx.set_option(name, int(val))
        opts[name] = int(val)
    stream = torch.cuda.current_stream().cuda_stream

    from umi_collapse_rs_amd.sharded import split_dedup_device

    def step(c=ctx):
        if split:
            s = split_dedup_device(c, dist, d_keys, None, d_freq, boff, args.umi_len, d_kept,
                                   k=args.k, percentage=args.p)
            for f in ("ms_prep", "ms_collapse", "ms_finalize"):
                s.setdefault(f, 0.0)
            s.setdefault("n_candidates", 0)
            return s
        s = c.dedup_batch_device(d_keys.data_ptr(), 0, d_freq.data_ptr(), boff, args.umi_len,
                                 d_kept.data_ptr(), 0, k=args.k, percentage=args.p,
                                 stream=stream, d_bucket_off=d_boff.data_ptr())
        if world > 1:  # all-gatherv of the kept mask: packed to bits on the device, padded all_gather
            slot = step_no[0] & 1
            step_no[0] += 1
            if gather_work[slot] is not None:
                gather_work[slot].wait()
            c.pack_mask_device(d_kept.data_ptr(), n, gather_in[slot].data_ptr(), stream=stream)  # over RCCL/xGMI
            gather_work[slot] = dist.all_gather_into_tensor(gather_out[slot], gather_in[slot], async_op=True)
        return s

    def drain_gathers():
        for i in range(2):
            if gather_work[i] is not None:
                gather_work[i].wait()
                gather_work[i] = None

    def timed(n_steps, c=ctx):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ss = [step(c) for _ in range(n_steps)]
        drain_gathers()  # every step's mask has arrived everywhere inside the timed region
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, ss

    for _ in range(args.warmup):
        step()
    drain_gathers()
    dt, stats = timed(args.steps)
    kept_n = int(d_kept.sum().item())
    assert kept_n == stats[-1]["n_kept"]
    if world > 1 and not split:  # this rank's slice of the gathered bit mask is its own kept mask
        slot = (step_no[0] - 1) & 1
        width = gather_in[0].numel()
        mine = gather_out[slot][rank * width:(rank + 1) * width]
        bits = ((mine[:, None] >> torch.arange(8, device=dev, dtype=torch.uint8)[None, :]) & 1).sum()
        assert int(bits.item()) == kept_n, "gathered mask differs from the local one"
    kept_ref = d_kept.clone()

    extras = not args.no_extras
    # a second block of at least one second: clocks at steady state
    sustained = None
    if extras:
        n_sus = max(args.steps, int(1.05 / max(dt / args.steps, 1e-6)) + 1)
        dts, _ = timed(n_sus)
        sustained = {"steps": n_sus, "seconds": dts, "ms_per_step": dts / n_sus * 1e3}

    # the same step with every pair of the position evaluated: the bit-sliced all-pairs mask kernel
    # on the unsorted bucket (no n-gram partition, no key sort, no early out)
    brute = None
    if extras and world == 1 and one_position and n <= 1_200_000:
        cb = umi.Context(dev_index, profile=True)
        try:
            for name, v in (("seg_index", 0), ("bs_sorted", 0)):
                cb.set_option(name, v)
            step(cb)
            nb_steps = 3
            dtb, sb = timed(nb_steps, cb)
            assert bool((d_kept == kept_ref).all().item()), "brute-force kept mask differs"
            brute = {"value": w_local * nb_steps / dtb, "ms_per_step": dtb / nb_steps * 1e3,
                     "steps": nb_steps, "pairs_evaluated": sb[-1]["n_pairs_evaluated"],
                     "kernel": "bs_pair_kernel (bit-sliced filter, every pair of the position; "
                               "options seg_index=0 bs_sorted=0); kept mask equal to the default path's"}
        finally:
            cb.close()

    # the same pass through the host-buffer entry point (H2D of keys/freq + D2H of the mask
    # inside the call): the PCIe-inclusive rate, reported beside `value`, never as it
    host_ms = None
    if rank == 0 and world == 1 and extras:
        ctx.dedup_batch(st["keys"], None, st["freq"], boff, args.umi_len, k=args.k,
                        percentage=args.p, want_root=False)
        t1 = time.perf_counter()
        hk, _, _ = ctx.dedup_batch(st["keys"], None, st["freq"], boff, args.umi_len, k=args.k,
                                   percentage=args.p, want_root=False)
        host_ms = (time.perf_counter() - t1) * 1e3
        assert int(hk.sum()) == kept_n

    # read staging on the device (N2): the same 1,000,000 reads as UMI text + alignment key, resident
    # in HBM, through umi_stage_reads_device; its output must be the arrays the timed steps ran on
    staging = None
    if rank == 0 and world == 1 and extras and cfg == "2":
        b2 = synth.uniform_reads(2, args.reads, args.umi_len)
        d_umi = torch.from_numpy(synth.BASES[b2].reshape(-1).copy()).to(dev)
        d_akey = torch.zeros(args.reads, dtype=torch.int64, device=dev)
        o_keys = torch.zeros(args.reads, dtype=torch.int64, device=dev)
        o_freq = torch.zeros(args.reads, dtype=torch.int32, device=dev)
        o_rep = torch.zeros(args.reads, dtype=torch.int64, device=dev)
        o_off = torch.zeros(args.reads + 1, dtype=torch.int64, device=dev)

        def stage_once():
            return ctx.stage_reads_device(d_akey.data_ptr(), d_umi.data_ptr(), 0, args.reads, args.umi_len,
                                          o_keys.data_ptr(), 0, o_freq.data_ptr(), o_rep.data_ptr(),
                                          o_off.data_ptr(), merge=0, align_key_bits=1, stream=stream)
        ne, nbk = stage_once()
        assert ne == n and nbk == 1 and bool((o_keys[:n] == d_keys).all().item()) and bool((o_freq[:n] == d_freq).all().item())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_st = 10
        for _ in range(n_st):
            stage_once()
        torch.cuda.synchronize()
        ms_stage = (time.perf_counter() - t1) / n_st * 1e3
        staging = {"ms_per_call": ms_stage, "reads_per_s": args.reads / (ms_stage * 1e-3),
                   "reads_per_s_with_hot_path": args.reads / ((ms_stage + dt / args.steps * 1e3) * 1e-3),
                   "note": "umi_stage_reads_device on the same reads (12-byte UMI text + alignment key per read, "
                           "resident in HBM): encode, sort by (position, UMI, file index), merge equal UMIs, rank "
                           "order; output equal to the arrays of the timed steps (checked)"}

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        mean = lambda f: float(np.mean([s[f] for s in stats]))
        pair_ms, coll_ms, prep_ms = mean("ms_pairs"), mean("ms_collapse"), mean("ms_prep")
        s0 = stats[-1]
        walked = min(1.0, s0["n_pairs_evaluated"] / max(w_local, 1)) if w_local else None
        std_cfg2 = (cfg == "2" and args.reads == 1_000_000 and args.umi_len == 12 and args.k == 1
                    and not opts and not split)
        # ---- roofline of the dominant kernel of the step, from the stamped profile
        prof, stale_why = load_profile() if std_cfg2 else (None, "not the profiled workload")
        roofline = {"bound": "valu", "kernel": None, "achieved": None, "peak": VALU_PEAK_TLANEOPS,
                    "unit": "Tlaneop/s", "frac": None, "traffic": None,
                    "stale": bool(stale_why), "source": os.path.relpath(PROFILE_JSON, ROOT)}
        kernels_view = None
        if stale_why:
            roofline["why_null"] = stale_why
        if prof and not stale_why:
            ks = prof["kernels"]
            dom = max(ks, key=lambda k_: ks[k_]["avg_us"] * ks[k_]["calls_per_step"])
            kd = ks[dom]
            # phase of the step the kernel runs in: its live time comes from this run's HIP events
            phase_ms = {"prep": prep_ms, "pairs": pair_ms, "collapse": coll_ms}[kd["phase"]]
            phase_prof_us = sum(v["avg_us"] * v["calls_per_step"] for v in ks.values() if v["phase"] == kd["phase"])
            live_us = phase_ms * 1e3 * (kd["avg_us"] * kd["calls_per_step"] / max(phase_prof_us, 1e-9))
            valu = kd.get("SQ_INSTS_VALU")
            achieved = None if valu is None else valu * 64 / (live_us * 1e-6) / 1e12
            traffic = None
            if kd.get("FETCH_SIZE_KB") is not None and kd.get("WRITE_SIZE_KB") is not None:
                traffic = (kd["FETCH_SIZE_KB"] + kd["WRITE_SIZE_KB"]) * 1024
            roofline.update({
                "kernel": dom, "achieved": achieved,
                "frac": None if achieved is None else achieved / VALU_PEAK_TLANEOPS,
                "traffic": traffic, "kernel_us_live": live_us, "kernel_us_rocprof": kd["avg_us"],
                "valu_insts_per_launch": valu, "phase": kd["phase"],
                "note": "executed-instruction view of the kernel that takes the largest share of the "
                        "step: achieved = SQ_INSTS_VALU of one launch (rocprofv3 --pmc, profiles/) x 64 "
                        "lanes / the kernel's time in THIS run (its phase's HIP-event time x its share "
                        "of the phase under rocprofv3); peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz.  "
                        "Integer/bitwise work with 0 algorithmic HBM bytes per pair; traffic = "
                        "(FETCH_SIZE + WRITE_SIZE) x 1024 of one launch, uncorrected (4-16 B per lane "
                        "scattered accesses: the guide's x2 is calibrated for 16 B/lane streams)."})
            kernels_view = {k_: {"us": v["avg_us"], "per_step": v["calls_per_step"], "phase": v["phase"]}
                            for k_, v in sorted(ks.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls_per_step"])[:8]}
        metric = ("UMI-pair Hamming comparisons/s, effective (W pairs of the position / hot-path time; "
                  "12-bp UMIs, all-pairs adjacency + directional collapse; the n-gram partition compares "
                  "walked_fraction of W, the all-pairs kernels' result bit for bit)")
        out = {
            "metric": metric,
            "value": w_total * args.steps / dt,
            "unit": "UMI-pair comparisons/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "strong" if split else "weak",
            "vs_baseline": None, "dtype": "u32" if args.umi_len <= 16 else "u64", "data": "synthetic",
            "config": {"workload": "%s, --data naive --algo dir -k %d -p %g" % (workload, args.k, args.p),
                       "reads_per_position": args.reads, "positions": world if one_position else len(nb_sizes) * world,
                       "unique_umis_rank0": n, "pairs_W_total": w_total,
                       "parallelism": ("tile-task split x%d + edge all-gatherv" % world) if split
                       else "bucket-sharded x%d" % world},
            "value_effective": w_total * args.steps / dt,
            "value_executed": s0["n_pairs_evaluated"] * world * args.steps / dt,
            "walked_fraction": walked,
            "value_bruteforce": brute,
            "sustained": sustained,
            "sustained_ms_per_step": None if sustained is None else sustained["ms_per_step"],
            "reads_per_s": reads_total * args.steps / dt,
            "staging_device": staging,
            "host_buffer_path": None if host_ms is None else {
                "ms_per_call": host_ms, "pairs_per_s": w_local / (host_ms * 1e-3),
                "note": "umi_dedup_batch with pageable host arrays: PCIe copies included"},
            "kept_rank0": kept_n,
            "roofline": roofline,
            "roofline_hbm": {
                "bound": "hbm", "achieved": BYTES_PER_UMI * n / (ms_step * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": BYTES_PER_UMI * n / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "bytes_per_umi": BYTES_PER_UMI,
                "note": "whole step: 16 algorithmic bytes per unique UMI / step time"},
            "kernels_rocprof": kernels_view,
            "phases_ms": {"prep": prep_ms, "pairs": pair_ms, "collapse": coll_ms,
                          "finalize": mean("ms_finalize")},
            "counters": {"n_edges": s0["n_edges"], "n_candidates": s0["n_candidates"],
                         "n_rounds": s0["n_rounds"], "pairs_evaluated": s0["n_pairs_evaluated"]},
        }
        if world == 1 and extras and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(st, args.cpu_sample, args.k, args.p, args.umi_len,
                                               one_position)
        if world == 1 and extras and std_cfg2:
            out["end_to_end"] = end_to_end(2_000_000, 20_000, max(1, min(len(os.sched_getaffinity(0)), 16)))
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
