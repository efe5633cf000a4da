#!/usr/bin/env python3
"""Benchmark of the hot path (BASELINE.json): UMI-pair Hamming comparisons/s on 12-bp UMIs,
per-position all-pairs adjacency + directional collapse, and reads deduplicated/s.

A step = one pass of the whole hot path (filter keys, pair evaluation, collapse, kept mask) over one
batch of synthetic input resident in HBM.

N = 1 (the judged line): BASELINE config 2 -- 1,000,000 reads, 12-bp UMIs, ONE alignment position
(uniform UMIs, ~9.7e5 unique -> W ~ 4.7e11 unordered pairs), --data naive --algo dir -k 1 -p 0.5.
The other single-GPU shapes ride along under "configs": config 3 (10 M reads in 100,000 positions),
one GPU's share of config 5 (6.25 M reads, 20-bp, k = 2) and 2m (one deep position from the
molecule model), each with its own roofline and CPU baseline.

N > 1: BASELINE config 4 -- 100 M reads in 10^6 positions over 8 GPUs, i.e. 12.5 M reads in 125,000
positions per rank (weak scaling in N), buckets sharded by rank with no data-path collective, the
kept mask packed to bits and all-gathered over RCCL/xGMI every step.  Config 5's per-rank share and
config 2 (one deep position per rank) follow in the same run under "configs".  `--config 2` makes
config 2 the top-level workload again; `--split` is strong scaling of ONE deep position (pair work
split over the ranks, edge lists all-gathered, collapse replicated).

`value` is an EFFECTIVE rate: W / t, the unordered pairs of the positions per second of hot-path
time.  A deep position is not walked pair by pair: it is cut into n-gram sub-buckets (two UMIs
within k substitutions agree on one of k+1 base ranges) and only the pairs inside them are compared
-- `walked_fraction` of W, `value_executed` per second; the result is the all-pairs kernels'
(`value_bruteforce`: the same step with the partition off, every pair evaluated), bit for bit.

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
BYTES_PER_UMI = 16       # 8 B key + 4 B freq in, 4 B label out
OPS_PER_PAIR = 3         # xor, popcount, compare: the ideal N-free one-word form
OPS_PER_PAIR_REAL = 5    # ... with 64-bit xor and popcount as two 32-bit ops each
PROFILE_JSON = os.path.join(ROOT, "profiles", "r03_counters.json")
KERNEL_NAMES = {0: None, 1: "small_bucket_kernel", 2: "seg_pair_kernel"}


def stamped_sources():
    """The sources whose kernels the counters of profiles/r03_counters.json belong to: the hot path's.  The
    read staging (umihip_stage.hip, umihip_radix.hip) is measured by itself (tools/stage_prof.sh,
    profiles/r03_staging_kernel_split.txt) and not part of the stamp."""
    return [f for f in sorted(glob.glob(os.path.join(ROOT, "umi_collapse_rs_amd", "csrc", "*")))
            if os.path.basename(f) not in ("umihip_stage.hip", "umihip_radix.hip")]


def source_sha256():
    """sha256 over the kernel and host sources of the hot path (sorted file names, contents)."""
    h = hashlib.sha256()
    for f in stamped_sources():
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def load_profile(cfg):
    """rocprofv3 numbers of one config's step (tools/profile_r03.sh -> tools/summarize_r03.py): per
    kernel the average duration, the SQ instruction counts and the fabric-side bytes, stamped with
    the sha256 of the sources they were measured on.  Returns (entry or None, stale reason or None)."""
    if not os.path.exists(PROFILE_JSON):
        return None, "no " + os.path.relpath(PROFILE_JSON, ROOT)
    prof = json.load(open(PROFILE_JSON))
    entry = prof.get("configs", {}).get(cfg)
    if entry is None:
        return None, "config %s not profiled" % cfg
    if entry.get("source_sha256") != source_sha256():
        return entry, "counters measured on other sources (sha256 %s...)" % entry.get("source_sha256", "?")[:12]
    return entry, None


# ---- workloads ------------------------------------------------------------------------------
def make_workload(cfg, rank, reads=None):
    """Synthetic input of one rank for a BASELINE config (umi_collapse_rs_amd/synth.py, seeded)."""
    from umi_collapse_rs_amd import synth
    if cfg == "2":
        reads = reads or 1_000_000
        st = synth.config2(seed=2 + 1000 * rank, n_reads=reads, umi_len=12)
        return dict(cfg=cfg, st=st, umi_len=12, k=1, reads=reads, one_position=True,
                    workload="BASELINE config 2 per GPU: %d reads, 12-bp UMIs, one alignment position "
                             "(uniform UMIs)" % reads)
    if cfg == "2m":
        reads = reads or 1_000_000
        st = synth.config2m(seed=22 + 1000 * rank, n_reads=reads, umi_len=12)
        return dict(cfg=cfg, st=st, umi_len=12, k=1, reads=reads, one_position=True,
                    workload="one deep alignment position from the molecule model: %d reads of %d molecules, "
                             "12-bp UMIs, error 0.01 per base" % (reads, st["n_molecules"]))
    if cfg in ("3", "4"):
        reads = reads or (10_000_000 if cfg == "3" else 12_500_000)  # config 4: 100 M reads over 8 GPUs
        st = synth.config3(seed=int(cfg) + 1000 * rank, n_reads=reads, n_positions=reads // 100, umi_len=12)
        what = ("BASELINE config 3" if cfg == "3" else
                "BASELINE config 4, one GPU's share (an eighth of 100 M reads in 10^6 positions)")
        return dict(cfg=cfg, st=st, umi_len=12, k=1, reads=reads, one_position=False,
                    workload="%s: %d reads, 12-bp UMIs, %d alignment positions (molecule model)"
                             % (what, reads, reads // 100))
    if cfg == "5":
        reads = reads or 6_250_000
        st = synth.config3(seed=5 + 1000 * rank, n_reads=reads, n_positions=reads // 100, umi_len=20)
        return dict(cfg=cfg, st=st, umi_len=20, k=2, reads=reads, one_position=False,
                    workload="BASELINE config 5, one GPU's share (an eighth of 50 M reads): %d reads, 20-bp UMIs, "
                             "%d alignment positions, k=2 (molecule model)" % (reads, reads // 100))
    if cfg == "wide24":
        reads = reads or 1_000_000
        st = synth.config2m(seed=24 + 1000 * rank, n_reads=reads, umi_len=24)
        return dict(cfg=cfg, st=st, umi_len=24, k=1, reads=reads, one_position=True,
                    workload="one deep alignment position of dual 12 + 12 UMIs (24 bases: keys of two words), molecule "
                             "model: %d reads of %d molecules, error 0.01 per base" % (reads, st["n_molecules"]))
    raise SystemExit("unknown config " + cfg)


def pairs_of(bucket_off):
    sz = np.diff(bucket_off.astype(np.int64))
    return int((sz * (sz - 1) // 2).sum())


def cpu_baseline(wl, n_sample, p, all_cores=True):
    """The oracle (a scalar C port of the reference path) on a bounded sample of the same
    workload: one deep position -> n_sample unique UMIs drawn in rank order from it;
    many positions -> a prefix of whole buckets."""
    import oracle as orc
    st, umi_len, k = wl["st"], wl["umi_len"], wl["k"]
    n = len(st["keys"])
    if wl["one_position"]:
        rng = np.random.default_rng(12345)
        idx = np.sort(rng.choice(n, size=min(n_sample, n), replace=False))
        keys, freq = st["keys"][idx], st["freq"][idx]
        boff = np.array([0, len(idx)], np.uint64)
        what = "%d unique UMIs sampled in rank order from the same position" % len(idx)
    else:
        nb = min(len(st["bucket_off"]) - 1, n_sample)
        boff = st["bucket_off"][: nb + 1]
        m = int(boff[-1])
        keys, freq = st["keys"][:m], st["freq"][:m]
        what = "the first %d buckets (%d unique UMIs)" % (nb, m)
    w = pairs_of(boff)
    t0 = time.perf_counter()
    run = orc.dedup_batch_wide if keys.ndim == 2 else orc.dedup_batch
    kept, _, calls = run(keys, None, freq, boff, umi_len, k, p)
    dt = time.perf_counter() - t0
    reads = int(freq.astype(np.int64).sum())
    out = {"value": w / dt, "unit": "UMI-pair comparisons/s", "cores": 1, "kind": "port",
           "sample": "%s (W=%d pairs, %d umi_dist calls, %.1f s); the Rust reference cannot be "
                     "built here (no rustc)" % (what, w, calls, dt),
           "dist_calls_per_s": calls / dt, "umis_per_s": len(keys) / dt, "reads_per_s": reads / dt}
    # The machine's CPU ceiling for a bucket-parallel host (the reference itself is single
    # threaded, deduplicate_sam.rs:207): the same sample once per core, concurrently (ctypes
    # drops the GIL inside the oracle call).  Reported beside `value`, never instead of it.
    if all_cores:
        from concurrent.futures import ThreadPoolExecutor
        cores = max(1, min(len(os.sched_getaffinity(0)), 16))  # the box's CPU share for one GPU
        if cores > 1:
            t0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                list(ex.map(lambda _: run(keys, None, freq, boff, umi_len, k, p), range(cores)))
            dta = time.perf_counter() - t0
            out["all_cores"] = {"value": cores * w / dta, "cores": cores, "reads_per_s": cores * reads / dta,
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
        best, walls = None, []
        for _ in range(3):
            # (a pause between the runs: the driver puts the GPU context of a process that has just
            # ended away in the background, ~0.12 s during which the next one waits for its own)
            time.sleep(1.0)
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
            walls.append(round(dt, 4))
            if best is None or dt < best[0]:
                best = (dt, split)
        return {"reads": n_reads, "positions": n_positions, "bam_bytes": os.path.getsize(src),
                "wall_s": best[0], "reads_per_s": n_reads / best[0], "split_s": best[1], "wall_s_runs": walls,
                "threads": threads, "generate_s": t_gen,
                "note": "bin/umicollapse --merge avgqual, process start to exit (HIP context "
                        "creation included), BAM generated on this box by tools/make_bam.py"}
    except Exception as e:  # the hot-path numbers stand without it
        return {"error": repr(e)[:300]}
    finally:
        for f in (src, dst):
            if os.path.exists(f):
                os.unlink(f)


# ---- one workload resident on this rank's GPU ---------------------------------------------------
class Resident:
    """Inputs of one workload in HBM, the bucket table included, and the buffers of the mask gather.
    step() = one pass of the hot path on this rank's share (+ the packed-mask all-gather for N > 1)."""

    def __init__(self, wl, ctx, dev, world, rank, dist, p, split=False, coll=None):
        import torch
        self.wl, self.ctx, self.dev, self.world, self.rank, self.dist, self.p = wl, ctx, dev, world, rank, dist, p
        self.split = split and world > 1
        self.coll = world > 1 if coll is None else coll  # the mask gather runs (a 1-rank group: rehearsal)
        st = wl["st"]
        self.n = len(st["keys"])
        self.boff = st["bucket_off"]
        self.w_local = pairs_of(self.boff)
        self.n_words = st["keys"].shape[1] if st["keys"].ndim == 2 else 1
        self.d_keys = torch.from_numpy(np.ascontiguousarray(st["keys"]).view(np.int64)).to(dev)
        self.d_freq = torch.from_numpy(st["freq"]).to(dev)
        self.d_kept = torch.zeros(max(1, self.n), dtype=torch.uint8, device=dev)
        self.d_boff = torch.from_numpy(np.ascontiguousarray(self.boff).view(np.int64)).to(dev)  # an input too
        self.stream = torch.cuda.current_stream().cuda_stream
        sizes, w_all = [self.n], [self.w_local]
        if self.coll:
            t = torch.tensor([self.n, self.w_local], dtype=torch.int64, device=dev)
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            sizes = [int(x[0].item()) for x in allt]
            w_all = [int(x[1].item()) for x in allt]
        self.sizes = sizes
        self.w_total = self.w_local if self.split else sum(w_all)
        self.reads_total = wl["reads"] * (1 if self.split else world)
        # the kept mask travels as bits: ceil(n / 8) bytes per rank, padded to the largest slice (two
        # sets: the all-gather of one step runs on the collective's own stream while the next step's
        # kernels run on this one; a set is reused two steps later, after its collective was waited for)
        width = (max(sizes) + 7) // 8
        self.gather_in = [torch.zeros(width, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.gather_out = [torch.zeros(width * world, dtype=torch.uint8, device=dev) if self.coll else None
                           for _ in range(2)]
        self.gather_work = [None, None]
        self.step_no = 0

    def step(self, ctx=None):
        c = ctx or self.ctx
        wl = self.wl
        if self.split:
            from umi_collapse_rs_amd.sharded import split_dedup_device
            s = split_dedup_device(c, self.dist, self.d_keys, None, self.d_freq, self.boff, wl["umi_len"],
                                   self.d_kept, k=wl["k"], percentage=self.p)
            for f in ("ms_prep", "ms_collapse", "ms_finalize", "ms_kernel"):
                s.setdefault(f, 0.0)
            s.setdefault("n_candidates", 0)
            s.setdefault("kernel_id", 0)
            return s
        if self.n_words > 1:
            s = c.dedup_batch_wide_device(self.d_keys.data_ptr(), 0, self.n_words, self.d_freq.data_ptr(), self.boff,
                                          wl["umi_len"], self.d_kept.data_ptr(), 0, k=wl["k"], percentage=self.p,
                                          stream=self.stream)
        elif self.coll:
            # the call in two halves: the packing and the all-gather of the mask are enqueued behind its
            # kernels while those run (~30 us of host time per step that would otherwise pass with the GPU
            # idle); where the call has decisions to take on the host, begin runs it to its end
            c.dedup_batch_device_begin(self.d_keys.data_ptr(), 0, self.d_freq.data_ptr(), self.boff, wl["umi_len"],
                                       self.d_kept.data_ptr(), 0, k=wl["k"], percentage=self.p, stream=self.stream,
                                       d_bucket_off=self.d_boff.data_ptr())
            s = None
        else:
            s = c.dedup_batch_device(self.d_keys.data_ptr(), 0, self.d_freq.data_ptr(), self.boff, wl["umi_len"],
                                     self.d_kept.data_ptr(), 0, k=wl["k"], percentage=self.p, stream=self.stream,
                                     d_bucket_off=self.d_boff.data_ptr())
        if self.coll:  # all-gatherv of the kept mask: packed to bits on the device, padded all_gather
            slot = self.step_no & 1
            self.step_no += 1
            if self.gather_work[slot] is not None:
                self.gather_work[slot].wait()
            c.pack_mask_device(self.d_kept.data_ptr(), self.n, self.gather_in[slot].data_ptr(), stream=self.stream)
            self.gather_work[slot] = self.dist.all_gather_into_tensor(self.gather_out[slot], self.gather_in[slot],
                                                                      async_op=True)  # RCCL over xGMI
            if s is None:
                s = c.dedup_batch_end()
        return s

    def drain_gathers(self):
        for i in range(2):
            if self.gather_work[i] is not None:
                self.gather_work[i].wait()
                self.gather_work[i] = None

    def timed(self, n_steps, ctx=None):
        """EXACTLY n_steps steps between barrier + synchronize on both sides; max over ranks."""
        import torch
        if self.coll:
            self.dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ss = [self.step(ctx) for _ in range(n_steps)]
        self.drain_gathers()  # every step's mask has arrived everywhere inside the timed region
        torch.cuda.synchronize()
        if self.coll:
            self.dist.barrier()
        dt = time.perf_counter() - t0
        if self.coll:
            tt = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, ss

    def check_gathered(self, kept_n):
        """this rank's slice of the gathered bit mask is its own kept mask"""
        import torch
        if not self.coll or self.split:
            return
        slot = (self.step_no - 1) & 1
        width = self.gather_in[0].numel()
        mine = self.gather_out[slot][self.rank * width:(self.rank + 1) * width]
        bits = ((mine[:, None] >> torch.arange(8, device=self.dev, dtype=torch.uint8)[None, :]) & 1).sum()
        assert int(bits.item()) == kept_n, "gathered mask differs from the local one"


def roofline_of(wl, n, stats, ms_step):
    """the roofline of the kernel that does the step's pair work, from its
    ALGORITHMIC work and its duration measured live with HIP events in this run (umi_stats.ms_kernel,
    recorded on the stream the kernel is launched on, in a block of steps of its own: see measure).
    A deep position (seg_pair_kernel) is bound by
    integer VALU work -- 3 lane-ops per pair it evaluates (xor, popcount, compare), 0 algorithmic HBM
    bytes per pair; a batch of small positions (the fused small_bucket_kernel) streams 16 B per unique
    UMI and is priced against HBM.  traffic: fabric-side bytes of one launch from the committed
    rocprofv3 --pmc passes (FETCH_SIZE doubled as the gfx950 guide prescribes, + WRITE_SIZE)."""
    s0 = stats[-1]
    kid = int(s0.get("kernel_id", 0))
    kernel_ms = float(np.mean([s["ms_kernel"] for s in stats])) if kid else None
    prof, stale_why = load_profile(wl["cfg"])
    kname = KERNEL_NAMES.get(kid)
    pk = None
    if prof and kname:
        pk = next((v for k_, v in prof["kernels"].items() if kname in k_), None)
    traffic = None
    if pk and pk.get("FETCH_SIZE_KB") is not None and pk.get("WRITE_SIZE_KB") is not None:
        traffic = (2 * pk["FETCH_SIZE_KB"] + pk["WRITE_SIZE_KB"]) * 1024
    alg_bytes = BYTES_PER_UMI * n
    out = {"kernel": kname, "kernel_us": None if kernel_ms is None else kernel_ms * 1e3,
           "kernel_us_rocprof": None if not pk else pk["avg_us"],
           "traffic": traffic, "traffic_ratio": None if traffic is None else traffic / alg_bytes,
           "algorithmic_bytes": alg_bytes, "counters_source": os.path.relpath(PROFILE_JSON, ROOT),
           "counters_stale": bool(stale_why)}
    if stale_why:
        out["counters_stale_why"] = stale_why
    if kid == 2:
        pe = s0["n_pairs_evaluated"]
        ach = OPS_PER_PAIR * pe / (kernel_ms * 1e-3) / 1e12
        out.update({"bound": "valu", "unit": "Tlaneop/s", "peak": VALU_PEAK_TLANEOPS, "achieved": ach,
                    "frac": ach / VALU_PEAK_TLANEOPS, "frac_algorithmic": ach / VALU_PEAK_TLANEOPS,
                    "frac_algorithmic_5op": ach * OPS_PER_PAIR_REAL / OPS_PER_PAIR / VALU_PEAK_TLANEOPS,
                    "pairs_evaluated": pe, "ops_per_pair": OPS_PER_PAIR,
                    "hbm_frac_whole_step": alg_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS})
        if pk and pk.get("SQ_INSTS_VALU") is not None:
            out["frac_issue"] = pk["SQ_INSTS_VALU"] * 64 / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS
            out["valu_lane_insts_per_pair"] = pk["SQ_INSTS_VALU"] * 64 / max(pe, 1)
    elif kid == 1:
        ach = alg_bytes / (kernel_ms * 1e-3) / 1e9
        pe = s0["n_pairs_evaluated"]
        out.update({"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": ach,
                    "frac": ach / HBM_PEAK_GBS, "frac_algorithmic": ach / HBM_PEAK_GBS,
                    "frac_whole_step": alg_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "pairs_evaluated": pe,
                    "valu_frac_algorithmic": OPS_PER_PAIR * pe / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS})
        if pk and pk.get("SQ_INSTS_VALU") is not None:
            out["frac_issue"] = pk["SQ_INSTS_VALU"] * 64 / (kernel_ms * 1e-3) / 1e12 / VALU_PEAK_TLANEOPS
            out["valu_wave_insts_per_position"] = pk["SQ_INSTS_VALU"] / max(1, len(wl["st"]["bucket_off"]) - 1)
    else:
        out.update({"bound": "valu", "unit": "Tlaneop/s", "peak": VALU_PEAK_TLANEOPS, "achieved": None, "frac": None})
    return out


def measure(res, steps, warmup, profile=True):
    """warmup untimed steps, then `steps` timed ones; the block of one workload's result.
    The timed steps run with the library's event records off: a hipEventRecord between two kernels
    costs the stream ~5 us, and a call records up to nine -- 0.04 ms of a 0.14 ms step.  The phase
    times and the duration of the kernel the roofline is quoted on come from a second block of the
    same `steps` steps, on the same stream, with the records on (`ms_per_step_with_events`)."""
    import torch
    res.ctx.set_option("profile", 0)
    for _ in range(warmup):
        res.step()
    res.drain_gathers()
    dt, stats_plain = res.timed(steps)
    kept_n = int(res.d_kept[:res.n].sum().item())
    assert kept_n == stats_plain[-1]["n_kept"]
    res.check_gathered(kept_n)
    ms_step = dt / steps * 1e3
    stats, ms_events = stats_plain, None
    if profile:
        res.ctx.set_option("profile", 1)
        res.step()
        res.drain_gathers()
        dt_ev, stats = res.timed(steps)
        ms_events = dt_ev / steps * 1e3
        res.ctx.set_option("profile", 0)
    mean = lambda f: float(np.mean([s[f] for s in stats]))
    s0 = stats[-1]
    wl = res.wl
    out = {
        "workload": "%s, --data naive --algo dir -k %d -p %g" % (wl["workload"], wl["k"], res.p),
        "ms_per_step": ms_step, "steps": steps, "ms_per_step_with_events": ms_events,
        "value": res.w_total * steps / dt, "unit": "UMI-pair comparisons/s (effective: W / t)",
        "value_executed": s0["n_pairs_evaluated"] * (1 if res.split else res.world) * steps / dt,
        "walked_fraction": min(1.0, s0["n_pairs_evaluated"] / max(res.w_local, 1)) if res.w_local else None,
        "reads_per_s": res.reads_total * steps / dt,
        "unique_umis_rank0": res.n, "positions_rank0": len(res.boff) - 1, "pairs_W_total": res.w_total,
        "kept_rank0": kept_n,
        "phases_ms": {"prep": mean("ms_prep"), "pairs": mean("ms_pairs"), "collapse": mean("ms_collapse"),
                      "finalize": mean("ms_finalize")},
        "counters": {"n_edges": s0["n_edges"], "n_candidates": s0["n_candidates"], "n_rounds": s0["n_rounds"],
                     "pairs_evaluated": s0["n_pairs_evaluated"]},
        "roofline": roofline_of(wl, res.n, stats, ms_step),
    }
    return out, dt, stats, kept_n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=0, help="reads per rank of the top-level workload (0: the config's)")
    ap.add_argument("-p", type=float, default=0.5)
    ap.add_argument("--cpu-sample", type=int, default=80_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed block of the top-level workload (profiling runs): no sustained block, "
                         "no brute-force step, no host-buffer call, no CPU baseline, no end-to-end run, no "
                         "other configs")
    ap.add_argument("--opt", action="append", default=[], help="ctx option name=value (tuning)")
    ap.add_argument("--split", action="store_true",
                    help="N>1 only: strong scaling -- ONE giant position (config 2), its pair work split over "
                         "the ranks, edge lists all-gathered, collapse replicated")
    ap.add_argument("--config", default=None, choices=["2", "2m", "3", "4", "5", "wide24"],
                    help="the top-level workload per GPU (default: 2 at N=1, 4 at N>1): 2 = one giant position of "
                         "uniform UMIs, 2m = one deep position from the molecule model, 3 = 10M reads in 100k "
                         "positions, 4 = one GPU's share of the 8-GPU config (12.5M reads in 125k positions), 5 = "
                         "one GPU's share of 50M reads with 20-bp UMIs, k=2, wide24 = one deep position of 24-base "
                         "(dual 12 + 12) UMIs")
    ap.add_argument("--no-profile", action="store_true",
                    help="no HIP events inside the library's calls (phases_ms and the roofline's kernel time are then "
                         "missing): what the event records cost the step")
    ap.add_argument("--force-collective", action="store_true",
                    help="N=1 rehearsal of the N>1 step: a process group of one rank is formed (RCCL unless "
                         "BENCH_BACKEND says otherwise) and every step packs and all-gathers its mask")
    ap.add_argument("--also", default=None,
                    help="comma list of the configs measured after the top-level one, under \"configs\" "
                         "(default: 3,5,2m at N=1; 5,2 at N>1; 'none')")
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
    coll = world > 1 or args.force_collective
    backend = None
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        backend = os.environ.get("BENCH_BACKEND", "nccl")  # nccl = RCCL over xGMI
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import umi_collapse_rs_amd as umi
    from umi_collapse_rs_amd import synth

    split = args.split and world > 1
    cfg = args.config or ("2" if (world == 1 or split) else "4")
    extras = not args.no_extras
    if args.also is None:
        also = [] if (not extras or split) else ([c for c in ("3", "5", "2m", "wide24") if c != cfg] if world == 1
                                                 else [c for c in ("5", "2") if c != cfg])
    else:
        also = [] if args.also == "none" else [c for c in args.also.split(",") if c]

    ctx = umi.Context(dev_index)
    opts = {}
    for o in args.opt:
        name, val = o.split("=")
        ctx.set_option(name, int(val))
        opts[name] = int(val)

    # ---- the top-level workload
    wl = make_workload(cfg, 0 if split else rank, args.reads or None)
    res = Resident(wl, ctx, dev, world, rank, dist, args.p, split, coll)
    block, dt, stats, kept_n = measure(res, args.steps, args.warmup, not args.no_profile)
    kept_ref = res.d_kept.clone()
    n, st = res.n, wl["st"]
    ms_step = block["ms_per_step"]

    # a second block of at least one second: clocks at steady state
    sustained = None
    if extras:
        n_sus = max(args.steps, int(1.05 / max(dt / args.steps, 1e-6)) + 1)
        dts, _ = res.timed(n_sus)
        sustained = {"steps": n_sus, "seconds": dts, "ms_per_step": dts / n_sus * 1e3}

    # the same step with every pair of the position evaluated: the all-pairs popcount tile kernel
    # (no n-gram partition)
    brute = None
    if extras and world == 1 and wl["one_position"] and n <= 1_200_000:
        cb = umi.Context(dev_index)
        try:
            cb.set_option("seg_index", 0)
            res.step(cb)
            nb_steps = 3
            dtb, sb = res.timed(nb_steps, cb)
            assert bool((res.d_kept == kept_ref).all().item()), "brute-force kept mask differs"
            brute = {"value": res.w_local * nb_steps / dtb, "ms_per_step": dtb / nb_steps * 1e3,
                     "steps": nb_steps, "pairs_evaluated": sb[-1]["n_pairs_evaluated"],
                     "kernel": "pair_kernel (popcount tiles: xor + bcnt + min3 on every pair of the position; "
                               "option seg_index=0); kept mask equal to the default path's"}
        finally:
            cb.close()

    # the same pass through the host-buffer entry point (H2D of keys/freq + D2H of the mask
    # inside the call): the PCIe-inclusive rate, reported beside `value`, never as it
    host_ms = None
    if rank == 0 and world == 1 and extras and res.n_words == 1:
        ctx.dedup_batch(st["keys"], None, st["freq"], res.boff, wl["umi_len"], k=wl["k"], percentage=args.p,
                        want_root=False)
        t1 = time.perf_counter()
        hk, _, _ = ctx.dedup_batch(st["keys"], None, st["freq"], res.boff, wl["umi_len"], k=wl["k"],
                                   percentage=args.p, want_root=False)
        host_ms = (time.perf_counter() - t1) * 1e3
        assert int(hk.sum()) == kept_n

    # read staging on the device (N2): the same 1,000,000 reads as UMI text + alignment key, resident
    # in HBM, through umi_stage_reads_device; its output must be the arrays the timed steps ran on
    staging = None
    if rank == 0 and world == 1 and extras and cfg == "2":
        reads = wl["reads"]
        b2 = synth.uniform_reads(2, reads, 12)
        d_umi = torch.from_numpy(synth.BASES[b2].reshape(-1).copy()).to(dev)
        d_akey = torch.zeros(reads, dtype=torch.int64, device=dev)
        o_keys = torch.zeros(reads, dtype=torch.int64, device=dev)
        o_freq = torch.zeros(reads, dtype=torch.int32, device=dev)
        o_rep = torch.zeros(reads, dtype=torch.int64, device=dev)
        o_off = torch.zeros(reads + 1, dtype=torch.int64, device=dev)

        def stage_once():
            return ctx.stage_reads_device(d_akey.data_ptr(), d_umi.data_ptr(), 0, reads, 12,
                                          o_keys.data_ptr(), 0, o_freq.data_ptr(), o_rep.data_ptr(),
                                          o_off.data_ptr(), merge=0, align_key_bits=1, stream=res.stream)
        ne, nbk = stage_once()
        assert ne == n and nbk == 1 and bool((o_keys[:n] == res.d_keys).all().item()) and bool(
            (o_freq[:n] == res.d_freq).all().item())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_st = 10
        for _ in range(n_st):
            stage_once()
        torch.cuda.synchronize()
        ms_stage = (time.perf_counter() - t1) / n_st * 1e3
        staging = {"ms_per_call": ms_stage, "reads_per_s": reads / (ms_stage * 1e-3),
                   "reads_per_s_with_hot_path": reads / ((ms_stage + ms_step) * 1e-3),
                   "note": "umi_stage_reads_device on the same reads (12-byte UMI text + alignment key per read, "
                           "resident in HBM): encode, sort by (position, UMI, file index), merge equal UMIs, rank "
                           "order; output equal to the arrays of the timed steps (checked)"}

    cpu = None
    if rank == 0 and world == 1 and extras and not args.no_cpu_baseline:
        cpu = cpu_baseline(wl, args.cpu_sample if wl["one_position"] else 50_000, args.p)

    # ---- the other configs of this GPU count, each the same way (20 steps unless told otherwise)
    del res, kept_ref
    others = {}
    for c in also:
        w2 = make_workload(c, rank)
        r2 = Resident(w2, ctx, dev, world, rank, dist, args.p, False, coll)
        b2, _, _, _ = measure(r2, args.steps, args.warmup, not args.no_profile)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            # (a prefix of buckets / a 40,000-UMI cut that the oracle finishes in about a second)
            b2["cpu_baseline"] = cpu_baseline(w2, 40_000 if w2["one_position"] else 5_000, args.p, all_cores=False)
        others[c] = b2
        del r2, w2
        torch.cuda.empty_cache()

    if rank == 0:
        std = (not opts and not args.reads)
        metric = ("UMI-pair Hamming comparisons/s, effective (W pairs of the positions / hot-path time; "
                  "12-bp UMIs, all-pairs adjacency + directional collapse; the n-gram partition compares "
                  "walked_fraction of W, the all-pairs kernels' result bit for bit)")
        out = {
            "metric": metric,
            "value": block["value"],
            "unit": "UMI-pair comparisons/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "ms_per_step_with_events": block["ms_per_step_with_events"], "higher_is_better": True,
            "scaling": "strong" if split else "weak",
            "vs_baseline": None, "dtype": "u32" if wl["umi_len"] <= 16 else "u64", "data": "synthetic",
            "config": {"workload": block["workload"], "baseline_config": cfg, "standard": std,
                       "reads_per_rank": wl["reads"], "positions_per_rank": block["positions_rank0"],
                       "unique_umis_rank0": n, "pairs_W_total": block["pairs_W_total"], "collective": backend,
                       "parallelism": ("pair-work split x%d + edge all-gatherv" % world) if split
                       else ("bucket-sharded x%d, packed kept mask all-gathered (RCCL)" % world if world > 1
                             else "one GPU")},
            "value_effective": block["value"],
            "value_executed": block["value_executed"],
            "walked_fraction": block["walked_fraction"],
            "value_bruteforce": brute,
            "sustained": sustained,
            "sustained_ms_per_step": None if sustained is None else sustained["ms_per_step"],
            "reads_per_s": block["reads_per_s"],
            "staging_device": staging,
            "host_buffer_path": None if host_ms is None else {
                "ms_per_call": host_ms, "pairs_per_s": block["pairs_W_total"] / (host_ms * 1e-3),
                "note": "umi_dedup_batch with pageable host arrays: PCIe copies included"},
            "kept_rank0": kept_n,
            "roofline": block["roofline"],
            "phases_ms": block["phases_ms"],
            "counters": block["counters"],
            "configs": others,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if world == 1 and extras and cfg == "2" and std:
            out["end_to_end"] = end_to_end(2_000_000, 20_000, max(1, min(len(os.sched_getaffinity(0)), 16)))
        print(json.dumps(out))
    ctx.close()
    if coll:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
