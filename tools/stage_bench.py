#!/usr/bin/env python3
"""Read staging on the device (umi_stage_reads_device) by itself: reads of a config (UMI text, alignment
key, score) resident in HBM -> entries in canonical order; ms per call and reads/s.
    python3 tools/stage_bench.py [--config 2|3] [--reads N] [--calls K]
Under rocprofv3 --kernel-trace --stats it gives the per-kernel split (tools/README.md)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="2", choices=["2", "3"])
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--calls", type=int, default=10)
    args = ap.parse_args()
    import torch
    import umi_collapse_rs_amd as umi
    from umi_collapse_rs_amd import synth
    dev = torch.device("cuda", 0)
    if args.config == "2":
        n = args.reads or 1_000_000
        bases = synth.uniform_reads(2, n, 12)
        pos = np.zeros(n, np.int64)
        bits = 1
    else:
        n = args.reads or 10_000_000
        parts = [synth.molecule_reads(3, 10_000, 100, 12, first_position=p0) for p0 in range(0, n // 100, 10_000)]
        pos = np.concatenate([p[0] for p in parts])
        bases = np.concatenate([p[1] for p in parts])
        bits = max(1, int(pos.max()).bit_length())
    d_umi = torch.from_numpy(synth.BASES[bases].reshape(-1).copy()).to(dev)
    d_akey = torch.from_numpy(pos).to(dev)
    o_keys = torch.zeros(n, dtype=torch.int64, device=dev)
    o_freq = torch.zeros(n, dtype=torch.int32, device=dev)
    o_rep = torch.zeros(n, dtype=torch.int64, device=dev)
    o_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    ctx = umi.Context(0)
    stream = torch.cuda.current_stream().cuda_stream

    def once():
        return ctx.stage_reads_device(d_akey.data_ptr(), d_umi.data_ptr(), 0, n, 12, o_keys.data_ptr(), 0,
                                      o_freq.data_ptr(), o_rep.data_ptr(), o_off.data_ptr(), merge=0,
                                      align_key_bits=bits, stream=stream)
    ne, nb = once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.calls):
        once()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.calls * 1e3
    print("config %s: %d reads -> %d entries in %d positions: %.3f ms per call, %.3e reads/s" % (args.config, n, ne, nb, ms, n / ms * 1e3))
    ctx.close()


if __name__ == "__main__":
    main()
