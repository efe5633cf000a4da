#!/usr/bin/env python3
"""Write a synthetic coordinate-sorted BAM of BASELINE shape for end-to-end
timing of umicollapse: python tools/make_bam.py out.bam --reads 2000000 --positions 20000"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--positions", type=int, default=10_000)
    ap.add_argument("--umi-len", type=int, default=12)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import bamio
    from umi_collapse_rs_amd import synth
    rpp = a.reads // a.positions
    header = bamio.make_header([("chr1", 250_000_000)])
    rng = np.random.default_rng(a.seed)
    with open(a.out, "wb") as f:
        buf = bytearray(header)
        idx = 0
        for p0 in range(0, a.positions, 2000):
            npos = min(2000, a.positions - p0)
            pos, bases = synth.molecule_reads(a.seed, npos, rpp, a.umi_len, first_position=p0)
            letters = synth.BASES[bases]
            quals = rng.integers(20, 41, (len(pos), 50)).astype(np.uint8)
            for i in range(len(pos)):
                buf += bamio.make_record("r%d_%s" % (idx, letters[i].tobytes().decode()), 0, 0,
                                         1000 + 10 * int(pos[i]), 60, [("M", 50)], 50, quals[i].tobytes())
                idx += 1
            f.write(bamio.bgzf_compress(bytes(buf), level=1)[:-28])  # drop the EOF marker between chunks
            buf = bytearray()
        f.write(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    print("wrote %s: %d reads, %d positions" % (a.out, idx, a.positions))


if __name__ == "__main__":
    main()
