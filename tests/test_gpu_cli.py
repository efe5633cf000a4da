"""umicollapse end to end on the GPU: BASELINE config 1 (10k-read synthetic sorted BAM,
--mode bam --data naive --merge avgqual) and variants, decompressed output stream compared
record for record with the reference restatement (tests/bamio.py + oracle)."""
import os
import subprocess

import pytest

import bamio

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "umi_collapse_rs_amd", "bin", "umicollapse")


def run_cli(tmp_path, header, recs, extra):
    src, dst = str(tmp_path / "in.bam"), str(tmp_path / "out.bam")
    with open(src, "wb") as f:
        f.write(bamio.bgzf_compress(header + b"".join(recs)))
    r = subprocess.run([CLI, "-i", src, "-o", dst] + extra, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    out_header, out_recs = bamio.split_records(bamio.bgzf_decompress(open(dst, "rb").read()))
    return out_header, out_recs, r.stderr


def test_config1_10k_reads_avgqual(tmp_path):
    header, recs = bamio.synthetic_bam(1, 500, 20, umi_len=12, err=0.01, extras=False)
    assert len(recs) == 10_000
    oh, orecs, log = run_cli(tmp_path, header, recs,
                             ["--mode", "bam", "--data", "naive", "--merge", "avgqual", "--num-threads", "4"])
    exp, st = bamio.expected_output(recs, k=1, p=0.5, algo="dir", merge="avgqual")
    assert oh == header
    assert orecs == exp
    assert "Number of input reads: 10000" in log
    assert "Number of unique alignment positions: 500" in log
    assert "Number of UMIs: %d" % len(st["keys"]) in log
    assert "Number of reads after deduplicating: %d" % len(exp) in log


@pytest.mark.parametrize("extra,kw", [
    (["--merge", "mapqual"], dict(merge="mapqual")),
    (["--merge", "any", "-k", "2", "-p", "0.75"], dict(merge="any", k=2, p=0.75)),
    (["--algo", "adj"], dict(algo="adj", merge="mapqual")),
    (["--keep-unmapped", "-u", "10"], dict(merge="mapqual", keep_unmapped=True, umi_len=10)),
    (["--data", "ngrambktree", "-k", "0"], dict(merge="mapqual", k=0)),
])
def test_cli_variants_with_clips_strands_refs_and_n(tmp_path, extra, kw):
    header, recs = bamio.synthetic_bam(7, 150, 60, umi_len=10, err=0.03)
    oh, orecs, _ = run_cli(tmp_path, header, recs, extra)
    exp, _ = bamio.expected_output(recs, **kw)
    assert oh == header
    assert orecs == exp


def test_deep_position_with_and_without_pruning(tmp_path):
    """One position with ~6,000 distinct 8-bp UMIs (bit-sliced tiles; label propagation over a
    giant component): --data naive (plain all-pairs) and the default --data (range pruning) must
    both give the reference restatement's records."""
    import numpy as np
    rng = np.random.default_rng(5)
    header = bamio.make_header([("chr1", 1_000_000)])
    recs = []
    for i in range(9000):
        umi = "".join("ACGT"[c] for c in rng.integers(0, 4, 8))
        recs.append(bamio.make_record("r%d_%s" % (i, umi), 0, 0, 5000, int(rng.integers(0, 61)),
                                      [("M", 50)], 50, bytes([30] * 50)))
    exp, st = bamio.expected_output(recs, k=1, merge="mapqual")
    assert len(st["keys"]) > 5000
    for data in ("naive", "ngrambktree"):
        oh, orecs, log = run_cli(tmp_path, header, recs, ["--data", data, "--num-threads", "4"])
        assert orecs == exp, data
