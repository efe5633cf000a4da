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


@pytest.mark.parametrize("umi_len,extra", [(24, []), (24, ["--stage", "host", "-k", "2"]), (45, ["--merge", "avgqual"]),
                                           (24, ["--devices", "0,0"])])
def test_umis_beyond_21_bases_end_to_end(tmp_path, umi_len, extra):
    """Dual 12 + 12 UMIs (24 bases) and longer: keys of several words through the device staging (or the
    host's), the fused kernel and the segment index, one or two workers; the decompressed output stream
    against the restatement (plain-Python staging model + the oracle's multi-word path)."""
    header, recs = bamio.synthetic_bam(9, 120, 50, umi_len=umi_len, err=0.02)
    rng = __import__("numpy").random.default_rng(umi_len)
    centre = rng.integers(0, 4, umi_len)
    for i in range(3000):  # one deep position of a few hundred clustered UMIs next to the small ones
        u = centre.copy()
        flip = rng.random(umi_len) < 0.08
        u[flip] = rng.integers(0, 4, int(flip.sum()))
        recs.append(bamio.make_record("d%d_%s" % (i, "".join("ACGT"[c] for c in u)), 0, 0, 777_000,
                                      int(rng.integers(0, 61)), [("M", 50)], 50, bytes([30] * 50)))
    kw = dict(merge="avgqual") if "avgqual" in extra else dict(merge="mapqual")
    if "-k" in extra:
        kw["k"] = 2
    oh, orecs, log = run_cli(tmp_path, header, recs, extra + ["--num-threads", "4"])
    exp, st = bamio.expected_output(recs, **kw)
    assert oh == header and orecs == exp
    assert "Number of UMIs: %d" % len(st["keys"]) in log


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


@pytest.mark.parametrize("extra,kw", [
    ([], {}),
    (["--remove-unpaired", "--remove-chimeric", "--merge", "avgqual"],
     dict(remove_unpaired=True, remove_chimeric=True, merge="avgqual")),
    (["--algo", "adj", "--num-threads", "4"], dict(algo="adj")),
])
def test_paired_end_mode(tmp_path, extra, kw):
    """--paired end to end: first mates are deduplicated on (strand, position, reference,
    template length), the second mates of the survivors are written when the reference changes
    and at the end, in file order (UcWriter, deduplicate_sam.rs:382-459)."""
    header, recs = bamio.synthetic_paired_bam(21, 120, 50)
    kw.setdefault("merge", "mapqual")
    oh, orecs, log = run_cli(tmp_path, header, recs, ["--paired"] + extra)
    exp, st = bamio.expected_output(recs, paired=True, **kw)
    assert oh == header
    assert orecs == exp
    n_second = sum(1 for r in orecs if bamio.parse_record(r)["flag"] & 0x80)
    assert n_second > 100  # mates really travel
    # an unsorted file: blocks of records in shuffled order, so the written records change
    # reference many times and every change triggers a partial mate pass
    import numpy as np
    blocks = [recs[i:i + 300] for i in range(0, len(recs), 300)]
    order = np.random.default_rng(3).permutation(len(blocks))
    mixed = [r for b in order for r in blocks[b]]
    oh, orecs, _ = run_cli(tmp_path, header, mixed, ["--paired"] + extra)
    exp2, _ = bamio.expected_output(mixed, paired=True, **kw)
    assert orecs == exp2
    tids = [bamio.parse_record(r)["tid"] for r in orecs]
    assert sum(1 for a, b in zip(tids, tids[1:]) if a != b) > 4
    c = st["counters"]
    assert "Number of input reads: %d" % c["total"] in log
    assert "Number of removed unmapped reads: %d" % c["unmapped"] in log
    assert "Number of unpaired reads: %d" % c["unpaired"] in log
    assert "Number of chimeric reads: %d" % c["chimeric"] in log


@pytest.mark.parametrize("extra,kw", [
    (["--num-threads", "5"], {}),
    (["--algo", "adj", "--keep-unmapped"], dict(algo="adj", keep_unmapped=True)),
    (["-k", "2", "--merge", "avgqual", "--data", "naive"], dict(k=2, merge="avgqual")),
])
def test_tag_mode_writes_every_read_with_its_cluster(tmp_path, extra, kw):
    header, recs = bamio.synthetic_bam(9, 200, 40, umi_len=10, err=0.04)
    kw.setdefault("merge", "mapqual")
    oh, orecs, log = run_cli(tmp_path, header, recs, ["--tag"] + extra)
    exp, st, groups = bamio.expected_tagged_output(recs, **kw)
    assert oh == header
    assert orecs == exp
    assert "Number of groups of reads: %d" % groups in log
    assert len(orecs) > groups or kw.get("algo") == "adj"


@pytest.mark.parametrize("extra,kw", [
    (["--merge", "avgqual"], dict(merge="avgqual")),
    (["--merge", "any", "-k", "2"], dict(merge="any", k=2)),
    (["--algo", "adj", "--keep-unmapped"], dict(algo="adj", merge="mapqual", keep_unmapped=True)),
])
def test_staging_on_the_gpu_and_on_the_host_write_the_same_file(tmp_path, extra, kw):
    """--stage gpu (umi_stage_reads: sorts + segmented merge on the device) and --stage host (the
    threaded hash-map staging) against the restatement, clipped reads, both strands, several
    references and N bases included; the two output files are the same bytes."""
    header, recs = bamio.synthetic_bam(11, 400, 50, umi_len=12, err=0.03)
    outs = {}
    for where in ("gpu", "host"):
        d = tmp_path / where
        d.mkdir()
        oh, orecs, log = run_cli(d, header, recs, extra + ["--stage", where])
        assert "staging (%s)" % where in log
        outs[where] = open(str(d / "out.bam"), "rb").read()
        exp, _ = bamio.expected_output(recs, **kw)
        assert oh == header and orecs == exp
    assert outs["gpu"] == outs["host"]
