"""Read staging on the device (umi_stage_reads, SURVEY.md 8f N2's "second kernel") against the
oracle's restatement of src/deduplicate_sam.rs:148-176 (orc_stage_reads): keys, N masks, freq,
representative reads and the bucket table bit for bit, in the canonical order."""
import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import umi_collapse_rs_amd as umi
    c = umi.Context(0)
    yield c
    c.close()


ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_reads(rng, n_reads, n_pos, L, n_mol, err=0.03, n_frac=0.0, sorted_file=True):
    """Reads of a molecule model: positions, per position a few molecules, copies with errors."""
    pos = rng.integers(0, n_pos, n_reads)
    if sorted_file:
        pos = np.sort(pos)
    mol = rng.integers(0, n_mol, n_reads)
    centres = rng.choice(ALPHA, (n_pos, n_mol, L))
    umi = centres[pos, mol].copy()
    flip = rng.random(umi.shape) < err
    umi[flip] = rng.choice(ALPHA, int(flip.sum()))
    if n_frac:
        isn = rng.random(umi.shape) < n_frac
        umi[isn] = ord("N")
    score = rng.integers(0, 42, n_reads).astype(np.int32)
    return pos, umi.reshape(-1), score


def dense_ids(pos):
    """Alignment-key ids numbered by first appearance (what orc_stage_reads is given)."""
    _, first, inv = np.unique(pos, return_index=True, return_inverse=True)
    order = np.argsort(np.argsort(first))
    return order[inv].astype(np.uint32)


def compare(got, want):
    for f in ("bucket_off", "keys", "nmask", "freq", "rep"):
        assert len(got[f]) == len(want[f]), (f, len(got[f]), len(want[f]))
        assert (got[f] == want[f]).all(), (f, np.nonzero(got[f] != want[f])[0][:5])


@pytest.mark.parametrize("n_reads,n_pos,L,n_mol,n_frac,merge,sorted_file", [
    (20000, 300, 12, 8, 0.0, 1, True),      # BASELINE config 1 shape
    (20000, 300, 12, 8, 0.01, 1, False),    # positions interleaved in the file, N bases
    (50000, 1, 10, 3000, 0.0, 1, True),     # one deep position
    (30000, 5000, 20, 2, 0.002, 0, False),  # 20-bp UMIs, merge "any", many tiny positions
    (4000, 40, 21, 5, 0.0, 1, True),        # the longest one-word UMI
    (100, 100, 5, 1, 0.0, 1, True),
    (1, 1, 12, 1, 0.0, 1, True),
])
def test_staging_against_the_oracle(ctx, n_reads, n_pos, L, n_mol, n_frac, merge, sorted_file):
    rng = np.random.default_rng(n_reads + 7 * L + merge)
    pos, umi, score = make_reads(rng, n_reads, n_pos, L, n_mol, n_frac=n_frac, sorted_file=sorted_file)
    want = orc.stage_reads(dense_ids(pos), umi, score, L, merge)
    # the library takes any injective 64-bit key: the raw positions, scattered over all 64 bits
    key = (pos.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
    compare(ctx.stage_reads(key, umi, score, L, merge), want)
    # ... or dense ids with the bit count that covers them
    ids = dense_ids(pos).astype(np.uint64)
    compare(ctx.stage_reads(ids, umi, score, L, merge, align_key_bits=max(1, int(ids.max()).bit_length())), want)
    if merge:  # equal scores: the first read of a UMI stands for it (merge/mod.rs:35: >= keeps the earlier)
        flat = np.full(n_reads, 30, np.int32)
        compare(ctx.stage_reads(key, umi, flat, L, 1), orc.stage_reads(dense_ids(pos), umi, flat, L, 1))
        compare(ctx.stage_reads(key, umi, None, L, 1), orc.stage_reads(dense_ids(pos), umi, None, L, 1))


def test_staging_feeds_the_hot_path(ctx):
    """stage -> dedup on the staged arrays == oracle staging -> oracle dedup: the survivors'
    representative reads, in output order."""
    rng = np.random.default_rng(99)
    pos, umi, score = make_reads(rng, 60000, 200, 12, 30, err=0.02)
    st = ctx.stage_reads(pos.astype(np.uint64), umi, score, 12, 1)
    kept, _, _ = ctx.dedup_batch(st["keys"], None, st["freq"], st["bucket_off"], 12, k=1)
    ost = orc.stage_reads(dense_ids(pos), umi, score, 12, 1)
    okept, _, _ = orc.dedup_batch(ost["keys"], None, ost["freq"], ost["bucket_off"], 12, 1)
    assert (st["rep"][kept.astype(bool)] == ost["rep"][okept.astype(bool)]).all()


def test_staging_rejects_what_the_reference_panics_on(ctx):
    import umi_collapse_rs_amd as umi
    umis = np.frombuffer(b"ACGTACGTACGTACGTACGTacgt", dtype=np.uint8)  # lowercase: utils/mod.rs:77-79
    with pytest.raises(umi.UmiHipError):
        ctx.stage_reads(np.zeros(2, np.uint64), umis, None, 12)
    st = ctx.stage_reads(np.zeros(0, np.uint64), np.zeros(0, np.uint8), None, 12)  # no reads at all
    assert len(st["keys"]) == 0 and st["bucket_off"].tolist() == [0]


def test_device_form_runs_on_the_callers_stream_default_stream_included(ctx):
    """umi_stage_reads_device with hip_stream = NULL works on the default stream like every other
    device-pointer call: inputs filled by kernels enqueued on the default stream just before the call,
    with no synchronisation in between, are what it stages (a non-blocking private stream would not
    be ordered behind them).  A long fill chain makes the race wide: the inputs only reach their final
    values after ~100 passes over 4 M reads."""
    import torch
    rng = np.random.default_rng(5)
    n, L = 4_000_000, 12
    pos, umi, score = make_reads(rng, n, 5000, L, 6)
    dev = torch.device("cuda", 0)
    h_key = torch.from_numpy(pos.astype(np.int64))
    d_key = torch.zeros(n, dtype=torch.int64, device=dev)
    d_umi = torch.from_numpy(umi.copy()).to(dev)
    d_score = torch.from_numpy(score).to(dev)
    final = h_key.to(dev)
    outs = [torch.zeros(n, dtype=torch.int64, device=dev) for _ in range(3)]  # keys, nmask, rep
    d_freq = torch.zeros(n, dtype=torch.int32, device=dev)
    d_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    assert torch.cuda.current_stream().cuda_stream == 0
    for i in range(100):  # on the default stream, nothing waited for
        d_key.add_(1)
    d_key.copy_(final)
    ne, nb = ctx.stage_reads_device(d_key.data_ptr(), d_umi.data_ptr(), d_score.data_ptr(), n, L,
                                    outs[0].data_ptr(), outs[1].data_ptr(), d_freq.data_ptr(), outs[2].data_ptr(),
                                    d_off.data_ptr(), merge=1, align_key_bits=64, stream=0)
    torch.cuda.synchronize()
    want = orc.stage_reads(dense_ids(pos), umi, score, L, 1)
    assert ne == len(want["keys"]) and nb == len(want["bucket_off"]) - 1
    assert (outs[0][:ne].cpu().numpy().view(np.uint64) == want["keys"]).all()
    assert (d_freq[:ne].cpu().numpy() == want["freq"]).all()
    assert (outs[2][:ne].cpu().numpy().view(np.uint64) == want["rep"]).all()
    assert (d_off[:nb + 1].cpu().numpy().view(np.uint64) == want["bucket_off"]).all()


@pytest.mark.parametrize("L,n_reads,n_pos,n_mol,n_frac,merge,sorted_file", [
    (24, 30000, 400, 10, 0.0, 1, True),     # dual 12 + 12 UMIs: two words
    (22, 20000, 1, 4000, 0.01, 0, False),   # one deep position, the straddling base 21 (may be N)
    (45, 20000, 3000, 3, 0.002, 1, False),  # three words, positions interleaved
    (85, 5000, 50, 20, 0.0, 1, True),       # four words
    (12, 20000, 300, 8, 0.01, 1, False),    # one word, against the same model
])
def test_staging_of_umis_of_any_length_against_the_definition(ctx, L, n_reads, n_pos, n_mol, n_frac, merge, sorted_file):
    """umi_stage_reads_wide against a plain-Python model of deduplicate_sam.rs:148-176 (a dict per
    position) with the canonical order: keys (all words), N masks, freq, representative reads, table."""
    from helpers import stage_model
    rng = np.random.default_rng(31 * L + merge)
    pos, umi, score = make_reads(rng, n_reads, n_pos, L, n_mol, n_frac=n_frac, sorted_file=sorted_file)
    strings = [bytes(umi[i * L:(i + 1) * L]).decode() for i in range(n_reads)]
    w_umis, w_freq, w_rep, w_off = stage_model(pos, strings, score, merge)
    words = (3 * L + 63) // 64
    wk, wm = orc.encode_keys_wide(w_umis) if words > 1 else tuple(x.reshape(-1, 1) for x in orc.encode_keys(w_umis))
    key = (pos.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)
    for akey, bits in ((key, 64), (dense_ids(pos).astype(np.uint64), max(1, int(dense_ids(pos).max()).bit_length()))):
        got = ctx.stage_reads_wide(akey, umi, score, L, merge, align_key_bits=bits)
        assert (got["bucket_off"] == w_off).all()
        assert (got["keys"] == wk).all() and (got["nmask"] == wm).all()
        assert (got["freq"] == w_freq).all() and (got["rep"] == w_rep).all()
    # ... and what it feeds: the batched call on the staged arrays against the oracle on the model's
    st = got
    kept, root, _ = ctx.dedup_batch_wide(st["keys"], st["nmask"] if st["nmask"].any() else None, st["freq"],
                                         st["bucket_off"], L, k=1) if words > 1 else ctx.dedup_batch(
        st["keys"][:, 0], st["nmask"][:, 0] if st["nmask"].any() else None, st["freq"], st["bucket_off"], L, k=1)
    if words > 1:
        okept, oroot, _ = orc.dedup_batch_wide(wk, wm, w_freq, w_off, L, 1)
    else:
        okept, oroot, _ = orc.dedup_batch(wk[:, 0], wm[:, 0], w_freq, w_off, L, 1)
    assert (kept == okept).all() and (root == oroot).all()


@pytest.mark.parametrize("L,n_reads,shift,bits", [(12, 70_001, 1, 20), (12, 70_001, 0, 37), (9, 3_000, 3, 64), (16, 513, 2, 12),
                                                   (12, 255, 0, 5), (7, 2, 1, 1)])
def test_device_form_with_text_off_the_word_boundary_and_any_key_width(ctx, L, n_reads, shift, bits):
    """The encode kernel takes the UMI text as whole words where it starts on a 4-byte boundary and byte by
    byte where it does not (and for the file's last, partial chunk of 256 reads); the reads are sorted on one
    composed key where align_key_bits + the packed UMI fit 64 bits (bits = 20, 12, 5, 1 here; 37 + 28 and 64
    do not: a sort per key word).  Same entries every way."""
    import torch
    rng = np.random.default_rng(31 * L + n_reads + shift)
    n_pos = max(1, min(n_reads // 20, (1 << min(bits, 20)) - 1))
    pos, umi, score = make_reads(rng, n_reads, n_pos, L, 4, n_frac=0.004, sorted_file=False)
    dev = torch.device("cuda", 0)
    d_key = torch.from_numpy(pos.astype(np.int64)).to(dev)
    raw = torch.zeros(len(umi) + 8, dtype=torch.uint8, device=dev)
    raw[shift:shift + len(umi)] = torch.from_numpy(umi.copy()).to(dev)
    d_score = torch.from_numpy(score).to(dev)
    outs = [torch.zeros(n_reads, dtype=torch.int64, device=dev) for _ in range(3)]  # keys, nmask, rep
    d_freq = torch.zeros(n_reads, dtype=torch.int32, device=dev)
    d_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=dev)
    ne, nb = ctx.stage_reads_device(d_key.data_ptr(), raw.data_ptr() + shift, d_score.data_ptr(), n_reads, L,
                                    outs[0].data_ptr(), outs[1].data_ptr(), d_freq.data_ptr(), outs[2].data_ptr(),
                                    d_off.data_ptr(), merge=1, align_key_bits=bits,
                                    stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = orc.stage_reads(dense_ids(pos), umi, score, L, 1)
    assert ne == len(want["keys"]) and nb == len(want["bucket_off"]) - 1
    assert (outs[0][:ne].cpu().numpy().view(np.uint64) == want["keys"]).all()
    assert (outs[1][:ne].cpu().numpy().view(np.uint64) == want["nmask"]).all()
    assert (d_freq[:ne].cpu().numpy() == want["freq"]).all()
    assert (outs[2][:ne].cpu().numpy().view(np.uint64) == want["rep"]).all()
    assert (d_off[:nb + 1].cpu().numpy().view(np.uint64) == want["bucket_off"]).all()


def test_many_tiny_positions_and_few_deep_ones_take_the_sort(ctx):
    """Both ways to the canonical order on one context: positions of some tens of entries are ordered by a wave
    each; a file of (nearly) singleton positions, and one with a position beyond 1,024 entries, go through the
    order sort."""
    rng = np.random.default_rng(99)
    for n_reads, n_pos, n_mol in ((40000, 500, 30), (40000, 30000, 1), (40000, 4, 4000), (40000, 500, 30)):
        pos, umi, score = make_reads(rng, n_reads, n_pos, 12, n_mol, n_frac=0.002, sorted_file=False)
        compare(ctx.stage_reads(pos.astype(np.uint64), umi, score, 12, merge=1), orc.stage_reads(dense_ids(pos), umi, score, 12, 1))


def test_local_order_with_64_bit_order_keys():
    """Positions of up to 1,024 entries are ordered by a wave each (no sort): with 32-bit order keys where the
    freq field and the read index fit them together, else with 64-bit ones -- forced here through the
    environment (it takes 2^24 reads otherwise), in a process of its own: the library reads it once."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oracle as orc, umi_collapse_rs_amd as umi\n"
        "import test_gpu_stage as t\n"
        "rng = np.random.default_rng(77)\n"
        "c = umi.Context(0)\n"
        "for n_reads, n_pos, L, n_mol, merge, srt in ((30000, 400, 12, 9, 1, False), (5000, 3, 10, 300, 0, True), (70, 70, 7, 1, 1, True)):\n"
        "    pos, u, score = t.make_reads(rng, n_reads, n_pos, L, n_mol, n_frac=0.003, sorted_file=srt)\n"
        "    got = c.stage_reads(pos.astype(np.uint64), u, score, L, merge=merge)\n"
        "    t.compare(got, orc.stage_reads(t.dense_ids(pos), u, score, L, merge))\n"
        "c.close(); print('ok')\n"
    ) % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, UMIHIP_STAGE_WIDE_ORDER="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
