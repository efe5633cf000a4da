"""Device staging at BASELINE config 3's scale: 10,000,000 reads in 100,000 positions (molecule
model), everything resident in HBM; then the hot path on the staged arrays (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import umi_collapse_rs_amd as umi
from umi_collapse_rs_amd import synth
L, n_pos, rpp = 12, 100_000, 100
pos, bases = [], []
for p0 in range(0, n_pos, 10_000):
    p, b = synth.molecule_reads(3, 10_000, rpp, L, first_position=p0)
    pos.append(p); bases.append(b)
pos = np.concatenate(pos); bases = np.concatenate(bases)
n = len(pos)
dev = torch.device("cuda", 0)
d_umi = torch.from_numpy(synth.BASES[bases].reshape(-1).copy()).to(dev)
d_key = torch.from_numpy(pos.astype(np.int64)).to(dev)
d_sc = torch.randint(0, 40, (n,), dtype=torch.int32, device=dev)
o_keys = torch.zeros(n, dtype=torch.int64, device=dev); o_freq = torch.zeros(n, dtype=torch.int32, device=dev)
o_rep = torch.zeros(n, dtype=torch.int64, device=dev); o_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
kept = torch.zeros(n, dtype=torch.uint8, device=dev)
ctx = umi.Context(0)
s = torch.cuda.current_stream().cuda_stream
def once():
    ne, nb = ctx.stage_reads_device(d_key.data_ptr(), d_umi.data_ptr(), d_sc.data_ptr(), n, L, o_keys.data_ptr(), 0,
                                    o_freq.data_ptr(), o_rep.data_ptr(), o_off.data_ptr(), merge=1, align_key_bits=17, stream=s)
    off = o_off[:nb + 1].cpu().numpy().view(np.uint64)
    st = ctx.dedup_batch_device(o_keys.data_ptr(), 0, o_freq.data_ptr(), off, L, kept.data_ptr(), 0, k=1, stream=s,
                                d_bucket_off=o_off.data_ptr())
    return ne, nb, st
ne, nb, st = once()
ref = synth.config3(seed=3, n_reads=n, n_positions=n_pos, umi_len=L)
assert ne == len(ref["keys"]) and nb == n_pos and (o_keys[:ne].cpu().numpy().view(np.uint64) == ref["keys"]).all()
assert (o_freq[:ne].cpu().numpy() == ref["freq"]).all()
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); once(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
t1 = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx.stage_reads_device(d_key.data_ptr(), d_umi.data_ptr(), d_sc.data_ptr(), n, L, o_keys.data_ptr(), 0, o_freq.data_ptr(),
                           o_rep.data_ptr(), o_off.data_ptr(), merge=1, align_key_bits=17, stream=s)
    torch.cuda.synchronize(); t1.append(time.perf_counter() - t0)
print("reads %d -> entries %d in %d positions, kept %d" % (n, ne, nb, st["n_kept"]))
print("staging alone: %.3f ms = %.3g reads/s; staging + table D2H + hot path: %.3f ms = %.3g reads/s" % (
    min(t1) * 1e3, n / min(t1), min(ts) * 1e3, n / min(ts)))
