"""Where the time of a many-small-buckets call goes: wall time of the device entry point against
the library's own event times (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import umi_collapse_rs_amd as umi
from umi_collapse_rs_amd import synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
L, k = (20, 2) if cfg == "5" else (12, 1)
reads = {"3": 10_000_000, "4": 12_500_000, "5": 6_250_000}[cfg]
st = synth.config3(seed=int(cfg), n_reads=reads, n_positions=reads // 100, umi_len=L)
dev = torch.device("cuda", 0)
dk = torch.from_numpy(st["keys"].view(np.int64)).to(dev); df = torch.from_numpy(st["freq"]).to(dev)
n = len(st["keys"]); kept = torch.zeros(n, dtype=torch.uint8, device=dev)
db = torch.from_numpy(np.ascontiguousarray(st["bucket_off"]).view(np.int64)).to(dev)
ctx = umi.Context(0, profile=True)
s = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    ctx.dedup_batch_device(dk.data_ptr(), 0, df.data_ptr(), st["bucket_off"], L, kept.data_ptr(), 0, k=k, stream=s, d_bucket_off=(db.data_ptr() if len(sys.argv) > 2 else 0))
ts = []; ev = []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ctx.dedup_batch_device(dk.data_ptr(), 0, df.data_ptr(), st["bucket_off"], L, kept.data_ptr(), 0, k=k, stream=s, d_bucket_off=(db.data_ptr() if len(sys.argv) > 2 else 0))
    ts.append(time.perf_counter() - t0); ev.append((r["ms_prep"], r["ms_pairs"], r["ms_collapse"], r["ms_finalize"], r["ms_total"]))
print("config", cfg, "entries", n, "buckets", len(st["bucket_off"]) - 1)
print("wall ms/call: median %.4f min %.4f" % (np.median(ts) * 1e3, min(ts) * 1e3))
print("events ms (prep, pairs, collapse, finalize, total):", np.round(np.median(np.array(ev), axis=0), 4))
