"""Debug helper: replay one seed of tests/test_gpu_fuzz.py under several option sets and say
where the output differs from the oracle (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import oracle as orc
import umi_collapse_rs_amd as umi
import test_gpu_fuzz as tf

seed = int(sys.argv[1])
rng = np.random.default_rng(9000 + seed)
L = int(rng.choice([4, 6, 8, 9, 11, 12, 13, 16, 17, 20, 21]))
k = int(rng.choice([0, 1, 1, 1, 2, 2, 3, 4]))
p = float(rng.choice([0.5, 0.5, 0.3, 0.75, 1.0, 0.0]))
algo, amf = (0, 0) if rng.random() < 0.75 else (1, int(rng.choice([0, 1, 3])))
n_frac = float(rng.choice([0.0, 0.0, 0.02, 0.2]))
src = open(tf.__file__).read()
sizes_line = [l for l in src.split("\n") if "sizes = [" in l][0]
i = src.index("sizes = [")
j = src.index("cap = 4 ** L // 2")
exec(src[i:j].replace("\n    ", "\n"))
cap = 4 ** L // 2
keys, nm, fr, off = [], [], [], [0]
for n_target in sizes:
    umis, freq = tf.clustered_bucket(rng, min(n_target, cap), L, n_frac) if n_target else ([], [])
    kk, mm = orc.encode_keys(umis)
    keys.append(kk); nm.append(mm); fr.extend(freq); off.append(off[-1] + len(umis))
keys, nm = np.concatenate(keys), np.concatenate(nm)
fr, off = np.array(fr, np.int32), np.array(off, np.uint64)
print("seed", seed, "L", L, "k", k, "p", p, "algo", algo, "amf", amf, "n_frac", n_frac, "sizes", np.diff(off.astype(np.int64)).tolist())
okept, oroot, _ = orc.dedup_batch(keys, nm, fr, off, L, k, p, algo, amf)
for opts in ({}, {"seg_index": 0}, {"two_phase": 1}, {"seg_index": 0, "two_phase": 1}, {"fused_max": 0}, {"seg_min": 130}):
    ctx = umi.Context(0)
    for name, v in opts.items():
        ctx.set_option(name, v)
    kept, root, st = ctx.dedup_batch(keys, nm if nm.any() else None, fr, off, L, k, p, algo, amf)
    ctx.close()
    bad = np.nonzero((kept != okept) | (root != oroot))[0]
    print(opts, "mismatches", len(bad), "edges", st["n_edges"], "cand", st["n_candidates"], "rounds", st["n_rounds"], "eval", st["n_pairs_evaluated"])
    if len(bad):
        b = np.searchsorted(off, bad[:8], side="right") - 1
        print("   at", bad[:8].tolist(), "buckets", b.tolist(), "sizes", (off[b + 1] - off[b]).tolist(),
              "kept", kept[bad[:8]].tolist(), "okept", okept[bad[:8]].tolist(), "root", root[bad[:8]].tolist(), "oroot", oroot[bad[:8]].tolist())
