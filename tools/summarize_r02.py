#!/usr/bin/env python3
"""Condense the outputs of tools/profile_r02.sh (under gpurun_out/prof_r02/) into
profiles/r02_config2_counters.json -- what bench.py's `roofline` reads -- and the small csv
summaries kept beside it:   python3 tools/summarize_r02.py <tag>

The json is stamped with the sha256 of umi_collapse_rs_amd/csrc/* as bench.py computes it;
bench.py prints frac: null, stale: true when the sources have changed since."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (source_sha256)

tag = sys.argv[1] if len(sys.argv) > 1 else "head"
P = os.path.join(ROOT, "gpurun_out", "prof_r02")
STEPS = 7  # --steps 5 --warmup 2 of tools/profile_r02.sh

PHASES = (("prep", ("prep_kernel", "bucket_rise", "seg_scan", "seg_scatter", "seg_count", "seg_hist", "iota", "rocprim",
                    "build_planes", "gather_kernel")),
          ("pairs", ("seg_pair", "seg_edge", "pair_kernel", "bs_pair", "bs_run", "bs_tab", "tab_scan",
                     "verify_list", "small_bucket")),
          ("collapse", ("uf_", "dag_hook", "jump_kernel", "cc_hook", "hook_kernel", "map_label", "adj_mark",
                        "adj_promote")),
          ("finalize", ("map_finalize", "finalize_kernel", "adj_finalize")))


def phase_of(name):
    for ph, keys in PHASES[::-1]:  # (map_finalize before map_label ...)
        if any(k in name for k in keys):
            return ph
    return "other"


def short(name):
    n = name.replace("umihip::(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0]
    return n.strip()[:80]


def counters(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(P, sub, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


stats_csv = glob.glob(os.path.join(P, "trace", "*kernel_stats.csv"))[0]
shutil.copy(stats_csv, os.path.join(ROOT, "profiles", "r02_config2_kernel_stats_%s.csv" % tag))
kernels = {}
for r in csv.DictReader(open(stats_csv)):
    name = short(r["Name"])
    if name.startswith("__amd_rocclr"):
        continue
    kernels[name] = {"avg_us": float(r["AverageNs"]) / 1e3, "calls": int(r["Calls"]),
                     "calls_per_step": int(r["Calls"]) / STEPS, "phase": phase_of(name)}
for sub in ("insts", "cycles", "fetch", "write"):
    for name, d in counters(sub).items():
        if name in kernels:
            for c, v in d.items():
                key = {"FETCH_SIZE": "FETCH_SIZE_KB", "WRITE_SIZE": "WRITE_SIZE_KB"}.get(c, c)
                kernels[name][key] = v
lines = {}
for src, dst in (("bench_under_rocprof.log", "bench_under_rocprof"), ("bench_plain.log", "bench_line")):
    line = [l for l in open(os.path.join(P, src)) if l.startswith("{")][-1]
    lines[dst] = json.loads(line)
    open(os.path.join(ROOT, "profiles", "r02_config2_%s_%s.json" % (dst, tag)), "w").write(line)
sha_file = {}
for l in open(os.path.join(P, "sources.sha256")):
    h, f = l.split()
    sha_file[os.path.basename(f)] = h
out = {
    "round": 2, "tag": tag,
    "workload": lines["bench_line"]["config"]["workload"],
    "command": "bash tools/profile_r02.sh  (rocprofv3 --kernel-trace --stats, then one --pmc pass per "
               "counter set, each over `python3 bench.py --steps 5 --warmup 2 --no-extras`)",
    "source_sha256": bench.source_sha256(),
    "source_files_sha256_on_the_box": sha_file,
    "ms_per_step_plain": lines["bench_line"]["ms_per_step"],
    "phases_ms_plain": lines["bench_line"]["phases_ms"],
    "kernels": kernels,
}
json.dump(out, open(os.path.join(ROOT, "profiles", "r02_config2_counters.json"), "w"), indent=1, sort_keys=True)
with open(os.path.join(ROOT, "profiles", "r02_config2_counters_%s.csv" % tag), "w", newline="") as f:
    w = csv.writer(f)
    cols = ["avg_us", "calls_per_step", "phase", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH",
            "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
            "SQ_ACTIVE_INST_ANY", "SQ_WAVES", "FETCH_SIZE_KB", "WRITE_SIZE_KB"]
    w.writerow(["kernel"] + cols)
    for name, d in sorted(kernels.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls_per_step"]):
        w.writerow([name] + [("%.4g" % d[c] if isinstance(d.get(c), float) else d.get(c, "")) for c in cols])
# the files on the box must be the files here
mine = {os.path.basename(f): __import__("hashlib").sha256(open(f, "rb").read()).hexdigest()
        for f in glob.glob(os.path.join(ROOT, "umi_collapse_rs_amd", "csrc", "*"))}
if mine != sha_file:
    print("WARNING: sources changed since the profile was taken:", sorted(k for k in mine if mine[k] != sha_file.get(k)))
print("ok: %d kernels, step %.3f ms" % (len(kernels), out["ms_per_step_plain"]))
