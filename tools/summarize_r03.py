#!/usr/bin/env python3
"""Condense the outputs of tools/profile_r03.sh (gpurun_out/prof_r03/c<config>/) into
profiles/r03_counters.json -- what bench.py's `roofline.traffic` / `frac_issue` read -- and the
small csv summaries kept beside it:   python3 tools/summarize_r03.py [tag]

Every config's entry is stamped with the sha256 of umi_collapse_rs_amd/csrc/* as bench.py computes
it; bench.py marks the counters stale when the sources have changed since (the fractions it prints
come from durations measured live and do not depend on this file)."""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (source_sha256)

tag = sys.argv[1] if len(sys.argv) > 1 else "head"
BASE = os.path.join(ROOT, "gpurun_out", "prof_r03")
STEPS = 13  # --steps 5 --warmup 2: 2 + 5 plain steps, then 1 + 5 with the event records on (bench.py measure)
OUT = os.path.join(ROOT, "profiles", "r03_counters.json")

PHASES = (("prep", ("prep_kernel", "bucket_rise", "seg_scan", "seg_scatter", "seg_count", "seg_hist", "iota", "rocprim")),
          ("pairs", ("seg_pair", "seg_edge", "pair_kernel", "small_bucket", "wide_")),
          ("collapse", ("uf_", "dag_", "jump_kernel", "map_label", "adj_mark", "adj_promote")),
          ("finalize", ("map_finalize", "finalize_kernel", "adj_finalize")))


def phase_of(name):
    for ph, keys in PHASES[::-1]:  # (map_finalize before map_label ...)
        if any(k in name for k in keys):
            return ph
    return "other"


def short(name):
    n = name.replace("umihip::(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0].strip()[:80]


def counters(P, sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(P, sub, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


doc = json.load(open(OUT)) if os.path.exists(OUT) else {"round": 3, "configs": {}}
mine = {os.path.basename(f): hashlib.sha256(open(f, "rb").read()).hexdigest() for f in bench.stamped_sources()}
for P in sorted(glob.glob(os.path.join(BASE, "c*"))):
    cfg = os.path.basename(P)[1:]
    stats_csv = glob.glob(os.path.join(P, "trace", "*kernel_stats.csv"))
    if not stats_csv:
        continue
    shutil.copy(stats_csv[0], os.path.join(ROOT, "profiles", "r03_config%s_kernel_stats_%s.csv" % (cfg, tag)))
    kernels = {}
    for r in csv.DictReader(open(stats_csv[0])):
        name = short(r["Name"])
        if name.startswith("__amd_rocclr") or name.startswith("at::") :
            continue
        kernels[name] = {"avg_us": float(r["AverageNs"]) / 1e3, "calls": int(r["Calls"]),
                         "calls_per_step": int(r["Calls"]) / STEPS, "phase": phase_of(name)}
    for sub in ("insts", "cycles", "fetch", "write"):
        for name, d in counters(P, sub).items():
            if name in kernels:
                for c, v in d.items():
                    kernels[name][{"FETCH_SIZE": "FETCH_SIZE_KB", "WRITE_SIZE": "WRITE_SIZE_KB"}.get(c, c)] = v
    lines = {}
    for src, dst in (("bench_under_rocprof.log", "bench_under_rocprof"), ("bench_plain.log", "bench_line")):
        line = [l for l in open(os.path.join(P, src)) if l.startswith("{")][-1]
        lines[dst] = json.loads(line)
        if dst != "bench_line":  # (the plain lines kept in profiles/ are tools/final_r03.sh's: taken after this file exists)
            open(os.path.join(ROOT, "profiles", "r03_config%s_%s_%s.json" % (cfg, dst, tag)), "w").write(line)
    sha_file = {}
    for l in open(os.path.join(P, "sources.sha256")):
        h, f = l.split()
        if os.path.basename(f) in mine:  # (the staging's sources are not part of the stamp)
            sha_file[os.path.basename(f)] = h
    if mine != sha_file:
        print("WARNING config %s: sources changed since the profile was taken:" % cfg,
              sorted(k for k in mine if mine[k] != sha_file.get(k)))
    doc["configs"][cfg] = {
        "tag": tag, "workload": lines["bench_line"]["config"]["workload"],
        "command": "bash tools/profile_r03.sh %s  (rocprofv3 --kernel-trace --stats, then one --pmc pass per counter "
                   "set, each over `python3 bench.py --config %s --steps 5 --warmup 2 --no-extras`)" % (cfg, cfg),
        "source_sha256": bench.source_sha256() if mine == sha_file else "sources differ from the box's",
        "ms_per_step_plain": lines["bench_line"]["ms_per_step"],
        "ms_per_step_under_rocprof": lines["bench_under_rocprof"]["ms_per_step"],
        "phases_ms_plain": lines["bench_line"]["phases_ms"],
        "roofline_plain": lines["bench_line"]["roofline"],
        "kernels": kernels,
    }
    with open(os.path.join(ROOT, "profiles", "r03_config%s_counters_%s.csv" % (cfg, tag)), "w", newline="") as f:
        w = csv.writer(f)
        cols = ["avg_us", "calls_per_step", "phase", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH",
                "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                "SQ_ACTIVE_INST_ANY", "SQ_WAVES", "FETCH_SIZE_KB", "WRITE_SIZE_KB"]
        w.writerow(["kernel"] + cols)
        for name, d in sorted(kernels.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls_per_step"]):
            w.writerow([name] + [("%.4g" % d[c] if isinstance(d.get(c), float) else d.get(c, "")) for c in cols])
    ksum = sum(v["avg_us"] * v["calls_per_step"] for v in kernels.values())
    print("config %s: %d kernels, %.1f launches per step, kernel sum %.1f us, step %.3f ms plain" % (
        cfg, len(kernels), sum(v["calls_per_step"] for v in kernels.values()), ksum, lines["bench_line"]["ms_per_step"]))
json.dump(doc, open(OUT, "w"), indent=1, sort_keys=True)
