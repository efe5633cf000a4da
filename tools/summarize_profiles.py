#!/usr/bin/env python3
"""Condense the outputs of tools/profile_config2.sh and tools/pmc_pairs.sh (under gpurun_out/)
into the small files kept in profiles/: usage  python3 tools/summarize_profiles.py <tag>"""
import collections
import csv
import glob
import json
import shutil
import sys

tag = sys.argv[1]
P = "gpurun_out/prof_config2"
shutil.copy(glob.glob(P + "/trace/*kernel_stats.csv")[0], "profiles/r01_config2_kernel_stats_%s.csv" % tag)
for src, dst in (("bench_under_rocprof.log", "bench_under_rocprof"), ("bench_plain.log", "bench_line")):
    line = [l for l in open("%s/%s" % (P, src)) if l.startswith("{")][-1]
    json.loads(line)
    open("profiles/r01_config2_%s_%s.json" % (dst, tag), "w").write(line)

rows = []
for counter, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob("%s/%s/*counter_collection.csv" % (P, d))[0])):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        rows.append((counter, k[:150], len(v), sum(v) / len(v)))
with open("profiles/r01_config2_pmc_fetch_write_%s.csv" % tag, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["counter", "kernel", "dispatches", "avg_KB_per_dispatch"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], "%.1f" % r[3]])

acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_pu/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ("bs_tab_kernel", "bs_run_kernel", "tab_scan_kernel")):
            acc[(r["Kernel_Name"][:150], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open("profiles/r01_config2_sq_counters_%s.csv" % tag, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "avg_per_dispatch"])
    for (k, c), v in sorted(acc.items()):
        w.writerow([k, c, "%.4g" % (sum(v) / len(v))])
print("ok")
