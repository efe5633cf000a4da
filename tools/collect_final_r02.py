#!/usr/bin/env python3
"""After tools/final_r02.sh has run on the GPU box: everything under gpurun_out/ that is kept goes to
profiles/ -- the config-2 counters json (tools/summarize_r02.py), the bench lines of every config,
the config-3 kernel statistics and counters."""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", "final")
P = os.path.join(ROOT, "profiles")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_r02.py"), "head"])


def last_json_line(path):
    return [l for l in open(path) if l.startswith("{")][-1]


for c in ("2m", "3", "4", "5"):
    open(os.path.join(P, "r02_config%s_bench_line.json" % c), "w").write(last_json_line(os.path.join(F, "bench_config%s.log" % c)))
open(os.path.join(P, "r02_config2_bench_line_full.json"), "w").write(last_json_line(os.path.join(F, "bench_config2.log")))
open(os.path.join(P, "r02_config2_bench_line_2ranks_gloo_one_gpu.json"), "w").write(
    last_json_line(os.path.join(F, "bench_2ranks_gloo.log")))
C3 = os.path.join(ROOT, "gpurun_out", "prof_c3")
shutil.copy(glob.glob(os.path.join(C3, "trace", "*kernel_stats.csv"))[0], os.path.join(P, "r02_config3_kernel_stats.csv"))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("insts", "fetch", "write"):
    for f in glob.glob(os.path.join(C3, sub, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("umihip::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = sorted({c for d in acc.values() for c in d})
with open(os.path.join(P, "r02_config3_counters.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel"] + cols)
    for n, d in acc.items():
        if n.startswith("at::") or n.startswith("__amd"):
            continue
        w.writerow([n] + ["%.4g" % (sum(d[c]) / len(d[c])) if c in d else "" for c in cols])
print("profiles/ updated")
