#!/usr/bin/env python3
"""The outputs of tools/final_r03.sh (gpurun_out/final_r03/) into profiles/: the bench lines of the
round's last sources (they carry the counters profiles/r03_counters.json holds for those sources),
the end-to-end split, the full-size record.    python3 tools/collect_final_r03.py"""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final_r03")
DST = os.path.join(ROOT, "profiles")


def line_of(path):
    return [l for l in open(path) if l.startswith("{")][-1]


d = line_of(os.path.join(SRC, "bench_line_default.log"))
open(os.path.join(DST, "r03_bench_line_default.json"), "w").write(d)
open(os.path.join(DST, "r03_config2_bench_line_final.json"), "w").write(d)
j = json.loads(d)
print("default: %.4f ms/step, %.3e %s, roofline frac %.4f (stale counters: %s), end to end %.2f M reads/s" % (
    j["ms_per_step"], j["value"], j["unit"], j["roofline"]["frac"], j["roofline"].get("counters_stale"),
    (j.get("end_to_end") or {}).get("reads_per_s", 0) / 1e6))
for c in ("3", "4", "5", "2m", "wide24"):
    p = os.path.join(SRC, "bench_line_config%s.log" % c)
    if os.path.exists(p):
        l = line_of(p)
        open(os.path.join(DST, "r03_config%s_bench_line_final.json" % c), "w").write(l)
        j = json.loads(l)
        print("config %s: %.4f ms/step, %.3e reads/s, %s %.1f us, frac %.4f" % (
            c, j["ms_per_step"], j["reads_per_s"], j["roofline"]["kernel"], j["roofline"]["kernel_us"], j["roofline"]["frac"]))
shutil.copy(os.path.join(SRC, "end_to_end_split.txt"), os.path.join(DST, "r03_end_to_end_split.txt"))
keep = [l for l in open(os.path.join(SRC, "fullscale.txt")) if "entries in" in l or l.startswith("    ") or "passed" in l]
open(os.path.join(DST, "r03_fullscale_configs_4_5.txt"), "w").write(
    "tests/test_gpu_fullscale.py -s on the 1-GPU box: BASELINE configs 4 and 5 at full size through an 8-worker\n"
    "context (umi.Context([0] * 8)); per worker: gather into pinned memory | the call | scatter back\n\n" + "".join(keep))
