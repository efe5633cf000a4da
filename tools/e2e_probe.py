#!/usr/bin/env python3
"""Where the end-to-end run's time goes (on the GPU box): the program's own split, the wall clock of
the process, and what lies outside the program's clock (loader before main, teardown after _Exit).
    python3 tools/e2e_probe.py [reads] [positions]"""
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "umi_collapse_rs_amd", "bin", "umicollapse")
tmp = os.environ.get("TMPDIR", "/tmp")
src, dst = os.path.join(tmp, "e2e_in.bam"), os.path.join(tmp, "e2e_out.bam")
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
positions = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_bam.py"), src, "--reads", str(reads),
                       "--positions", str(positions)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
for f in ("enabled", "defrag"):
    try:
        print("transparent_hugepage/%s: %s" % (f, open("/sys/kernel/mm/transparent_hugepage/" + f).read().strip()))
    except OSError as e:
        print(e)
print("input: %d reads, %d positions, %.1f MB" % (reads, positions, os.path.getsize(src) / 1e6))


def run(extra, env=None, label=""):
    e = dict(os.environ, UMICOLLAPSE_CLOCK="1")
    e.update(env or {})
    time.sleep(1.0)  # (the driver puts the context of the run before away in the background: ~0.12 s of the next run's start)
    t0 = time.time()
    r = subprocess.run([CLI, "-i", src, "-o", dst, "--merge", "avgqual", "--num-threads", "16"] + extra,
                       capture_output=True, text=True, env=e)
    t1 = time.time()
    m = re.search(r"clock: main at ([0-9.]+), exit at ([0-9.]+)", r.stderr)
    ph = [l for l in r.stderr.splitlines() if l.startswith("phases:")]
    a, b = (float(m.group(1)), float(m.group(2))) if m else (t0, t1)
    print("%-22s wall %.3f s = before main %.3f + program %.3f + after exit %.3f   (%.2f M reads/s)" %
          (label, t1 - t0, a - t0, b - a, t1 - b, reads / (t1 - t0) / 1e6))
    if ph:
        print("   ", ph[0])
    for l in r.stderr.splitlines():
        if l.startswith("laps:"):
            print("   ", l)
    return r


for i in range(3):
    run([], label="gpu staging, run %d" % (i + 1))
run(["--stage", "host"], label="host staging")
run(["--compress-level", "0"], label="no compression")
os.unlink(src)
if os.path.exists(dst):
    os.unlink(dst)
