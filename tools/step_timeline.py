#!/usr/bin/env python3
"""tools/step_timeline.py -- the kernels of ONE headline step in launch order with durations and gaps, from a
rocprofv3 --kernel-trace of `bench.py --child --steps 4` (run on the GPU box)."""
import csv
import glob
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = tempfile.mkdtemp(prefix="pgv_tl_", dir="/tmp")
wl = sys.argv[1] if len(sys.argv) > 1 else "headline"
subprocess.run(["rocprofv3", "--kernel-trace", "-d", d, "-o", "t", "--output-format", "csv", "--", sys.executable,
                os.path.join(ROOT, "bench.py"), "--child", "--workload", wl, "--steps", "4", "--warmup", "2", "--settle-ms", "0",
                "--overlap-lanes", "0"], cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
# the last big scan launch ends the last step; the step starts after the previous big scan's tail
big = [i for i, r in enumerate(rows) if "mfma_scan_kernel" in r[2] and r[1] - r[0] > 400000]
lo, hi = big[-2], big[-1]
# a step = from the first kernel after the previous step's last kernel ... find the center-ranking scan before `hi`
start = max(i for i in range(lo, hi) if "mfma_scan_kernel" in rows[i][2] and i != lo)   # the ranking launch of the last step
while start > lo + 1 and rows[start][0] - rows[start - 1][1] < 20000 and "mfma_scan" not in rows[start - 1][2]:
    start -= 1
end = hi
while end + 1 < len(rows) and rows[end + 1][0] - rows[end][1] < 50000:
    end += 1
prev_end = rows[start - 1][1]
t0 = rows[start][0]
print("%-60s %10s %10s %8s" % ("kernel", "start us", "dur us", "gap us"))
last = None
tot = 0
for s, e, name in rows[start:end + 1]:
    name = name.replace("void ", "").replace("pgv::(anonymous namespace)::", "").split("(")[0][:58]
    gap = (s - last) / 1e3 if last is not None else 0.0
    print("%-60s %10.1f %10.1f %8.1f" % (name, (s - t0) / 1e3, (e - s) / 1e3, gap))
    last = e
    tot += e - s
print("step: %.1f us from first start to last end, %.1f us of kernels, %d launches" % ((rows[end][1] - t0) / 1e3, tot / 1e3, end - start + 1))
