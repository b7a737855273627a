#!/usr/bin/env python3
"""Reads the kernel trace of one bench run (rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --path dp
--config c2 --no-ride-alongs --steps 3 --warmup 1 ...) and prints, for the last step, the timeline of the fill and path kernels:
every launch's start and duration, the idle time of the fill stream between consecutive fill launches, and how long the last
path kernel runs after the last fill kernel has ended.  python tools/kernel_timeline.py DIR"""
import csv
import glob
import os
import sys

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not files:
    sys.exit("no kernel_trace.csv under " + d)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        kind = "fill" if "dp_fill_kernel" in name else ("path" if ("dp_walk_kernel" in name or "dp_traceback_kernel" in name) else None)
        if kind:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
rows.sort()
if not rows:
    sys.exit("no DP kernels in the trace")
# steps: a gap of more than 2 ms with nothing running separates them (the bench synchronises between steps)
steps, cur, busy_to = [], [], None
for s, e, k in rows:
    if busy_to is not None and s - busy_to > 2_000_000 and cur:
        steps.append(cur)
        cur = []
    cur.append((s, e, k))
    busy_to = max(busy_to or 0, e)
steps.append(cur)
last = steps[-1]
t0 = last[0][0]
fills = [x for x in last if x[2] == "fill"]
paths = [x for x in last if x[2] == "path"]
print("steps seen:", len(steps), " launches in the last one: fill", len(fills), "path", len(paths))
print("step, first start to last end: %.3f ms" % ((max(e for _, e, _ in last) - t0) / 1e6))
print("fill kernels: sum %.3f ms; path kernels: sum %.3f ms" % (sum(e - s for s, e, _ in fills) / 1e6, sum(e - s for s, e, _ in paths) / 1e6))
gaps = [(fills[k + 1][0] - fills[k][1]) / 1e6 for k in range(len(fills) - 1)]
print("idle between consecutive fill launches: sum %.3f ms, max %.3f ms" % (sum(gaps), max(gaps) if gaps else 0))
print("last path kernel ends %.3f ms after the last fill kernel" % ((max(e for _, e, _ in paths) - fills[-1][1]) / 1e6 if paths else 0))
for s, e, k in last:
    print("  %-4s start %9.3f ms  duration %8.3f ms" % (k, (s - t0) / 1e6, (e - s) / 1e6))
