#!/usr/bin/env python3
"""Condenses gpurun_out/refresh/ (the rocprofv3 output directories of tools/refresh_profiles.sh) into the small files kept under
profiles/ (see profiles/README.md).  refresh_profiles.sh runs it on the GPU box with PM_PROFILE_OUT=gpurun_out/profiles_new (the raw
directories are far bigger than what gpurun carries back) and deletes the raw output; copy gpurun_out/profiles_new/* into
profiles/ afterwards.  Round tag as first argument (default r03)."""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "refresh")
P = os.environ.get("PM_PROFILE_OUT") or os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
tool = os.path.join(ROOT, "tools", "summarize_prof.py")
if os.path.exists(os.path.join(O, "bench_long.json")):  # (round 5: the compact line the driver reads, and the long form beside it)
    bench = json.load(open(os.path.join(O, "bench_long.json")))
    shutil.copy(os.path.join(O, "bench.json"), os.path.join(P, "%s_bench_default.json" % TAG))
    shutil.copy(os.path.join(O, "bench_long.json"), os.path.join(P, "%s_bench_long.json" % TAG))
    shutil.copy(os.path.join(O, "bench.json"), os.path.join(P, "bench.json"))
else:  # a subset run (PM_REFRESH_ONLY): the long form of the last full run names the workloads
    bench = json.load(open(os.path.join(ROOT, "profiles", "%s_bench_long.json" % TAG)))
table_path = os.path.join(P, "pmc_traffic.json")
old_table = os.path.join(ROOT, "profiles", "pmc_traffic.json")
table = json.load(open(old_table)) if os.path.exists(old_table) else {}
keep = ("FETCH_SIZE", "WRITE_SIZE")
old_sq = os.path.join(ROOT, "profiles", "%s_sq_counters.json" % TAG)
sq_all = json.load(open(old_sq)) if os.path.exists(old_sq) else {}  # a subset run replaces only its own workloads
keys = {"ns": "dp:ns:%d:%d:%d" % (bench["config"]["pairs_per_rank"], bench["config"]["rows"], bench["config"]["columns"]),
        "c1": "dp:c1:10000:2:1000", "c2": "dp:c2:100000:4:0", "deep": "dp:deep:4096:32:10000", "tr": "translate:4:1000000:2500:16:6000"}
for name, key in keys.items():
    if not os.path.isdir(os.path.join(O, name + "_stats")):
        continue
    subprocess.run([sys.executable, tool, "stats", os.path.join(O, name + "_stats"), os.path.join(P, "%s_%s_kernel_stats.csv" % (TAG, name))], check=True)
    subprocess.run([sys.executable, tool, "pmc", "/tmp/pmc_%s.json" % name, "fetch=" + os.path.join(O, name + "_fetch"),
                    "write=" + os.path.join(O, name + "_write"), "sq=" + os.path.join(O, name + "_sq")], check=True)
    allk = json.load(open("/tmp/pmc_%s.json" % name))
    table[key] = {k: {c: v for c, v in x.items() if c in keep} for k, x in allk.items()
                  if any(t in k for t in ("dp_", "translate_", "scatter_live", "rocprim"))}
    sq_all[key] = {k: {c: round(v["mean"]) for c, v in x.items() if c.startswith("SQ_")} for k, x in allk.items()
                   if any(t in k for t in ("dp_fill", "dp_walk", "translate_"))}
table = {k: v for k, v in table.items() if k in keys.values()}  # no entries of workloads (or rounds) that are not measured any more
table["_about"] = ("per-kernel means over the dispatches of `rocprofv3 --pmc FETCH_SIZE` and `rocprofv3 --pmc WRITE_SIZE` (separate runs of "
                   "`python3 bench.py <workload> --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end`, one workload per run: tools/refresh_profiles.sh); "
                   "units: KB as rocprofv3 reports; bench.py applies the gfx950 x2 correction to FETCH_SIZE")
json.dump(table, open(table_path, "w"), indent=1)
json.dump(sq_all, open(os.path.join(P, "%s_sq_counters.json" % TAG), "w"), indent=1)
print("DP   %.1f GCUPS  %.3f ms/step  %s" % (bench["value"], bench["ms_per_step"], bench["kernel_ms"]))
for ride in ("c1", "c2", "deep"):
    if ride in bench:
        print("%-4s %.1f GCUPS  %.3f ms/step  %s" % (ride, bench[ride]["value"], bench[ride]["ms_per_step"], bench[ride]["kernel_ms"]))
t = bench.get("translate")
if t:
    print("TR   %.3g units/s  %.3f ms/step  %s  frac %.3f" % (t["value"], t["ms_per_step"], t["kernel_ms"], t["roofline"]["frac"]))
for name in keys:
    f = os.path.join(P, "%s_%s_kernel_stats.csv" % (TAG, name))
    if os.path.exists(f):
        print(name)
        for r in csv.DictReader(open(f)):
            if float(r["Percentage"]) > 0.3:
                print("   %-60s %4s x %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
