#!/usr/bin/env python3
"""After `gpurun -- bash tools/refresh_profiles.sh`: condense gpurun_out/refresh/ into profiles/ (see profiles/README.md)."""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "refresh")
P = os.path.join(ROOT, "profiles")
tool = os.path.join(ROOT, "tools", "summarize_prof.py")
subprocess.run([sys.executable, tool, "stats", os.path.join(O, "stats"), os.path.join(P, "r01_bench_kernel_stats.csv")], check=True)
subprocess.run([sys.executable, tool, "pmc", "/tmp/pmc_all.json", "fetch=" + os.path.join(O, "fetch"), "write=" + os.path.join(O, "write"),
                "sq=" + os.path.join(O, "sq")], check=True)
allk = json.load(open("/tmp/pmc_all.json"))
keep = ("FETCH_SIZE", "WRITE_SIZE")
dp = {k: {c: v for c, v in x.items() if c in keep} for k, x in allk.items() if "dp_" in k}
tr = {k: {c: v for c, v in x.items() if c in keep} for k, x in allk.items() if "translate_" in k or "scatter_live" in k or "rocprim" in k.lower()}
bench = json.load(open(os.path.join(O, "bench.json")))
cfg = bench["config"]
table = {"dp:%d:%d:%d" % (cfg["pairs_per_rank"], cfg["rows"], cfg["columns"]): dp, "translate:4:1000000:2500:16:6000": tr,
         "_about": "per-kernel means over the dispatches of `rocprofv3 --pmc FETCH_SIZE` and `rocprofv3 --pmc WRITE_SIZE` (two separate runs of "
                   "`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline`); units: KB as rocprofv3 reports; bench.py applies the gfx950 x2 "
                   "correction to FETCH_SIZE"}
json.dump(table, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
sq = {k: {c: round(v["mean"]) for c, v in x.items() if c.startswith("SQ_")} for k, x in allk.items() if "dp_fill" in k or "translate_" in k}
json.dump(sq, open(os.path.join(P, "r01_sq_counters.json"), "w"), indent=1)
shutil.copy(os.path.join(O, "bench.json"), os.path.join(P, "r01_bench_default.json"))
print("DP   %.1f GCUPS  %.3f ms/step  %s" % (bench["value"], bench["ms_per_step"], bench["kernel_ms"]))
t = bench["translate"]
print("TR   %.3g units/s  %.3f ms/step  %s  frac %.3f  traffic %s" % (t["value"], t["ms_per_step"], t["kernel_ms"], t["roofline"]["frac"], t["roofline"]["traffic"]))
print("CLI ", t.get("cli_whole_job"), t.get("cpu_baseline", {}).get("value"))
for r in csv.DictReader(open(os.path.join(P, "r01_bench_kernel_stats.csv"))):
    if float(r["Percentage"]) > 0.3:
        print("   %-50s %4s x %8.1f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
