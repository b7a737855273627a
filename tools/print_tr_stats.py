#!/usr/bin/env python3
"""Prints the translate kernels' average durations from a rocprofv3 --kernel-trace --stats directory (argument), or, without an argument,
the translate leg of gpurun_out/bench_long.json."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if any(x in r["Name"] for x in ("translate_", "count_", "flag_", "expand_")):
                print("%-72s calls %5s avg %9.1f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3))
else:
    d = json.load(open(os.path.join(ROOT, "gpurun_out", "bench_long.json")))
    d = d.get("translate", d)
    print("units/s %.4g  ms/step %.4f  wide %.4g  wide_all64 %.4g  %s" % (d["value"], d["ms_per_step"], d["wide"]["value"], d["wide_all64"]["value"], d["kernel_ms"]))
