#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/...) into the small files kept under profiles/.

  summarize_prof.py stats <dir> <out.csv>          kernel_stats.csv with kernel names cut to 100 chars
  summarize_prof.py pmc <out.json> <label>=<dir>...  per-kernel averages of every counter found, KB units as rocprofv3 reports them
  summarize_prof.py traffic profiles/pmc_traffic.json <config key> fetch=<dir> write=<dir>   table bench.py reads `traffic` from
"""
import collections
import csv
import glob
import json
import os
import sys


def stats(src, out):
    f = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.reader(open(f)))
    with open(out, "w", newline="") as fo:
        w = csv.writer(fo)
        for r in rows:
            w.writerow([r[0][:100]] + r[1:])


def pmc(out, pairs):
    res = {}
    for p in pairs:
        label, d = p.split("=", 1)
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"][:100], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            res.setdefault(k, {})[c] = {"dispatches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v), "pass": label}
    json.dump(res, open(out, "w"), indent=1)


def traffic(out, key, pairs):
    """Merge the FETCH_SIZE / WRITE_SIZE means of one bench configuration into profiles/pmc_traffic.json under `key`
    (the key bench.py builds from its workload arguments)."""
    tmp = out + ".tmp"
    pmc(tmp, pairs)
    table = json.load(open(out)) if os.path.exists(out) else {}
    table[key] = json.load(open(tmp))
    os.unlink(tmp)
    json.dump(table, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4:])
    else:
        pmc(sys.argv[2], sys.argv[3:])
