#!/bin/bash
# GPU box: kernel trace of two passes of one DP shape, printed as a timeline (ms from the first kernel of the last pass).
# usage: tools/trace_once.sh SPEC [out name]   SPEC as tools/shares_probe.py takes it (ns:N, c2:N, deep:N, u:N:ROWS:LEN)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${2:-trace_once}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/tools/shares_probe.py --trace $1 > $O/run.log 2>&1 || echo "rocprofv3 failed"
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "dp_" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0][-46:], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?")))
rows.sort()
# the last pass: kernels after the last gap of more than 1 ms
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - max(e for _, e, *_ in rows[:i]) > 1_000_000:
        cut = i
last = rows[cut:]
t0 = last[0][0]
for s, e, n, g, w in last:
    print("%9.3f .. %9.3f  (%8.3f ms)  grid %8s x %4s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, g, w, n))
PY
