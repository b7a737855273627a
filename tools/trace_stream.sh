#!/bin/bash
# GPU box: kernel and copy trace of two calls of the host-fed engine on one DP shape, the second printed as a timeline (ms from its
# first event).  usage: tools/trace_stream.sh SPEC SEGMENTS [out name]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${3:-trace_stream}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python3 $R/tools/stream_probe.py --trace $1 $2 > $O/run.log 2>&1 || echo "rocprofv3 failed"
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "dp_" in n:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.split("(")[0][-40:] + " grid " + r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
for f in glob.glob("$O/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "?") + " " + r.get("Bytes", r.get("Size", "?"))))
rows.sort()
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - max(e for _, e, *_ in rows[:i]) > 20_000_000:
        cut = i
last = rows[cut:]
t0 = last[0][0]
for s, e, n in last:
    if e - s > 200_000:
        print("%9.3f .. %9.3f  (%8.3f ms)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
PY
rm -rf $O/*/
