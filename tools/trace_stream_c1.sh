#!/bin/bash
# GPU box: kernels AND memory copies of the host-fed engine on BASELINE configs[1] (tools/stream_timing.py), the last "columns" pass as a
# timeline (ms from its first event).  usage: tools/trace_stream_c1.sh [out name]   (SEGMENTS=n as stream_timing.py takes it)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-trace_stream}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ONLY_COLUMNS=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python3 $R/tools/stream_timing.py > $O/run.log 2>&1 || echo "rocprofv3 failed"
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:] + " grid " + r.get("Grid_Size_X", "?")))
for f in glob.glob("$O/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?") + " " + r.get("Size", r.get("Bytes", "?")) + " B"))
rows.sort()
cut = 0
busy = rows[0][1]
for i in range(1, len(rows)):
    if rows[i][0] - busy > 1_000_000:
        cut = i
    busy = max(busy, rows[i][1])
last = rows[cut:]
t0 = last[0][0]
for s, e, n in last:
    if e - s > 3000:
        print("%8.3f .. %8.3f  (%7.3f ms)  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n))
PY
