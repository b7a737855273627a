#!/bin/bash
# GPU box: the shader clock the fill kernel runs at, with and without its checkpoint stores (round 4: do the stores cost clock?).
# SQ_BUSY_CYCLES per launch / the launch's duration (kernel trace of the same command).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/clockprobe
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in ckpt scores; do
  if [ $mode = scores ]; then export RUN_ONCE_SCORES_ONLY=1; else unset RUN_ONCE_SCORES_ONLY; fi
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/${mode}_pmc -- python3 $R/tools/dp_run_once.py 5120:8:4096 3 > $O/${mode}_pmc.log 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $O/${mode}_trace -- python3 $R/tools/dp_run_once.py 5120:8:4096 3 > $O/${mode}_trace.log 2>&1
done
python3 - <<PY
import csv, glob
for mode in ("ckpt", "scores"):
    cyc = {}
    for f in glob.glob("$O/%s_pmc/**/*counter_collection.csv" % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            if "dp_fill" in r["Kernel_Name"]:
                cyc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    dur = []
    for f in glob.glob("$O/%s_trace/**/*kernel_trace.csv" % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            if "dp_fill" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print(mode, "fill launches ms:", ["%.3f" % d for d in dur], {k: "%.4g" % (sum(v) / len(v)) for k, v in cyc.items()})
    if dur and "SQ_BUSY_CYCLES" in cyc:
        b = sum(cyc["SQ_BUSY_CYCLES"]) / len(cyc["SQ_BUSY_CYCLES"])
        d = min(dur)
        print("   SQ_BUSY_CYCLES / duration = %.2f G/s (divide by the number of SQ instances counted: 32 shader engines x ... -> GHz)" % (b / d / 1e6))
PY
