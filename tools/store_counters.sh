#!/bin/bash
# GPU box: what the fill kernel's checkpoint stores cost the memory side (round 4): one rocprofv3 --pmc pass per group.
# usage: tools/store_counters.sh [pairs:rows:len] [out dir under gpurun_out]
R=$GRAFT_REPO_ROOT
SHAPE=${1:-5120:8:4096}
O=$R/gpurun_out/${2:-storepmc}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_avr" \
           "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "TA_BUSY_avr TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "TCC_EA0_WRREQ_LEVEL_sum TCC_REQ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python3 $R/tools/dp_run_once.py $SHAPE 2 > $O/g$i.log 2>&1 || echo "group $i failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$O/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        print("   %-40s %18.0f  (per launch, %d launches)" % (c, acc[k][c] / cnt[k][c], cnt[k][c]))
PY
