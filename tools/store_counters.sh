#!/bin/bash
# GPU box: what the fill kernel's checkpoint stores cost the memory side (round 4): one rocprofv3 --pmc pass per group.
# usage: tools/store_counters.sh [pairs:rows:len] [out dir under gpurun_out]
# Round 5: the groups are ones rocprofv3 accepts (round 4's group 4 -- four TA counters and GRBM_GUI_ACTIVE in one pass -- ended the
# profiled process with rocprofiler_create_counter_config error 38, "exceeds the capabilities of the hardware", and the script hid
# it behind an echo: the TA-side counters its conclusion argued from were never collected).  TA counters two at a time, GRBM alone;
# a group that fails stops the script with a non-zero exit.
R=$GRAFT_REPO_ROOT
SHAPE=${1:-5120:8:4096}
O=$R/gpurun_out/${2:-storepmc}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_avr" \
           "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "TA_BUSY_avr TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUSY_max" \
           "GRBM_GUI_ACTIVE" \
           "WRITE_SIZE" \
           "FETCH_SIZE" \
           "SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
  # (round 4's last group -- TCC_EA0_WRREQ_LEVEL_sum TCC_REQ_sum TCC_WRITE_sum TCC_HIT_sum TCC_MISS_sum -- ends rocprofv3 with signal 6
  # and then never returns: dropped, gpurun_out/storepmc5/g10.log of round 5 has the abort)
  i=$((i+1))
  if ! rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python3 $R/tools/dp_run_once.py $SHAPE 2 > $O/g$i.log 2>&1; then
    echo "store_counters.sh: group $i ($grp) failed:"; tail -5 $O/g$i.log
    exit 1
  fi
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
