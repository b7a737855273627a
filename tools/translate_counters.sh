#!/bin/bash
# GPU box: what the translate kernels move between the L2s and the memory side, by request size (round 5).  FETCH_SIZE tallies every
# read request at 64 bytes (MI355X_MICROARCH.md: exactly half of a wide streaming read, whose requests are 128 bytes) -- a kernel of
# scattered 4- and 8-byte reads is not a streaming read, so its requests are counted by size here: bytes = 32 x RDREQ_32B + 64 x RDREQ_64B
# + 128 x RDREQ_128B, and for writes 64 x WRREQ_64B + 32 x (WRREQ - WRREQ_64B).  One rocprofv3 --pmc pass per group; a group that fails
# stops the script.
# usage: tools/translate_counters.sh [out dir under gpurun_out]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-trpmc}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "FETCH_SIZE" \
           "WRITE_SIZE"; do
  i=$((i+1))
  if ! rocprofv3 --pmc $grp --output-format csv -d $O/g$i -- python3 $R/bench.py --path translate --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end --long-form none > $O/g$i.log 2>&1; then
    echo "translate_counters.sh: group $i ($grp) failed:"; tail -5 $O/g$i.log
    exit 1
  fi
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$O/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:64]
        if "translate" not in k and "scan" not in k and "tile_sums" not in k:
            continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for k in sorted(acc):
    m = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    print(k)
    for c in sorted(m):
        print("   %-28s %16.0f  (per launch, %d launches)" % (c, m[c], cnt[k][c]))
    if "TCC_EA0_RDREQ_sum" in m:
        rd = 32 * m.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * m.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * m.get("TCC_EA0_RDREQ_128B_sum", 0)
        other = m["TCC_EA0_RDREQ_sum"] - m.get("TCC_EA0_RDREQ_32B_sum", 0) - m.get("TCC_EA0_RDREQ_64B_sum", 0) - m.get("TCC_EA0_RDREQ_128B_sum", 0)
        print("   read bytes by request size   %16.0f  (requests of no counted size: %.0f)" % (rd, other))
    if "TCC_EA0_WRREQ_sum" in m:
        wr = 64 * m.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (m["TCC_EA0_WRREQ_sum"] - m.get("TCC_EA0_WRREQ_64B_sum", 0))
        print("   write bytes by request size  %16.0f" % wr)
PY
