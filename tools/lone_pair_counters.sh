# GPU box: instruction and wait counters of the fill kernel for ONE pair of 32 x 10 kbp (a lone wave per SIMD): two rocprofv3 --pmc passes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lonepmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d $O/g1 -- python3 $R/tools/dp_run_once.py 1:32:10000 2 > $O/g1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/g2 -- python3 $R/tools/dp_run_once.py 1:32:10000 2 > $O/g2.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$O/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:40]
        if "dp_fill" not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in acc:
    print(k)
    for c in sorted(acc[k]): print("   %-24s %14.0f" % (c, acc[k][c] / cnt[k][c]))
PY
