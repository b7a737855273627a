#!/bin/bash
# Runs on the GPU box (through gpurun): the default bench line, and per workload (DP ns, c1, c2, deep; translate) one rocprofv3
# kernel-trace pass and three PMC passes (FETCH_SIZE, WRITE_SIZE, SQ counters: separate runs, as the TCC slots require).
# Raw outputs land in gpurun_out/refresh/ and are condensed on the box by tools/refresh_profiles_local.py into
# gpurun_out/profiles_new/ (copy that into profiles/ afterwards).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
ONLY=${PM_REFRESH_ONLY:-ns c1 c2 deep tr} # a subset of the workloads: PM_REFRESH_ONLY="tr" (then the default bench line is skipped too)
# (round 5: stdout holds the compact line; the long form is the file bench.py writes beside it)
[ -z "$PM_REFRESH_ONLY" ] && { cd $R && python bench.py > $O/bench.json 2> $O/bench.err; cp $R/gpurun_out/bench_long.json $O/bench_long.json; }
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
run() { # name, bench arguments...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_stats -- python3 $R/bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --long-form none > $O/${name}_stats.log 2>&1 &&
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${name}_fetch -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --long-form none > $O/${name}_fetch.log 2>&1 &&
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${name}_write -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --long-form none > $O/${name}_write.log 2>&1 &&
  rocprofv3 --pmc $SQ --output-format csv -d $O/${name}_sq -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --long-form none > $O/${name}_sq.log 2>&1
}
for w in $ONLY; do
  if [ $w = tr ]; then run tr --path translate || break; else run $w --path dp --config $w --no-ride-alongs || break; fi
done
# condensed here: the raw directories are hundreds of MB, gpurun carries 64 MiB back
cd $R && PM_PROFILE_OUT=$R/gpurun_out/profiles_new python3 tools/refresh_profiles_local.py ${1:-r03} > $R/gpurun_out/profiles_new.log 2>&1
cp $O/bench.json $O/bench.err $O/bench_long.json $R/gpurun_out/profiles_new/ 2>/dev/null
rm -rf $O
true
