#!/bin/bash
# Runs on the GPU box (through gpurun): the default bench line, rocprofv3 kernel stats and the three PMC passes.
# Outputs land in gpurun_out/refresh/; tools/summarize_prof.py condenses them into profiles/ afterwards.
# Delete the local gpurun_out/refresh/ before the call: gpurun merges new files into it and older runs would linger.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
rm -rf $O; mkdir -p $O
cd $R && python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O/sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/sq.log 2>&1
true
