#!/bin/bash
# One short bench line per argument ("<config> <pairs>"), condensed: GCUPS, ms per step, kernel times, path mode, columns per lane.
# PM_DP_* overrides pass through the environment: PM_DP_BAND=0 bash tools/abl.sh "deep 128" "ns 2048"
run() { timeout -k 10 300 python bench.py --path dp --config $1 --dp-pairs $2 --no-cpu-baseline --no-c1 --no-end-to-end --steps 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(' ', round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k!='note'}, d['config']['kernel_variant']['checkpoints'], d['config']['kernel_variant']['cols_per_lane'])"; }
for cfg in "${@:-ns 12500}"; do echo "$cfg"; run $cfg; done
