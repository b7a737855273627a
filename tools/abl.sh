run() { timeout -k 10 300 python bench.py --path dp --config $1 --dp-pairs $2 --no-cpu-baseline --no-c1 --no-end-to-end --steps 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(' ', round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k!='note'}, d['config']['kernel_variant']['checkpoints'], d['config']['kernel_variant']['cols_per_lane'])"; }
for cfg in "deep 1" "deep 4" "deep 32" "deep 100" "deep 128" "c1 64" "ns 1" "ns 100"; do for g in 1 -1; do echo "groups=$g $cfg"; PM_DP_GROUPS=$g run $cfg; done; done
