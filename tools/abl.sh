run() { timeout -k 10 300 python bench.py --path dp --config $1 --dp-pairs $2 --no-cpu-baseline --no-c1 --no-end-to-end --steps 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(' ', round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k!='note'}, d['config']['kernel_variant']['checkpoints'], d['config']['kernel_variant']['cols_per_lane'])"; }
for cfg in "ns 1024" "ns 2048" "ns 4096" "ns 8192" "deep 512" "c1 2000" "c1 4000"; do for l in 16 32; do echo "lanes=$l $cfg"; PM_DP_WALK_LANES=$l run $cfg; done; done
