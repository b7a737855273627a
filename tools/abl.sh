run() { echo "$@"; env "$@" timeout -k 10 400 python bench.py --path dp --no-cpu-baseline --no-end-to-end --steps 5 --config ns 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(' ', round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k!='note'}, d['config']['kernel_variant']['uniform_depth']); c=d['c1']; print('  c1', round(c['value']), round(c['ms_per_step'],3), {k:round(v,3) for k,v in c['kernel_ms'].items() if k!='note'})"; }
run PM_DP_UNI=1
run PM_DP_UNI=0
