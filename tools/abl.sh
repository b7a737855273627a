run() { echo "$@"; timeout -k 10 400 python bench.py --path dp --no-cpu-baseline --no-c1 --no-end-to-end --steps 5 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(' ', round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k!='note'}, d['config']['chunks'])"; }
run --config ns
run --config c2 --steps 3
run --config deep
