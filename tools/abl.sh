timeout -k 10 200 python bench.py --path dp --no-cpu-baseline --steps 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['config']['chunks']); c=d['c1']; print('  c1', c['value'], c['ms_per_step'], c['kernel_ms'])"
