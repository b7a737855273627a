echo c2; timeout -k 10 500 python bench.py --path dp --config c2 --no-cpu-baseline --no-c1 --no-end-to-end --steps 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'], d['config']['chunks'])"
