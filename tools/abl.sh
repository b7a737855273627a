R=$GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py --path dp --no-cpu-baseline --no-end-to-end --steps 5 --config ns 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(' ', round(d['value']), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k!='note'}); c=d['c1']; print('  c1', round(c['value']), round(c['ms_per_step'],3), {k:round(v,3) for k,v in c['kernel_ms'].items() if k!='note'})"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/lay_f $R/gpurun_out/lay_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/lay_f -- python3 $R/bench.py --path dp --config ns --no-c1 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/lay_w -- python3 $R/bench.py --path dp --config ns --no-c1 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>&1
