timeout 600 python -m pytest tests/test_gpu_schedule_paths.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -2
run() { timeout 300 python bench.py $2 --steps 30 --no-cpu-baseline --no-adjacent --no-s0 --no-probe 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d.get('engine_sized_cache') or {}
print('$1', round(d['ms_per_step'],3), {k[:2]:round(v,4) for k,v in d['stages_ms'].items()}, '| engine', round(e.get('ms_per_step',0),3), {k[:2]:round(v,4) for k,v in (e.get('stages_ms') or {}).items()}, round(e.get('value',0)/1e9,3))"; }
run default ""
run spare30 "--spare-blocks 30 --no-engine-cache"
run spare3 "--spare-blocks 3 --no-engine-cache"
