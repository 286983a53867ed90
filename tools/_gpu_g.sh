timeout 600 python -m pytest tests/test_gpu_schedule_paths.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
timeout 300 python tools/soak_schedule_paths.py 300 2>&1 | tail -1
KVC_FUZZ_SEEDS=800 timeout 400 python -m pytest tests/test_gpu_parity.py -k fuzz -x -q 2>&1 | tail -1
for cfg in "--config c3" "--config c3 --mode reference" "--batch 64 --steady-cap 4096"; do
  timeout 300 python bench.py $cfg --steps 20 --no-cpu-baseline --no-adjacent --no-s0 --no-engine-cache --no-probe 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:24], round(d['ms_per_step'],3), {k[:2]:round(v,3) for k,v in d['stages_ms'].items()}, d['S1_schedule'])"
done
