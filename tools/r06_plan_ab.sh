#!/bin/bash
# round 6: C plan replay vs the Python walk -- parity subset, host enqueue time, step rates (same box)
mkdir -p gpurun_out
{
echo "== tests (C plans)"; timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_modules.py tests/test_gpu_facade.py -x -q 2>&1 | tail -5
for c in 0 1 0 1; do
  echo "== GAD_PLAN_C=$c host runahead"; GAD_PLAN_C=$c timeout 300 python tools/diag_host_runahead.py 2>&1 | tail -2
done
for c in 0 1 0 1; do
  echo "== GAD_PLAN_C=$c bench"; GAD_PLAN_C=$c timeout 600 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-sa-kernel 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']
        print(d['value'], {k:c.get(k) for k in ('iterations_per_s_sync_each_step','value_host_inclusive','value_host_prefetch','value_device_replay','value_f32_mfma','value_split')})
"
done
} > gpurun_out/r06_plan_ab.txt 2>&1
tail -40 gpurun_out/r06_plan_ab.txt
