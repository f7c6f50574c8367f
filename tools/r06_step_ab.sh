#!/bin/bash
# round 6: whole-step replay list vs per-plan replay vs Python walk -- full GPU suite, host enqueue time, step rates (same box)
mkdir -p gpurun_out
{
echo "== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for c in "GAD_STEP_PLAN=0 GAD_PLAN_C=0" "GAD_STEP_PLAN=0 GAD_PLAN_C=1" "GAD_STEP_PLAN=1 GAD_PLAN_C=1" "GAD_STEP_PLAN=0 GAD_PLAN_C=0" "GAD_STEP_PLAN=1 GAD_PLAN_C=1"; do
  echo "== $c host runahead"; env $c timeout 300 python tools/diag_host_runahead.py 2>&1 | tail -2
done
for c in "GAD_STEP_PLAN=0 GAD_PLAN_C=0" "GAD_STEP_PLAN=1 GAD_PLAN_C=1" "GAD_STEP_PLAN=0 GAD_PLAN_C=0" "GAD_STEP_PLAN=1 GAD_PLAN_C=1"; do
  echo "== $c bench"; env $c timeout 600 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-sa-kernel 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']
        print(d['value'], {k:c.get(k) for k in ('iterations_per_s_sync_each_step','value_host_inclusive','value_host_prefetch','value_device_replay','value_f32_mfma','value_split')})
"
done
} > gpurun_out/r06_step_ab.txt 2>&1
tail -40 gpurun_out/r06_step_ab.txt
