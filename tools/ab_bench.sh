#!/bin/bash
# A/B of environment settings on the bench rate: ab_bench.sh "GAD_X=1" "GAD_X=0" ... (each config run REPS times, interleaved)
REPS=${REPS:-2}
STEPS=${STEPS:-100}
for rep in $(seq $REPS); do
  for cfg in "$@"; do
    env $cfg python bench.py --steps $STEPS --warmup 20 --no-sa-kernel --no-cpu-baseline --no-host-rate --probe-steps 2 2>/dev/null | \
      CFG="$cfg" python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s steps/s %.1f   sync-each-step %.1f' % (os.environ['CFG'], d['value'], d['config']['iterations_per_s_sync_each_step']))"
  done
done
