#!/bin/bash
# A/B of two VALUES of an environment switch on ONE box:   bash tools/ab_env_vals.sh GAD_SOMETHING valA valB [repeats]
VAR=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do for v in $A $B; do
  env $VAR=$v python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-host-rate --no-sa-kernel 2>/dev/null | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$VAR=$v', round(r['value'],1), round(r['config']['iterations_per_s_sync_each_step'],1))"
done; done
