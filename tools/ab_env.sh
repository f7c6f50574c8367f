#!/bin/bash
# A/B of an environment switch on ONE box: bench.py alternating VAR=0 / VAR=1 (boxes differ by +-1.5 %: only same-call
# comparisons mean anything).   bash tools/ab_env.sh GAD_SOMETHING [repeats]
VAR=$1; N=${2:-3}
for i in $(seq $N); do for v in 0 1; do
  env $VAR=$v python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-host-rate --no-sa-kernel 2>/dev/null | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$VAR=$v', round(r['value'],1), round(r['config']['iterations_per_s_sync_each_step'],1))"
done; done
