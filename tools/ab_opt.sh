#!/bin/bash
# A/B of library option values on ONE box:  bash tools/ab_opt.sh <option> "<v1> <v2> ..." [repeats]
OPT=$1; VALS=$2; N=${3:-3}
for i in $(seq $N); do for v in $VALS; do
  env GAD_OPT_$OPT=$v python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-host-rate --no-sa-kernel 2>/dev/null | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$OPT=$v', round(r['value'],1), round(r['config']['iterations_per_s_sync_each_step'],1))"
done; done
