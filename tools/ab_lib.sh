#!/bin/bash
# A/B of two builds of the library on ONE box:  bash tools/ab_lib.sh <other libgaddpg.so> [repeats]   (GAD_LIB_PATH)
OTHER=$1; N=${2:-2}
run() { env "$@" python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-host-rate --no-sa-kernel 2>/dev/null | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$*', round(r['value'],1), round(r['config']['iterations_per_s_sync_each_step'],1))"; }
for i in $(seq $N); do run GAD_LIB_PATH=$OTHER; run GAD_LIB_PATH=; done
