#!/bin/bash
# A/B of several builds of the library on ONE box:  bash tools/ab_libs.sh [repeats] <lib.so> ...   ("" = the in-tree build)
N=$1; shift
run() { env GAD_LIB_PATH=$1 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-host-rate --no-sa-kernel 2>/dev/null | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print('${1:-in-tree}', round(r['value'],1), round(r['config']['iterations_per_s_sync_each_step'],1))"; }
for i in $(seq $N); do for l in "$@" ""; do run $l; done; done
