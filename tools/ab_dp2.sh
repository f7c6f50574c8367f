#!/bin/bash
run() { env "$@" python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-host-rate --no-sa-kernel 2>/dev/null | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$*', round(r['value'],1), round(r['config']['iterations_per_s_sync_each_step'],1))"; }
run GAD_BENCH_FORCE_DP=0
run GAD_BENCH_FORCE_DP=1 GAD_DP_BUCKETS=0 GAD_DP_DIRECT_RCCL=0
run GAD_BENCH_FORCE_DP=1 GAD_DP_BUCKETS=0 GAD_DP_DIRECT_RCCL=1
run GAD_BENCH_FORCE_DP=1 GAD_DP_BUCKETS=1 GAD_DP_DIRECT_RCCL=1
run GAD_BENCH_FORCE_DP=1 GAD_DP_BUCKETS=1 GAD_DP_DIRECT_RCCL=0
