#!/bin/bash
# The first N > 1 run of this tree: nothing here has ever executed with more than one GPU.  Run on a node with
# 2 / 4 / 8 MI355X; every bench line carries config.transport, rccl_nranks, rccl_comms, bucketed_exchange, replicas_bit_identical and
# the co-run self-check (allreduce_corun_checked, corun_*: known-answer all-reduces beside split-bf16 GEMM launches).
#   bash tools/first_multigpu_run.sh [max_gpus]        -> gpurun_out/first_multigpu/*
NMAX=${1:-8}
OUT=${GRAFT_REPO_ROOT:-.}/gpurun_out/first_multigpu
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
line() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']
        print('$1', 'value', round(d['value'],1), 'n_gpus', d['n_gpus'], {k:c.get(k) for k in ('transport','rccl_nranks','rccl_comms','bucketed_exchange','replicas_bit_identical','allreduce_corun_checked','corun_exchange_mismatches','corun_torch_kernel_mismatches','iterations_per_s')})
"; }
echo "== 1. the skipped 2-GPU test (direct RCCL, world size 2)"
timeout 900 python -m pytest tests/test_gpu_dp.py -q -m gpu -k "two_gpus" 2>&1 | tail -3 | tee $OUT/test_two_gpus.txt
for N in 1 2 4 8; do
  [ $N -gt $NMAX ] && break
  for COMMS in one lanes; do
    for BUCK in 0 1; do
      [ $N -eq 1 ] && [ "$COMMS$BUCK" != "one0" ] && continue
      tag="n${N}_comms-${COMMS}_buckets-${BUCK}"
      echo "== 2. bench --gpus $N  GAD_DP_COMMS=$COMMS GAD_DP_BUCKETS=$BUCK"
      GAD_DP_COMMS=$COMMS GAD_DP_BUCKETS=$BUCK timeout 1200 python bench.py --gpus $N --steps 100 --warmup 20 --no-cpu-baseline --no-sa-kernel \
        > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
      line $tag < $OUT/bench_$tag.json | tee -a $OUT/summary.txt
    done
  done
done
# 3. do the lanes' exchanges overlap?  kernel trace of rank 0 at the largest N (rccl kernels per stream against the step's GEMMs)
N=$NMAX
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== 3. rocprofv3 kernel trace, $N ranks, per-lane communicators + bucketed exchange"
GAD_DP_COMMS=lanes GAD_DP_BUCKETS=1 timeout 1500 rocprofv3 --kernel-trace --stats -d /tmp/prof_dp -o r -- \
  python bench.py --gpus $N --steps 30 --warmup 10 --no-cpu-baseline --no-sa-kernel --no-host-rate > $OUT/prof_bench.log 2>&1
python tools/prof_summary.py /tmp/prof_dp 40 stream 0 > $OUT/prof_streams.txt 2>&1
grep -i "nccl\|rccl" $OUT/prof_streams.txt | head -20
echo "summary: $OUT/summary.txt"
