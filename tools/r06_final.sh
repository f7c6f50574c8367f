#!/bin/bash
# round 6 evidence run: full GPU suite, then the profile collection (kernel stats, PMC traffic incl. query_and_group, SQ counters, bench line)
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r06_full_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r06_full_gpu_suite.txt 2>&1 && echo "smoke ok" >> gpurun_out/r06_full_gpu_suite.txt
bash tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1
python tools/prof_summary.py /tmp/prof_stats 90 stream 0 > gpurun_out/r06_streams.txt 2>&1
timeout 300 python tools/diag_host_runahead.py > gpurun_out/r06_host_enqueue.txt 2>&1
for la in 0 1 0 1; do GAD_TRAIN_LOOKAHEAD=$la timeout 300 python tools/diag_train_loop.py 2>/dev/null; done > gpurun_out/r06_train_loop.txt
tail -3 gpurun_out/r06_full_gpu_suite.txt; tail -c 1500 gpurun_out/r06_bench_B256.json
