#!/bin/bash
# Round evidence on the GPU box: bench line, rocprofv3 kernel stats, PMC traffic (FETCH / WRITE in separate passes) and the
# MFMA / VALU instruction counters.  Writes gpurun_out/rNN_*; copy what is to be judged into profiles/.
R=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
BENCH="python bench.py --no-cpu-baseline --no-host-rate --no-sa-kernel"
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o r -- $BENCH --steps 50 --warmup 10 > $OUT/${R}_stats_bench.log 2>&1
python tools/prof_summary.py /tmp/prof_stats 90 > $OUT/${R}_rocprofv3_kernel_stats_bench_B256.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_fetch -o r -- $BENCH --steps 6 --warmup 4 --probe-steps 2 > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_write -o r -- $BENCH --steps 6 --warmup 4 --probe-steps 2 > /dev/null 2>&1
python tools/prof_traffic.py /tmp/prof_fetch /tmp/prof_write $OUT/${R}_traffic.json > $OUT/${R}_traffic.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES -d /tmp/prof_mfma -o r -- $BENCH --steps 6 --warmup 4 --probe-steps 2 > /dev/null 2>&1
python tools/prof_pmc.py /tmp/prof_mfma 600 > $OUT/${R}_rocprofv3_pmc_SQ_mfma.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES -d /tmp/prof_waits -o r -- $BENCH --steps 6 --warmup 4 --probe-steps 2 > /dev/null 2>&1
python tools/prof_pmc.py /tmp/prof_waits 800 > $OUT/${R}_rocprofv3_pmc_SQ_waits.txt 2>&1
python tools/prof_derived.py $OUT/${R}_rocprofv3_pmc_SQ_mfma.txt $OUT/${R}_rocprofv3_pmc_SQ_waits.txt > $OUT/${R}_counter_table.txt 2>&1
# BASELINE's second metric under the counters: the configs[3] query_and_group kernel, merged into the same traffic file
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/qg_fetch -o r -- python tools/prof_query_and_group.py run > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/qg_write -o r -- python tools/prof_query_and_group.py run > /dev/null 2>&1
python tools/prof_query_and_group.py merge /tmp/qg_fetch /tmp/qg_write $OUT/${R}_traffic.json >> $OUT/${R}_traffic.txt 2>&1
cp $OUT/${R}_traffic.json profiles/${R}_traffic.json 2>/dev/null     # bench.py reads roofline.traffic from here
python bench.py --steps 200 --warmup 50 > $OUT/${R}_bench_B256.json 2> $OUT/${R}_bench_B256.err
tail -c 600 $OUT/${R}_bench_B256.json
