#!/bin/bash
mkdir -p gpurun_out
{
echo "== grouped dW gates"; timeout 900 python -m pytest tests/test_gpu_split_families.py -x -q -k "grouped or dw_wide" 2>&1 | tail -4
echo "== step gates"; timeout 1200 python -m pytest tests/test_gpu_step.py tests/test_gpu_forced_decisions.py -x -q 2>&1 | tail -3
echo "== bench A/B GAD_GROUP_DW"; bash tools/ab_env.sh GAD_GROUP_DW 3
} > gpurun_out/r06_exp8.txt 2>&1
cat gpurun_out/r06_exp8.txt
