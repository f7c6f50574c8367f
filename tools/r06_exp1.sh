#!/bin/bash
mkdir -p gpurun_out
{
echo "== wide forward phases"; GAD_LIB_PATH=tools/_ab/lib_wph.so timeout 600 python tools/ubench_wphases.py fwd_wide 2>&1 | tail -40
echo "== dw_wide_wgs A/B"; bash tools/ab_opt.sh dw_wide_wgs "256 128 64" 2
} > gpurun_out/r06_exp1.txt 2>&1
cat gpurun_out/r06_exp1.txt
