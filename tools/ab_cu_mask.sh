run() { env "$@" python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-host-rate --no-sa-kernel 2>/tmp/err.txt | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$*', round(r['value'],1), round(r['config']['iterations_per_s_sync_each_step'],1))" || tail -3 /tmp/err.txt; }
for i in 1 2; do
run X=0
run GAD_CU_MASK_C=128
run GAD_CU_MASK_C=even
run GAD_CU_MASK_C=64
run GAD_STREAM_MAP=1:A,11:C,2:B,12:C,3:C,20:A,21:A
run GAD_STREAM_MAP=1:A,11:C,2:B,12:C,3:C,20:A,21:A GAD_CU_MASK_C=128
run GAD_STREAM_MAP=1:A,11:C,2:B,12:C,3:C,20:A,21:A GAD_CU_MASK_C=even
done
