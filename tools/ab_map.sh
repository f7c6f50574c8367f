#!/bin/bash
# A/B of logical-stream -> physical-stream maps on ONE box:  bash tools/ab_map.sh "<map1>" "<map2>" ...   ("" = default)
for i in 1 2; do for m in "$@"; do
  env GAD_STREAM_MAP="$m" python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-host-rate --no-sa-kernel 2>/dev/null | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print('map[$m]', round(r['value'],1), round(r['config']['iterations_per_s_sync_each_step'],1))"
done; done
