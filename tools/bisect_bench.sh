#!/bin/bash
# bench rate of the given trees, interleaved REPS times on one box.  Trees are git worktrees built in place, e.g.
#   git worktree add -f _r01 <commit> && (cd _r01 && python -c "import __graft_entry__ as g; g.build()")
# (_r01/ and _bis/ are git- and gpurun-ignored; remove them from .gpurunignore while they are needed on the GPU box)
REPS=${REPS:-2}
STEPS=${STEPS:-100}
TREES=${TREES:-"_r01 _bis/2e9a10e _bis/7245c40 _bis/48ee1d7 _bis/cebbbed _bis/62dc03d _bis/38bceaf ."}
run() { (cd $1 && env ${2//+/ } python bench.py --steps $STEPS --warmup 20 --no-sa-kernel --no-cpu-baseline --no-host-rate 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-16s %-22s %.1f %s' % ('$1', '$2', d['value'], d['config'].get('iterations_per_s_sync_each_step','')))"); 
  if [ -n "$SMI" ]; then rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|Average Graphics\|junction" | head -3 | tr '\n' ' '; echo; fi; }
for rep in $(seq $REPS); do
  for t in $TREES; do
    if [ "$t" = "." ] && [ -n "$ENVS" ]; then for e in $ENVS; do run $t $e; done; else run $t; fi
  done
done
