#!/bin/bash
# A/B of kernel variants: scripts/ab_bench.sh <lib1.so> <lib2.so> ...  (run on the GPU box)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for lib in "$@"; do
  for rep in 1 2; do
    IMPG_GPU_LIB=$REPO/impg_amd/$lib timeout 300 python bench.py --ranges ${RANGES:-16384} --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['stage_ms_per_step_rank0']
print('$lib rep$rep value=%.3e ms/step=%.1f lookup=%.1f project=%.1f update=%.1f frac=%.3f' % (d['value'], d['ms_per_step'], s['lookup'], s['project'], s['update'], d['roofline']['frac']))"
  done
done
