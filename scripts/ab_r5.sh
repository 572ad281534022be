#!/bin/bash
# A/B of library variants on the headline step: scripts/ab_r5.sh <lib1.so> <lib2.so> ...  (run on the GPU box; libs under impg_amd/)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
for lib in "$@"; do
  for rep in 1 2; do
    IMPG_GPU_LIB=$REPO/impg_amd/$lib timeout 300 python bench.py --steps ${STEPS:-5} --warmup 2 --cpu-sample 0 --no-extras ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['stage_ms_per_step_rank0']
print('$lib rep$rep value=%.3e ms/step=%.2f lookup=%.2f project=%.2f update=%.2f self_check=%s' % (d['value'], d['ms_per_step'], s['lookup'], s['project'], s['update'], d.get('self_check')))"
  done
done
