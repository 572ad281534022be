#!/bin/bash
# The lane-sensitive multi-rank tests N times in a row with poisoned device buffers (IMPG_POISON), alternating patterns;
# summary -> gpurun_out/soak_multi.json (copied to profiles/ by hand).  usage: soak_multi.sh [N]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
N=${1:-20}
SEL="hitless or lane_schedules or save_load or row_stream or (matches_oracle and (3-2 or 5-3 or 8-2)) or concurrent_calls or mask_and_filter or failure_agreement_multi"
mkdir -p gpurun_out
pass=0; fail=0; t0=$(date +%s); lines=""
for i in $(seq 1 $N); do
  pat=$([ $((i % 2)) = 0 ] && echo a5 || echo ff)
  out=$(IMPG_POISON=$pat timeout 900 python -m pytest tests/test_multi_gpu.py -x -q -m gpu -k "$SEL" -p no:cacheprovider 2>&1 | tail -1)
  echo "run $i (poison $pat): $out"
  lines="$lines\"run $i (poison $pat): $out\","
  if echo "$out" | grep -q " passed" && ! echo "$out" | grep -q "failed"; then pass=$((pass+1)); else fail=$((fail+1)); fi
done
t1=$(date +%s)
echo "{\"what\": \"tests/test_multi_gpu.py -k '$SEL' under IMPG_POISON, $N runs in a row on one MI355X (scripts/soak_multi.sh)\", \"runs\": $N, \"passed\": $pass, \"failed\": $fail, \"seconds\": $((t1-t0)), \"lines\": [${lines%,}]}" > gpurun_out/soak_multi.json
cat gpurun_out/soak_multi.json
