#!/bin/bash
# exploratory PMC passes for project_kernel (run on the GPU box)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TCC|TCP|TA|TD|GRBM)_[A-Za-z0-9_]+" | sort -u > $OUT/counters.txt
wc -l $OUT/counters.txt
ARGS="--ranges ${RANGES:-8192} --steps 1 --warmup 1 --cpu-sample 0"
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $OUT/p$i -o p$i -- python $REPO/bench.py $ARGS > /dev/null 2> $OUT/p$i.err
  python3 $REPO/scripts/rocpd_summary.py $OUT/p$i/p${i}_results.db $OUT/p$i 2>&1 | tail -2
  grep -E "project_|lookup_count" $OUT/p${i}_pmc.csv | cut -c1-40,100-
  rm -rf $OUT/p$i
done
