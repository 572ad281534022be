#!/bin/bash
# quick kernel-trace of the bench (GPU box): per-kernel totals for the main kernels
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/tq
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
IMPG_GPU_LIB=${LIB:+$REPO/impg_amd/$LIB} timeout 300 rocprofv3 --kernel-trace -d $OUT/t -o t -- python $REPO/bench.py --ranges ${RANGES:-16384} --steps 2 --warmup 1 --cpu-sample 0 > /dev/null 2> $OUT/t.err
python3 $REPO/scripts/rocpd_summary.py $OUT/t/t_results.db $OUT/t
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))[:16]:
    print("%-40s calls=%s total_ms=%.2f avg_ms=%.3f max_ms=%.3f pct=%s" % (r["Name"][9:49], r["Calls"], int(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, int(r["MaxNs"])/1e6, r["Percentage"]))
PY
rm -rf $OUT/t
