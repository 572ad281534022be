#!/bin/bash
# kernel trace of the headline step (GPU box): per-kernel time PER STEP for the top kernels.
# usage: [LIB=libimpg_x.so] [STEPS=2] [TOP=30] [BENCH_ARGS=...] scripts/trace_r5.sh [tag]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-tq}
OUT=$REPO/gpurun_out/$TAG
STEPS=${STEPS:-2}
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
IMPG_GPU_LIB=${LIB:+$REPO/impg_amd/$LIB} timeout 400 rocprofv3 --kernel-trace -d $OUT/t -o t -- python $REPO/bench.py --steps $STEPS --warmup 1 --cpu-sample 0 --no-extras ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/t.err
python3 $REPO/scripts/rocpd_summary.py $OUT/t/t_results.db $OUT/t
python3 - <<PY
import csv, re
steps = $STEPS + 1
rows = list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))
tot = 0.0
for r in rows[:${TOP:-30}]:
    name = r["Name"]
    name = re.sub(r"^_ZN4impg\d+", "", name)[:58]
    ms = int(r["TotalDurationNs"]) / 1e6
    print("%-58s calls/step=%5.1f ms/step=%7.3f avg_us=%9.1f" % (name, int(r["Calls"]) / steps, ms / steps, float(r["AverageNs"]) / 1e3))
print("all kernels: %.2f ms/step" % (sum(int(r["TotalDurationNs"]) for r in rows) / 1e6 / steps))
PY
rm -rf $OUT/t
