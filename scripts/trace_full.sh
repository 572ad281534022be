#!/bin/bash
# kernel trace of the default bench command (GPU box): per-kernel totals per step
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/tf
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/t -o t -- python $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-extras > $OUT/bench.json 2> $OUT/t.err
python3 $REPO/scripts/rocpd_summary.py $OUT/t/t_results.db $OUT/t
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel ms per step: %.1f" % (tot/3e6))
for r in rows[:34]:
    print("%-44s calls=%-4s per-step %.2f ms  %.1f%%" % (r["Name"][9:53], r["Calls"], int(r["TotalDurationNs"])/3e6, 100*float(r["TotalDurationNs"])/tot))
PY
rm -rf $OUT/t
