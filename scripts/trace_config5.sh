#!/bin/bash
# kernel trace of a small BASELINE config-5 run (contiguous 5 kb windows, -x -m 5) on the GPU box
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/c5
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
DEPTHS=${DEPTHS:-5} CHUNK=${CHUNK:-100} timeout 600 rocprofv3 --kernel-trace -d $OUT/t -o t -- python $REPO/scripts/config45_probe.py 5 ${WINDOWS:-200} ${RECORDS:-5e6} ${NSEQ:-1000} > $OUT/probe.json 2> $OUT/t.err
python3 $REPO/scripts/rocpd_summary.py $OUT/t/t_results.db $OUT/t
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel ms total: %.1f" % (tot/1e6))
for r in rows[:16]:
    print("%-60s calls=%-5s %.1f ms  %.1f%%  avg %.3f ms" % (r["Name"][:60], r["Calls"], int(r["TotalDurationNs"])/1e6, 100*float(r["TotalDurationNs"])/tot, float(r["AverageNs"])/1e6))
PY
tail -2 $OUT/t.err; cat $OUT/probe.json
rm -rf $OUT/t
