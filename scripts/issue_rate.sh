#!/bin/bash
# issue-rate micro-benchmark + the SQ counters of the same binary (calibrates SQ_ACTIVE_INST_VALU against a known instruction stream)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/issue_rate
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$REPO/scripts/issue_rate > $OUT/issue_rate.json 2> $OUT/issue_rate.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/sq -o sq -- $REPO/scripts/issue_rate > /dev/null 2> $OUT/sq.err
python3 $REPO/scripts/rocpd_summary.py $OUT/sq/sq_results.db $OUT/sq_sum > $OUT/sq_summary.log 2>&1
rm -rf $OUT/sq
ls -la $OUT
