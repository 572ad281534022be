#!/bin/bash
# rocprofv3 evidence for the round: kernel-trace stats of the bench command and
# two separate PMC passes (FETCH_SIZE, WRITE_SIZE) -- run on the GPU box via gpurun.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--ranges ${RANGES:-16384} --steps 2 --warmup 1 --cpu-sample 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python $REPO/bench.py $ARGS > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python $REPO/bench.py $ARGS > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -50
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -20 $f; done
python3 $REPO/scripts/summarize_pmc.py $OUT || true
# keep only the small summaries
find $OUT -name "*kernel_trace.csv" -size +2M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
