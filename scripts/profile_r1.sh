#!/bin/bash
# rocprofv3 evidence for the round (run on the GPU box via gpurun): kernel trace
# of the bench command, then separate PMC passes (FETCH_SIZE, WRITE_SIZE, and the
# raw L2->fabric request counters used to calibrate them).  rocprofv3 7.2 writes
# a rocpd SQLite database; scripts/rocpd_summary.py turns it into the CSVs kept
# under profiles/.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r1}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--ranges ${RANGES:-16384} --steps 2 --warmup 1 --cpu-sample 0"
run() {  # name, rocprof args...
  local name=$1; shift
  timeout 400 rocprofv3 "$@" -d $OUT/$name -o $name -- python $REPO/bench.py $ARGS > $OUT/${name}_bench.json 2> $OUT/$name.err
  python3 $REPO/scripts/rocpd_summary.py $OUT/$name/${name}_results.db $OUT/$name
  rm -rf $OUT/$name
}
run trace --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run ea --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
head -6 $OUT/trace_kernel_stats.csv | cut -c1-60,140-
grep -E "project_kernel" $OUT/*_pmc.csv | cut -c1-60,150-
