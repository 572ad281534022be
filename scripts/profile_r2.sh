#!/bin/bash
# rocprofv3 evidence for round 2 (run on the GPU box via gpurun): the HEADLINE command itself
# (100 000 ranges, -x -m 3), not a sub-batch: kernel trace, then separate PMC passes -- FETCH_SIZE, WRITE_SIZE,
# the raw L2->fabric request counters that calibrate them, and the SQ instruction / cycle counters
# from which the VALU-issue fraction of project_kernel is derived.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r2}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--ranges ${RANGES:-100000} --steps ${STEPS:-1} --warmup ${WARMUP:-1} --cpu-sample 0 --no-extras ${EXTRA_ARGS:-}"
run() {  # name, rocprof args...
  local name=$1; shift
  timeout ${PASS_TIMEOUT:-600} rocprofv3 "$@" -d $OUT/$name -o $name -- python $REPO/bench.py $ARGS > $OUT/${name}_bench.json 2> $OUT/$name.err
  python3 $REPO/scripts/rocpd_summary.py $OUT/$name/${name}_results.db $OUT/$name
  rm -rf $OUT/$name
}
PASSES=${PASSES:-trace fetch write ea sq}   # (a --pmc pass of the headline step takes ~5 min: counters serialise the kernels)
for pass in $PASSES; do
  case $pass in
    trace) run trace --kernel-trace --stats ;;
    fetch) run fetch --pmc FETCH_SIZE ;;
    write) run write --pmc WRITE_SIZE ;;
    ea) run ea --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum ;;
    sq) run sq --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE ;;
  esac
done
head -8 $OUT/trace_kernel_stats.csv | cut -c1-60,140-
grep -E "project_" $OUT/*_pmc.csv | cut -c1-60,150-
