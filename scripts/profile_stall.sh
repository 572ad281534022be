#!/bin/bash
# Where project_kernel's cycles go (run on the GPU box via gpurun): separate rocprofv3 --pmc passes over the headline
# command -- wave wait/issue cycles, the texture addresser, the vector L1 and the L2.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-stall}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--ranges ${RANGES:-100000} --steps 1 --warmup 1 --cpu-sample 0 --no-extras"
run() {
  local name=$1; shift
  timeout 600 rocprofv3 "$@" -d $OUT/$name -o $name -- python $REPO/bench.py $ARGS > $OUT/${name}_bench.json 2> $OUT/$name.err
  python3 $REPO/scripts/rocpd_summary.py $OUT/$name/${name}_results.db $OUT/$name
  rm -rf $OUT/$name
}
run sqwait --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM
run ta --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
run tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_GATE_EN1_sum TCP_TOTAL_ACCESSES_sum
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TD_TD_BUSY_sum TD_TC_STALL_sum
grep -h "project_kernel\|^Name\|^Kernel" $OUT/*_pmc.csv | cut -c1-40,140-
