#!/bin/bash
# rocprofv3 evidence for round 4 (run on the GPU box via gpurun): the HEADLINE command itself (100 000 ranges, -x -m 3):
# kernel trace, then separate PMC passes (scripts/profile_r2.sh) -- FETCH_SIZE, WRITE_SIZE, the raw L2->fabric request
# counters that calibrate them, the SQ instruction / cycle counters, and the LDS / wait counters of the staged kernels;
# then the JSON summaries bench.py's roofline block reads.  Summaries land in gpurun_out/prof_r4_final/.
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
export TAG=r4_final
PASSES="${PASSES:-trace fetch write ea sq}" bash scripts/profile_r2.sh
OUT=$REPO/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE \
  -d $OUT/lds -o lds -- python $REPO/bench.py --ranges 100000 --steps 1 --warmup 1 --cpu-sample 0 --no-extras > $OUT/lds_bench.json 2> $OUT/lds.err
python3 $REPO/scripts/rocpd_summary.py $OUT/lds/lds_results.db $OUT/lds; rm -rf $OUT/lds
cd $REPO
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o /tmp/kernels.s impg_amd/csrc/kernels.hip 2>/dev/null
python3 scripts/valu_mix.py --asm /tmp/kernels.s --kernel project_entries_kernelILb1 --json $OUT/valu_mix_entries.json > /dev/null
python3 scripts/make_traffic_json.py $OUT $OUT/traffic.json > /dev/null
python3 scripts/make_sq_json.py $OUT $OUT/sq.json $OUT/valu_mix_entries.json > /dev/null
python3 -c "
import json
t=json.load(open('$OUT/traffic.json')); q=json.load(open('$OUT/sq.json'))
print('hbm bytes/pair', t['hbm_bytes_per_pair'], 'fetch x2', t['fetch_bytes_per_pair_corrected_x2'], 'write', t['write_bytes_per_pair'], 'rd128', t['fetch_bytes_per_pair_from_128B_requests'])
print('valu/pair', q['valu_insts_per_pair'], 'cpi', q['cycles_per_valu_inst'], 'valu_issue_frac', q['valu_issue_frac'])"
head -12 $OUT/trace_kernel_stats.csv | cut -c1-70,150-
