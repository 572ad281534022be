#!/bin/bash
# Round 5's closing evidence on one MI355X (run through gpurun): the headline command under rocprofv3 -- kernel trace, then
# separate FETCH_SIZE / WRITE_SIZE / EA / SQ / LDS passes (scripts/profile_r2.sh) -- the JSON summaries bench.py's roofline
# block reads (projection traffic + VALU issue, and every kernel >= 1 % of a step: scripts/make_per_kernel_json.py), the
# default bench line, one rank through the sharded path, the world sweep, configs 4 and 5, and the GPU suite with its
# durations (+ the two size-limited tests at their round-4 sizes).  Everything lands in gpurun_out/final_r5/ (copy what
# is to be judged into profiles/).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
F=$REPO/gpurun_out/final_r5
mkdir -p $F
export TAG=r5_final
if [ -z "${SKIP_PROFILE:-}" ]; then
PASSES="${PASSES:-trace fetch write ea sq}" bash scripts/profile_r2.sh > $F/profile.log 2>&1
OUT=$REPO/gpurun_out/prof_$TAG
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE \
  -d $OUT/lds -o lds -- python $REPO/bench.py --ranges 100000 --steps 1 --warmup 1 --cpu-sample 0 --no-extras > $OUT/lds_bench.json 2> $OUT/lds.err
  python3 $REPO/scripts/rocpd_summary.py $OUT/lds/lds_results.db $OUT/lds; rm -rf $OUT/lds )
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o /tmp/kernels.s impg_amd/csrc/kernels.hip 2>/dev/null
python3 scripts/valu_mix.py --asm /tmp/kernels.s --kernel project_entries_kernelILb1 --json $OUT/valu_mix_entries.json > /dev/null
python3 scripts/make_traffic_json.py $OUT $OUT/traffic.json > /dev/null
python3 scripts/make_sq_json.py $OUT $OUT/sq.json $OUT/valu_mix_entries.json > /dev/null
python3 scripts/make_per_kernel_json.py $OUT $OUT/per_kernel.json --asm /tmp/kernels.s > $F/per_kernel.txt
for f in traffic sq per_kernel; do cp $OUT/$f.json $REPO/profiles/r5_final_$f.json 2>/dev/null; done  # (so that the bench lines below price against this build's counters)
fi
timeout 500 python bench.py > $F/bench_full.json 2> $F/bench_full.err
timeout 300 python bench.py --force-sharded --steps 5 --warmup 2 --cpu-sample 0 --no-extras > $F/bench_sharded_1rank.json 2> $F/bench_sharded_1rank.err
timeout 400 python bench.py --world-sweep --no-extras --cpu-sample 0 --steps 3 > $F/bench_world_sweep.json 2> $F/bench_world_sweep.err
timeout 400 python bench.py --workload config4 --cpu-sample 0 --no-extras > $F/bench_config4.json 2> $F/bench_config4.err
timeout 300 python bench.py --workload config5 --ranges 20000 --steps 2 --warmup 1 --cpu-sample 0 --no-extras > $F/bench_config5_20000.json 2> $F/bench_config5_20000.err
timeout 300 python bench.py --min-identity 0.9 --steps 3 --cpu-sample 0 --no-extras > $F/bench_min_identity.json 2> $F/bench_min_identity.err
if [ -n "${CONFIG5_FULL:-}" ]; then  # BASELINE config 5 at its size: 10^6 windows, ~5 min
  timeout 900 python bench.py --workload config5 --ranges 1000000 --steps 1 --warmup 0 --cpu-sample 0 --no-extras > $F/bench_config5_1e6.json 2> $F/bench_config5_1e6.err
fi
if [ -n "${CONFIG4_FULL:-}" ]; then  # BASELINE config 4 at its size on ONE device: 10^8 records with prefix lines (222 GB)
  timeout 900 python bench.py --workload config4 --records 100000000 --steps 3 --warmup 1 > $F/bench_config4_1e8.json 2> $F/bench_config4_1e8.err
fi
(timeout 1200 python -m pytest tests -m gpu -x -q --durations=20 2>&1 | tail -34) > $F/gputest.log
tail -3 $F/gputest.log
(IMPG_CONFIG4_RECORDS=5e7 IMPG_CONFIG5_WINDOWS=2e5 timeout 900 python -m pytest tests/test_gpu_config45.py -m gpu -x -q --durations=4 2>&1 | tail -12) > $F/gputest_config45_full_size.log
tail -3 $F/gputest_config45_full_size.log
for f in bench_full bench_sharded_1rank bench_world_sweep bench_config4 bench_config5_20000 bench_min_identity bench_config5_1e6 bench_config4_1e8; do python3 -c "
import json,sys
try:
    d=json.loads(open('$F/$f.json').read().strip().splitlines()[-1]); print('$f', '%.4g' % d['value'], '%.2f ms' % d['ms_per_step'], d.get('stage_ms_per_step_rank0'), d.get('self_check'), (d.get('roofline') or {}).get('measured_traffic_frac'), (d.get('roofline') or {}).get('valu_issue_frac'), d.get('parity_vs_single'))
except Exception as e: print('$f', 'FAILED', e)
"; done
cat $F/per_kernel.txt 2>/dev/null | cut -c1-120
