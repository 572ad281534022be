#!/bin/bash
# Round 4's closing evidence on one MI355X (run through gpurun): the headline command under rocprofv3 (kernel trace, then
# FETCH_SIZE / WRITE_SIZE / SQ passes -- the EA calibration pass and the LDS pass are scripts/profile_r4.sh's), the JSON
# summaries bench.py's roofline block reads, the default bench line, one rank through the sharded path, configs 4 and 5,
# and the GPU suite.  Everything lands in gpurun_out/final_r4/ (copy what is to be judged into profiles/).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
F=$REPO/gpurun_out/final_r4
mkdir -p $F
export TAG=r4_final
PASSES="${PASSES:-trace fetch write sq}" bash scripts/profile_r2.sh > $F/profile.log 2>&1
OUT=$REPO/gpurun_out/prof_$TAG
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o /tmp/kernels.s impg_amd/csrc/kernels.hip 2>/dev/null
python3 scripts/valu_mix.py --asm /tmp/kernels.s --kernel project_entries_kernelILb1 --json $OUT/valu_mix_entries.json > /dev/null
python3 scripts/make_traffic_json.py $OUT $OUT/traffic.json > /dev/null
python3 scripts/make_sq_json.py $OUT $OUT/sq.json $OUT/valu_mix_entries.json > /dev/null
cp $OUT/traffic.json $REPO/profiles/r4_final_traffic.json 2>/dev/null  # (so that the bench lines below price against this build's counters)
cp $OUT/sq.json $REPO/profiles/r4_final_sq.json 2>/dev/null
timeout 400 python bench.py > $F/bench_full.json 2> $F/bench_full.err
timeout 300 python bench.py --force-sharded --steps 5 --warmup 2 --cpu-sample 0 --no-extras > $F/bench_sharded_1rank.json 2> $F/bench_sharded_1rank.err
timeout 400 python bench.py --workload config4 --cpu-sample 0 --no-extras > $F/bench_config4.json 2> $F/bench_config4.err
timeout 300 python bench.py --workload config5 --ranges 20000 --steps 2 --warmup 1 --cpu-sample 0 --no-extras > $F/bench_config5_20000.json 2> $F/bench_config5_20000.err
if [ -n "${CONFIG5_FULL:-}" ]; then  # BASELINE config 5 at its size: 10^6 windows, ~5 min
  timeout 900 python bench.py --workload config5 --ranges 1000000 --steps 1 --warmup 0 --cpu-sample 0 --no-extras > $F/bench_config5_1e6.json 2> $F/bench_config5_1e6.err
fi
(timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -20) > $F/gputest.log
tail -4 $F/gputest.log
for f in bench_full bench_sharded_1rank bench_config4 bench_config5_20000 bench_config5_1e6; do python3 -c "
import json,sys
try:
    d=json.loads(open('$F/$f.json').read().strip().splitlines()[-1]); print('$f', '%.4g' % d['value'], '%.2f ms' % d['ms_per_step'], d.get('stage_ms_per_step_rank0'), (d.get('roofline') or {}).get('measured_traffic_frac'), (d.get('roofline') or {}).get('valu_issue_frac'), d.get('parity_vs_single'))
except Exception as e: print('$f', 'FAILED', e)
"; done
head -14 $OUT/trace_kernel_stats.csv | cut -c1-70,150-
