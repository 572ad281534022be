#!/bin/bash
# Round 6's closing evidence on one MI355X (run through gpurun): the headline command's timed form (rows left in HBM,
# attributed layout: --no-tiers) under rocprofv3 -- kernel trace, then separate FETCH_SIZE / WRITE_SIZE / EA / SQ / LDS passes
# (scripts/profile_r2.sh) -- the JSON summaries bench.py's roofline block reads, the default bench line with its three tiers,
# configs 4 / 5, the skewed workload (+ its kernel trace), one rank through the sharded path, --min-identity, and the GPU
# suite with its durations.  Everything lands in gpurun_out/final_r6/ (copy what is to be judged into profiles/).
# PARTS: space-separated subset of "profile config4prof bench ordered config45full suite" (default: the first three and the suite).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
F=$REPO/gpurun_out/final_r6
mkdir -p $F
PARTS=${PARTS:-profile config4prof bench suite}
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
asm() {
  [ -f /tmp/kernels.s ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o /tmp/kernels.s impg_amd/csrc/kernels.hip 2>/dev/null
}
if has profile; then
  export TAG=r6_final
  EXTRA_ARGS="--no-tiers" PASSES="${PASSES:-trace fetch write ea sq}" bash scripts/profile_r2.sh > $F/profile.log 2>&1
  OUT=$REPO/gpurun_out/prof_$TAG
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE \
    -d $OUT/lds -o lds -- python $REPO/bench.py --ranges 100000 --steps 1 --warmup 1 --cpu-sample 0 --no-extras --no-tiers > $OUT/lds_bench.json 2> $OUT/lds.err
    python3 $REPO/scripts/rocpd_summary.py $OUT/lds/lds_results.db $OUT/lds; rm -rf $OUT/lds )
  asm
  python3 scripts/valu_mix.py --asm /tmp/kernels.s --kernel project_entries_kernelILb1ELi0ELi1 --json $OUT/valu_mix_entries.json > /dev/null
  python3 scripts/make_traffic_json.py $OUT $OUT/traffic.json > /dev/null
  python3 scripts/make_sq_json.py $OUT $OUT/sq.json $OUT/valu_mix_entries.json > /dev/null
  python3 scripts/make_per_kernel_json.py $OUT $OUT/per_kernel.json --asm /tmp/kernels.s > $F/per_kernel.txt
  for f in traffic sq per_kernel; do cp $OUT/$f.json $REPO/profiles/r6_final_$f.json 2>/dev/null; done  # (the bench lines below price against this build's counters)
  for f in trace_kernel_stats fetch_pmc write_pmc ea_pmc sq_pmc lds_pmc; do cp $OUT/$f.csv $F/r6_final_$f.csv 2>/dev/null; done
  cp $OUT/valu_mix_entries.json $F/r6_final_valu_mix_entries.json 2>/dev/null
  cp $OUT/trace_bench.json $F/r6_final_trace_bench.json 2>/dev/null
fi
if has config4prof; then  # config 4's counters on THIS round's index (the traffic fraction of its bench line), EA pass included
  export TAG=r6_config4
  EXTRA_ARGS="--workload config4 --form count" RANGES=100000 PASSES="trace fetch write ea sq" PASS_TIMEOUT=900 bash scripts/profile_r2.sh > $F/profile_config4.log 2>&1
  OUT=$REPO/gpurun_out/prof_$TAG
  asm
  python3 scripts/valu_mix.py --asm /tmp/kernels.s --kernel project_kernelILb1ELi0 --json $OUT/valu_mix_project.json > /dev/null
  python3 scripts/make_traffic_json.py $OUT $OUT/traffic.json > /dev/null
  python3 scripts/make_sq_json.py $OUT $OUT/sq.json $OUT/valu_mix_project.json > /dev/null
  for f in traffic sq; do cp $OUT/$f.json $REPO/profiles/r6_config4_$f.json 2>/dev/null; done
  for f in trace_kernel_stats fetch_pmc write_pmc ea_pmc sq_pmc; do cp $OUT/$f.csv $F/r6_config4_$f.csv 2>/dev/null; done
fi
if has bench; then
  timeout 600 python bench.py > $F/bench_full.json 2> $F/bench_full.err
  timeout 300 python bench.py --form count --no-extras --cpu-sample 0 --steps 5 > $F/bench_count_form.json 2> $F/bench_count_form.err
  timeout 300 python bench.py --force-sharded --steps 5 --warmup 2 --cpu-sample 0 --no-extras > $F/bench_sharded_1rank.json 2> $F/bench_sharded_1rank.err
  timeout 400 python bench.py --workload config4 --cpu-sample 0 --no-extras > $F/bench_config4.json 2> $F/bench_config4.err
  timeout 300 python bench.py --workload config5 --ranges 20000 --steps 2 --warmup 1 --cpu-sample 0 --no-extras > $F/bench_config5_20000.json 2> $F/bench_config5_20000.err
  timeout 300 python bench.py --workload skewed --steps 3 --warmup 1 --cpu-sample 0 --no-extras > $F/bench_skewed.json 2> $F/bench_skewed.err
  timeout 300 python bench.py --workload skewed --records 1000000 --steps 2 --warmup 1 --cpu-sample 0 --no-extras > $F/bench_skewed_1e6.json 2> $F/bench_skewed_1e6.err
  STEPS=2 TOP=14 BENCH_ARGS="--workload skewed" bash scripts/trace_r5.sh final_r6/skewed_trace > $F/skewed_trace.txt 2>&1
  timeout 300 python bench.py --min-identity 0.9 --steps 3 --cpu-sample 0 --no-extras --no-tiers > $F/bench_min_identity.json 2> $F/bench_min_identity.err
  if [ -n "${CONFIG5_FULL:-}" ]; then
    timeout 900 python bench.py --workload config5 --ranges 1000000 --steps 1 --warmup 0 --cpu-sample 0 --no-extras > $F/bench_config5_1e6.json 2> $F/bench_config5_1e6.err
  fi
  if [ -n "${CONFIG4_FULL:-}" ]; then
    timeout 900 python bench.py --workload config4 --records 100000000 --form count --steps 3 --warmup 1 --no-extras --cpu-sample 0 > $F/bench_config4_1e8.json 2> $F/bench_config4_1e8.err
  fi
  for f in bench_full bench_count_form bench_sharded_1rank bench_config4 bench_config5_20000 bench_skewed bench_skewed_1e6 bench_min_identity bench_config5_1e6 bench_config4_1e8; do python3 -c "
import json,sys
try:
    d=json.loads(open('$F/$f.json').read().strip().splitlines()[-1]); print('$f', '%.4g' % d['value'], '%.2f ms' % d['ms_per_step'], d.get('stage_ms_per_step_rank0'), d.get('self_check'), (d.get('roofline') or {}).get('measured_traffic_frac'), (d.get('roofline') or {}).get('valu_issue_frac'), d.get('parity_vs_single'), d.get('value_count_only'), d.get('value_ordered_rows_device'))
except Exception as e: print('$f', 'FAILED', e)
"; done
fi
if has ordered; then  # the ordered-rows tier as the timed step, under the kernel trace
  TAG=r6_ordered STEPS=2 WARMUP=1 PASSES="trace" EXTRA_ARGS="--form ordered" bash scripts/profile_r2.sh > $F/ordered_trace.log 2>&1
  cp $REPO/gpurun_out/prof_r6_ordered/trace_kernel_stats.csv $F/r6_ordered_trace_kernel_stats.csv 2>/dev/null
  cp $REPO/gpurun_out/prof_r6_ordered/trace_bench.json $F/r6_ordered_trace_bench.json 2>/dev/null
fi
if has config45full; then  # the config-4 / config-5 parity tests at the sizes the suite's defaults were cut from (round 5's review)
  (IMPG_CONFIG4_RECORDS=5e7 IMPG_CONFIG5_WINDOWS=2e5 timeout 1500 python -m pytest tests/test_gpu_config45.py -m gpu -x -q --durations=8 2>&1 | tail -16) > $F/gputest_config45_full.log
  tail -3 $F/gputest_config45_full.log
fi
if has suite; then
  (timeout 1500 python -m pytest tests -m gpu -x -q --durations=20 2>&1 | tail -34) > $F/gputest.log
  tail -3 $F/gputest.log
fi
cat $F/per_kernel.txt 2>/dev/null | cut -c1-120
