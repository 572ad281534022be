set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "identity or prefix_line or kats or random_query or random_transitive or counting_runs or synthetic_config or projection_order or store_cigar" 2>&1 | tail -5 > gpurun_out/r4c/tests.log
cat gpurun_out/r4c/tests.log
for K in 1 2 4 8 16; do
  echo "tiles=$K"
  IMPG_PROJ_TILES=$K RANGES=100000 bash scripts/ab_bench.sh libimpg_gpu.so 2>&1 | grep rep2 | tee -a gpurun_out/r4c/ab.log
done
IMPG_PROJ_TILES=8 RANGES=100000 bash scripts/ab_bench.sh libimpg_w8.so 2>&1 | tee -a gpurun_out/r4c/ab.log
IMPG_PROJ_TILES=4 RANGES=100000 bash scripts/ab_bench.sh libimpg_w8.so 2>&1 | grep rep2 | tee -a gpurun_out/r4c/ab.log
IMPG_GPU_LIB=$GRAFT_REPO_ROOT/impg_amd/libimpg_phase.so timeout 600 python scripts/phase_clocks.py 100000 > gpurun_out/r4c/phase.json 2> gpurun_out/r4c/phase.err
cat gpurun_out/r4c/phase.json; tail -3 gpurun_out/r4c/phase.err
