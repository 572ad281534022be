set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "identity or prefix_line or kats or random_query or random_transitive or counting_runs or synthetic_config or projection_order" 2>&1 | tail -5 > gpurun_out/r4a/tests.log
cat gpurun_out/r4a/tests.log
RANGES=100000 bash scripts/ab_bench.sh libimpg_r3.so libimpg_gpu.so 2>&1 | tee gpurun_out/r4a/ab.log
timeout 300 scripts/gather_rate > gpurun_out/r4a/gather_rate.json 2> gpurun_out/r4a/gather_rate.err
tail -c 600 gpurun_out/r4a/gather_rate.json
