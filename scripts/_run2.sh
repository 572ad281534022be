set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "identity" 2>&1 | tail -5 > gpurun_out/r4b/tests.log
cat gpurun_out/r4b/tests.log
IMPG_GPU_LIB=$GRAFT_REPO_ROOT/impg_amd/libimpg_phase.so timeout 600 python scripts/phase_clocks.py 100000 > gpurun_out/r4b/phase.json 2> gpurun_out/r4b/phase.err
cat gpurun_out/r4b/phase.json; tail -3 gpurun_out/r4b/phase.err
