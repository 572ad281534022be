set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
(timeout 700 python scripts/fuzz_parity.py 420 61000 2>&1 | tail -12) > gpurun_out/r6a/fuzz_small.log
tail -3 gpurun_out/r6a/fuzz_small.log
(timeout 500 python scripts/fuzz_parity.py 300 62000 big 2>&1 | tail -12) > gpurun_out/r6a/fuzz_big.log
tail -3 gpurun_out/r6a/fuzz_big.log
(IMPG_STAGE_DENSITY=0 timeout 500 python scripts/fuzz_parity.py 300 63000 2>&1 | tail -12) > gpurun_out/r6a/fuzz_dense.log
tail -3 gpurun_out/r6a/fuzz_dense.log
(IMPG_STAGE_DENSITY=0 timeout 400 python scripts/fuzz_parity.py 240 64000 big 2>&1 | tail -12) > gpurun_out/r6a/fuzz_dense_big.log
tail -3 gpurun_out/r6a/fuzz_dense_big.log
(IMPG_POISON=a5 timeout 400 python scripts/fuzz_parity.py 240 65000 2>&1 | tail -12) > gpurun_out/r6a/fuzz_poison.log
tail -3 gpurun_out/r6a/fuzz_poison.log
