cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 560 python scripts/fuzz_parity.py 500 90000 > gpurun_out/r5_soak_a.log 2>&1; tail -n 1 gpurun_out/r5_soak_a.log
timeout 460 python scripts/fuzz_parity.py 400 91000 big > gpurun_out/r5_soak_b.log 2>&1; tail -n 1 gpurun_out/r5_soak_b.log
IMPG_POISON=a5 timeout 360 python scripts/fuzz_parity.py 300 92000 > gpurun_out/r5_soak_c.log 2>&1; tail -n 1 gpurun_out/r5_soak_c.log
IMPG_STAGE_DENSITY=0 timeout 360 python scripts/fuzz_parity.py 300 93000 big > gpurun_out/r5_soak_d.log 2>&1; tail -n 1 gpurun_out/r5_soak_d.log
bash scripts/soak_multi.sh 4 | tail -1 | cut -c1-300
