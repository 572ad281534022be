timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout 120 python scripts/ab_locality.py 2>&1 | tail -1
timeout 200 bash scripts/trace_quick.sh 2>&1 | head -5
