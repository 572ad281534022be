timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout 200 python scripts/ab_locality.py 2>&1 | tail -2
timeout 300 python bench.py --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.3e ms/step %.1f stages %s' % (d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['stage_ms_per_step_rank0'].items()}))"
