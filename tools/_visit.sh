timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "zrlt or transform_stage or stream_golden or config3 or ragged or capacity" 2>&1 | tail -3
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('bench3', d['value'], d['ms_per_step'], {x:k[x] for x in k if 'zrlt' in x}, d['roofline']['stages_ms'])"
