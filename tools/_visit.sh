timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mtft or bwt or transform_stage or stream_golden or config3 or ragged or long_common" 2>&1 | tail -3
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('bench3', d['value'], d['ms_per_step'], d['enc_MBps'], d['dec_MBps'], 'rank', k.get('k_mtf_f_rank'), 'last', k.get('k_mtf_f_last'), 'place', k.get('k_bwt_i_place'), d['roofline']['stages_ms'])"
timeout 300 python bench.py --limit 33554432 --steps 8 --warmup 2 --no-cpu --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('  4 blocks', d['value'], d['ms_per_step'], 'place', k.get('k_bwt_i_place'))"
