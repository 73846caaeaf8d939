timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bwt or transform_stage or stream_golden or config3 or ragged or long_common or corrupted or real_files" 2>&1 | tail -3
for j in 0 1; do
KNZ_BWT_I_JUMP_ALL=$j timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('jumpall $j', d['value'], d['ms_per_step'], d['dec_MBps'], 'jump', k.get('k_bwt_i_jump'), 'inv', d['roofline']['stages_ms']['bwt_inverse'])"
done
KNZ_BWT_I_JUMP_ALL=0 timeout 300 python bench.py --limit 33554432 --steps 8 --warmup 2 --no-cpu --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('  4 blocks', d['value'], d['ms_per_step'], 'jump', k.get('k_bwt_i_jump'))"
