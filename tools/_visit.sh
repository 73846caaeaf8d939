for r in 2 3; do
KNZ_BWT_I_RULER_LOG=$r timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('rlog $r', d['value'], d['ms_per_step'], d['dec_MBps'], 'jump', k.get('k_bwt_i_jump'), 'inv', d['roofline']['stages_ms']['bwt_inverse'])"
KNZ_BWT_I_RULER_LOG=$r timeout 300 python bench.py --limit 33554432 --steps 8 --warmup 2 --no-cpu --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('  4 blocks', d['value'], d['ms_per_step'], 'jump', k.get('k_bwt_i_jump'))"
done
