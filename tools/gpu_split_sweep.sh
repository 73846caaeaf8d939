#!/bin/bash
# step time of configs 3 and 9 under KNZ_BWT_SPLIT = 1, 2, 3 (developer tool; run through gpurun)
mkdir -p gpurun_out/r6r
for C in 3 9; do for S in 1 2 3; do
  KNZ_BWT_SPLIT=$S timeout 300 python bench.py --config $C --steps 5 --warmup 2 --no-cpu --no-e2e --no-real 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $C split $S', json.dumps({'ms_per_step': d['ms_per_step'], 'enc_MBps': d['enc_MBps'], 'dec_MBps': d['dec_MBps'], 'bwt_forward_stage_ms': d['roofline']['stages_ms'].get('bwt_forward')}))" >> gpurun_out/r6r/split.txt
done; done
cat gpurun_out/r6r/split.txt
