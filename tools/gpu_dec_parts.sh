#!/bin/bash
# step / decode time of configs 3, 9, 2, 1 under KNZ_DEC_PARTS = 1, 2, 3 (developer tool; run through gpurun)
mkdir -p gpurun_out/r7f
for C in 3 9 2 1; do for S in 1 2 3; do
  KNZ_DEC_PARTS=$S timeout 300 python bench.py --config $C --steps 5 --warmup 2 --no-cpu --no-e2e --no-real 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $C dec_parts $S', json.dumps({'ms_per_step': d['ms_per_step'], 'enc_MBps': d['enc_MBps'], 'dec_MBps': d['dec_MBps']}))" >> gpurun_out/r7f/dec_parts.txt
done; done
cat gpurun_out/r7f/dec_parts.txt
