#!/bin/bash
mkdir -p gpurun_out/r7i
for M in 4 3 2; do for L in 33554432 58720256 109051904; do
  KNZ_DEC_PART_MIN=$M timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu --no-e2e --no-real --limit $L 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('part_min $M', json.dumps({'blocks': d['config']['blocks'], 'ms_per_step': d['ms_per_step'], 'enc_MBps': d['enc_MBps'], 'dec_MBps': d['dec_MBps']}))" >> gpurun_out/r7i/decmin.txt
done; done
cat gpurun_out/r7i/decmin.txt
