#!/bin/bash
# small-batch step times under split settings
mkdir -p gpurun_out/r6i
for cfg in "3 2" "4 2" "4 1" "3 1"; do
  set -- $cfg
  for L in 33554432 58720256 109051904; do
    KNZ_BWT_SPLIT=$1 KNZ_BWT_PART_MIN=$2 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-e2e --no-real --limit $L 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $1 min $2', json.dumps({'blocks': d['config']['blocks'], 'ms_per_step': d['ms_per_step'], 'enc_MBps': d['enc_MBps'], 'dec_MBps': d['dec_MBps']}))" >> gpurun_out/r6i/limits.txt
  done
done
cat gpurun_out/r6i/limits.txt
