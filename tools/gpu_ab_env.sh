#!/bin/bash
# A / B of one environment knob on ONE box: tools/gpu_ab_env.sh <tag> <VAR> [config ...] -- each config's step time with the variable unset and
# set to 1, three times in turn (developer tool; run through gpurun)
O=gpurun_out/$1; V=$2; shift 2; mkdir -p $O
for rep in 1 2 3; do for C in "$@"; do for X in "" 1; do
  env ${X:+$V=$X} timeout 300 python bench.py --config $C --steps 6 --warmup 2 --no-cpu --no-e2e --no-real 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $C $V=${X:-unset}', d['ms_per_step'], d['enc_MBps'], d['dec_MBps'], d['roofline']['stages_ms']['bwt_forward'])" | tee -a $O/ab.txt
done; done; done
