#!/bin/bash
# One-off GPU-box visits of round 5 (developer tool; run through gpurun): tools/gpu_visit.sh <tag> <what>
set -u
TAG=${1:-v}; WHAT=${2:-host}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
case "$WHAT" in
  host)   # where the host layer's wall time goes, and what lanes / batch sizes do to it
    timeout 300 python tools/host_e2e_sweep.py 3 4 0,4,8 > $OUT/e2e3.jsonl 2> $OUT/e2e3.err; echo "e2e3 rc=$?" >> $OUT/summary.txt
    timeout 300 python tools/host_e2e_sweep.py 2 4 0,8,16 > $OUT/e2e2.jsonl 2> $OUT/e2e2.err; echo "e2e2 rc=$?" >> $OUT/summary.txt
    cat $OUT/e2e3.jsonl $OUT/e2e2.jsonl; grep "knz \(out\|in\)\|^----" $OUT/e2e3.err $OUT/e2e2.err | cut -c1-330 ;;
  small)  # 4-block step with the per-kernel table
    timeout 300 python bench.py --limit 33554432 --steps 5 --warmup 2 --no-cpu --no-e2e > $OUT/bench3_4blocks.json 2> $OUT/bench3_4blocks.err; echo "small rc=$?" >> $OUT/summary.txt
    python -c "
import json,sys
d=json.loads(open('$OUT/bench3_4blocks.json').read().strip().splitlines()[-1])
print('4 blocks ms_per_step', d['ms_per_step'], 'enc', d['enc_MBps'], 'dec', d['dec_MBps'])
print(json.dumps(d['roofline']['stages_ms']))
print(json.dumps(dict(list(d['roofline']['kernels_ms'].items())[:40])))
" ;;
esac
cat $OUT/summary.txt
