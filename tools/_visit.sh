OUT=gpurun_out/r05j; mkdir -p $OUT
for sp in 2 3 4; do
KNZ_BWT_SPLIT=$sp timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $sp', d['value'], d['ms_per_step'], d['enc_MBps'], d['dec_MBps'])"
done
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench3.json 2> $OUT/bench3.err; python -c "
import json
d=json.loads(open('$OUT/bench3.json').read().strip().splitlines()[-1]); print('bench3', d['value'], d['ms_per_step'], d['cpu_baseline']['value'], d['bit_exact_vs_reference'], d['end_to_end'])"
