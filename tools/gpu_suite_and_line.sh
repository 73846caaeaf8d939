#!/bin/bash
# the GPU suite, then the default bench line (config 3 + real_bytes) and the 4 / 7 / 13 block limits (developer tool; run through gpurun)
O=gpurun_out/${1:-r8x}; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 > $O/bench3.json 2> $O/bench3.err; echo "bench3 rc=$?" | tee -a $O/summary.txt
python - <<PY
import json
d=json.loads(open("$O/bench3.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("config3 ms", d["ms_per_step"], "MB/s", d["value"], "enc", d["enc_MBps"], "dec", d["dec_MBps"], "exact", d["bit_exact_vs_reference"])
print("stages", r["stages_ms"])
print("top kernels", dict(list(r["kernels_ms"].items())[:16]))
rb=d.get("real_bytes"); print("real", rb and {k: rb[k] for k in ("value","ms_per_step","bwt_forward_stage_ms","bit_exact_vs_reference")})
print("e2e", d.get("end_to_end", {}).get("value"))
PY
for L in 33554432 58720256 109051904; do
  timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu --no-e2e --no-real --limit $L 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'blocks': d['config']['blocks'], 'ms_per_step': d['ms_per_step'], 'enc_MBps': d['enc_MBps'], 'dec_MBps': d['dec_MBps']}))" | tee -a $O/limits.jsonl
done
