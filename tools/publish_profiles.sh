#!/bin/bash
# Copies the summaries of a tools/gpu_round.sh visit (gpurun_out/<tag>/) into profiles/ under the round's prefix (developer tool).
# usage: tools/publish_profiles.sh <tag> <prefix>      e.g.  tools/publish_profiles.sh r03 r03
set -eu
SRC=gpurun_out/$1; P=profiles/$2
cpif() { [ -s "$1" ] && cp "$1" "$2" && echo "  $2"; return 0; }
cpif $SRC/bench3.json ${P}_bench_config3.json
cpif $SRC/bench2.json ${P}_bench_config2.json
cpif $SRC/bench1.json ${P}_bench_config1.json
cpif $SRC/bench4.json ${P}_bench_config4.json
cpif $SRC/bench5.json ${P}_bench_config5.json
cpif $SRC/bench7.json ${P}_bench_config7_text.json
cpif $SRC/bench8.json ${P}_bench_config8_repeats.json
cpif $SRC/bench3x2.json ${P}_bench_config3_2ranks_one_gpu_smoke.json
cpif $SRC/hard.txt ${P}_hard_inputs.txt
cpif $SRC/prof3_kernel_stats.txt ${P}_config3_kernel_stats.txt
cpif $SRC/prof3_kernel_stats.csv ${P}_config3_kernel_stats.csv
cpif $SRC/prof4_kernel_stats.txt ${P}_config4_kernel_stats.txt
cpif $SRC/prof4_kernel_stats.csv ${P}_config4_kernel_stats.csv
cpif $SRC/prof5_kernel_stats.txt ${P}_config5_kernel_stats.txt
cpif $SRC/prof5_kernel_stats.csv ${P}_config5_kernel_stats.csv
cpif $SRC/pmc3_traffic.txt ${P}_config3_pmc_traffic.txt
cpif $SRC/pmc_calibration.txt ${P}_pmc_calibration.txt
cpif $SRC/trace3_bwt_forward.txt ${P}_config3_bwt_forward_trace.txt
cpif $SRC/limits.jsonl ${P}_config3_limits.jsonl
cpif $SRC/issue_rates.txt ${P}_issue_rates.txt
for f in $SRC/pmc*_traffic.txt; do
  [ -s "$f" ] || continue
  c=$(basename $f | sed 's/pmc\([0-9]*\)_traffic.txt/\1/'); [ "$c" = "3" ] || cpif $f ${P}_config${c}_pmc_traffic.txt
done
for f in $SRC/pmc*_traffic.json; do
  [ -s "$f" ] || continue
  python - "$f" <<'PY'
import json, sys
new = json.load(open(sys.argv[1]))
try:
    cur = json.load(open("profiles/pmc_traffic.json"))
except Exception:
    cur = {}
cur.update(new)
json.dump(cur, open("profiles/pmc_traffic.json", "w"), indent=1, sort_keys=True)
print("  profiles/pmc_traffic.json (%s)" % ", ".join(sorted(new)))
PY
done
