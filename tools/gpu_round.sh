#!/bin/bash
# One GPU-box visit: parity suite, bench lines, rocprofv3 kernel stats (developer tool; run through gpurun).
# usage: tools/gpu_round.sh <tag> [steps...]   steps: tests bench1..5,7 prof3..5 pmc3 trace3 calib limits issue srtprobe cmd:<shell>
set -u
TAG=${1:-run}; shift || true
[ $# -eq 0 ] && set -- tests bench3
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for s in "$@"; do
  [ -n "${DRY:-}" ] && { echo "step: [$s]"; continue; }
  case "$s" in
    tests_bwt) timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bwt or transform or stream_golden or ragged or config3 or soak or jobs" < /dev/null > $OUT/pytest_bwt.log 2>&1; echo "tests_bwt rc=$?" >> $OUT/summary.txt; tail -15 $OUT/pytest_bwt.log ;;
    tests) timeout 1500 python -m pytest tests -m gpu -x -q < /dev/null > $OUT/pytest.log 2>&1; echo "tests rc=$?" >> $OUT/summary.txt; tail -5 $OUT/pytest.log ;;
    tests_fast) timeout 900 python -m pytest tests -m gpu -x -q -k "not full_size and not soak and not fpaq" > $OUT/pytest_fast.log 2>&1; echo "tests_fast rc=$?" >> $OUT/summary.txt; tail -5 $OUT/pytest_fast.log ;;
    bench3) timeout 600 python bench.py --steps 5 --warmup 2 > $OUT/bench3.json 2> $OUT/bench3.err; echo "bench3 rc=$?" >> $OUT/summary.txt; cat $OUT/bench3.json ;;
    bench3q) timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-e2e > $OUT/bench3q.json 2> $OUT/bench3q.err; echo "bench3q rc=$?" >> $OUT/summary.txt; cat $OUT/bench3q.json ;;
    bench2) timeout 600 python bench.py --config 2 --steps 10 --warmup 3 > $OUT/bench2.json 2> $OUT/bench2.err; echo "bench2 rc=$?" >> $OUT/summary.txt; cat $OUT/bench2.json ;;
    bench1) timeout 600 python bench.py --config 1 --steps 10 --warmup 3 --no-e2e > $OUT/bench1.json 2> $OUT/bench1.err; echo "bench1 rc=$?" >> $OUT/summary.txt; cat $OUT/bench1.json ;;
    bench3x2) timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --dist-backend gloo --share-device > $OUT/bench3x2.json 2> $OUT/bench3x2.err; echo "bench3x2 rc=$?" >> $OUT/summary.txt; tail -2 $OUT/bench3x2.json ;;
    bench7) timeout 600 python bench.py --config 7 --limit 211957760 --steps 3 --warmup 1 --no-e2e > $OUT/bench7.json 2> $OUT/bench7.err; echo "bench7 rc=$?" >> $OUT/summary.txt; cat $OUT/bench7.json ;;
    bench8) timeout 600 python bench.py --config 8 --steps 3 --warmup 1 --no-e2e ${BENCH8_ARGS:-} > $OUT/bench8.json 2> $OUT/bench8.err; echo "bench8 rc=$?" >> $OUT/summary.txt; cat $OUT/bench8.json ;;
    bench8q) timeout 600 python bench.py --config 8 --steps 3 --warmup 1 --no-e2e --no-cpu > $OUT/bench8q.json 2> $OUT/bench8q.err; echo "bench8q rc=$?" >> $OUT/summary.txt; cat $OUT/bench8q.json ;;
    hard) timeout 420 python tools/gpu_hard.py > $OUT/hard.txt 2>&1; echo "hard rc=$?" >> $OUT/summary.txt; cat $OUT/hard.txt ;;
    bench4) timeout 900 python bench.py --config 4 --limit 134217728 --steps 1 --warmup 1 --no-e2e > $OUT/bench4.json 2> $OUT/bench4.err; echo "bench4 rc=$?" >> $OUT/summary.txt; cat $OUT/bench4.json ;;
    bench5) timeout 900 python bench.py --config 5 --steps 2 --warmup 1 --no-e2e > $OUT/bench5.json 2> $OUT/bench5.err; echo "bench5 rc=$?" >> $OUT/summary.txt; cat $OUT/bench5.json ;;
    prof3) (cd /tmp && export KNZ_BWT_SPLIT=1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof3 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --no-real > $GRAFT_REPO_ROOT/$OUT/prof3.log 2>&1); echo "prof3 rc=$?" >> $OUT/summary.txt
           f=$(find $OUT/prof3 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python tools/rocprof_csv_summary.py $f > $OUT/prof3_kernel_stats.txt && cp $f $OUT/prof3_kernel_stats.csv && head -30 $OUT/prof3_kernel_stats.txt ;;
    prof2) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof2 -- python $GRAFT_REPO_ROOT/bench.py --config 2 --steps 5 --warmup 2 --no-cpu --no-e2e > $GRAFT_REPO_ROOT/$OUT/prof2.log 2>&1); echo "prof2 rc=$?" >> $OUT/summary.txt
           f=$(find $OUT/prof2 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python tools/rocprof_csv_summary.py $f > $OUT/prof2_kernel_stats.txt && cp $f $OUT/prof2_kernel_stats.csv && head -20 $OUT/prof2_kernel_stats.txt ;;
    pmc3) for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && export KNZ_BWT_SPLIT=1 && timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc3_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-real > $GRAFT_REPO_ROOT/$OUT/pmc3_$c.log 2>&1); echo "pmc3 $c rc=$?" >> $OUT/summary.txt; done
          ff=$(find $OUT/pmc3_FETCH_SIZE -name '*counter_collection.csv' | head -1); fw=$(find $OUT/pmc3_WRITE_SIZE -name '*counter_collection.csv' | head -1)
          [ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_stage_summary.py $ff $fw 3 $OUT/pmc3_traffic.json $OUT/pmc3_traffic.txt && cat $OUT/pmc3_traffic.txt | head -40 ;;
    trace3) (cd /tmp && export KNZ_BWT_SPLIT=1 && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace3 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-e2e --no-real > $GRAFT_REPO_ROOT/$OUT/trace3.log 2>&1); echo "trace3 rc=$?" >> $OUT/summary.txt
           f=$(find $OUT/trace3 -name '*kernel_trace.csv' | head -1); [ -n "$f" ] && python tools/rocprof_trace_list.py $f 1 > $OUT/trace3_bwt_forward.txt && tail -3 $OUT/trace3_bwt_forward.txt ;;
    prof4) (cd /tmp && export KNZ_BWT_SPLIT=1 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof4 -- python $GRAFT_REPO_ROOT/bench.py --config 4 --limit 134217728 --steps 1 --warmup 1 --no-cpu --no-e2e > $GRAFT_REPO_ROOT/$OUT/prof4.log 2>&1); echo "prof4 rc=$?" >> $OUT/summary.txt
           f=$(find $OUT/prof4 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python tools/rocprof_csv_summary.py $f > $OUT/prof4_kernel_stats.txt && cp $f $OUT/prof4_kernel_stats.csv && head -14 $OUT/prof4_kernel_stats.txt ;;
    prof5) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof5 -- python $GRAFT_REPO_ROOT/bench.py --config 5 --steps 1 --warmup 1 --no-cpu --no-e2e > $GRAFT_REPO_ROOT/$OUT/prof5.log 2>&1); echo "prof5 rc=$?" >> $OUT/summary.txt
           f=$(find $OUT/prof5 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python tools/rocprof_csv_summary.py $f > $OUT/prof5_kernel_stats.txt && cp $f $OUT/prof5_kernel_stats.csv && head -14 $OUT/prof5_kernel_stats.txt ;;
    calib) tools/bin/membench > $OUT/membench.txt 2>&1
           for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/calib_$c -- $GRAFT_REPO_ROOT/tools/bin/membench pmc > /dev/null 2>&1); echo "calib $c rc=$?" >> $OUT/summary.txt; done
           ff=$(find $OUT/calib_FETCH_SIZE -name '*counter_collection.csv' | head -1); fw=$(find $OUT/calib_WRITE_SIZE -name '*counter_collection.csv' | head -1)
           [ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_calibration.py $ff $fw $OUT/membench.txt > $OUT/pmc_calibration.txt && head -24 $OUT/pmc_calibration.txt ;;
    limits) for L in 33554432 58720256 109051904 211957760; do timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu --no-e2e --no-real --limit $L 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'blocks': d['config']['blocks'], 'bytes': d['config']['corpus_bytes'], 'ms_per_step': d['ms_per_step'], 'MBps': d['value'], 'enc_MBps': d['enc_MBps'], 'dec_MBps': d['dec_MBps']}))" >> $OUT/limits.jsonl; done; echo "limits rc=$?" >> $OUT/summary.txt; cat $OUT/limits.jsonl ;;
    issue) timeout 120 tools/bin/issuebench > $OUT/issue_rates.txt 2>&1; echo "issue rc=$?" >> $OUT/summary.txt; cat $OUT/issue_rates.txt ;;
    srtprobe) timeout 200 python tools/srt_probe.py 32 > $OUT/srt_probe.txt 2>&1; echo "srtprobe rc=$?" >> $OUT/summary.txt; tail -4 $OUT/srt_probe.txt ;;
    pmcx:*) # pmcx:<config>:<bytes or 0>: two counter passes of one configuration -> $OUT/pmc<config>_traffic.{json,txt}
          IFS=: read -r _ C LIM <<< "$s"; LARG=""; NB=211957760
          [ "$LIM" != "0" ] && [ -n "$LIM" ] && { LARG="--limit $LIM"; NB=$LIM; }
          [ "$C" = "4" ] && [ -z "$LARG" ] && NB=1000000000
          for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && export KNZ_BWT_SPLIT=${PMC_SPLIT:-1} && timeout 1500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc${C}_$c -- python $GRAFT_REPO_ROOT/bench.py --config $C $LARG --steps 1 --warmup 1 --reps 1 --no-cpu --no-e2e --no-real > $GRAFT_REPO_ROOT/$OUT/pmc${C}_$c.log 2>&1); echo "pmc$C $c rc=$?" >> $OUT/summary.txt; done
          ff=$(find $OUT/pmc${C}_FETCH_SIZE -name '*counter_collection.csv' | head -1); fw=$(find $OUT/pmc${C}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
          [ -n "$ff" ] && [ -n "$fw" ] && KNZ_PMC_NBYTES=$NB python tools/pmc_stage_summary.py $ff $fw $C $OUT/pmc${C}_traffic.json $OUT/pmc${C}_traffic.txt && head -16 $OUT/pmc${C}_traffic.txt ;;
    benchx:*) # benchx:<config>:<bytes or 0>:<steps>:<extra flags with + for spaces>
          IFS=: read -r _ C LIM ST EX <<< "$s"; LARG=""; [ "$LIM" != "0" ] && [ -n "$LIM" ] && LARG="--limit $LIM"
          timeout 1500 python bench.py --config $C $LARG --steps ${ST:-3} --warmup 1 ${EX//+/ } > $OUT/bench$C.json 2> $OUT/bench$C.err; echo "bench$C rc=$?" >> $OUT/summary.txt; cut -c1-400 $OUT/bench$C.json ;;
    profx:*) IFS=: read -r _ C LIM <<< "$s"; LARG=""; [ "$LIM" != "0" ] && [ -n "$LIM" ] && LARG="--limit $LIM"
          (cd /tmp && export KNZ_BWT_SPLIT=1 && timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof$C -- python $GRAFT_REPO_ROOT/bench.py --config $C $LARG --steps 1 --warmup 1 --reps 1 --no-cpu --no-e2e --no-real > $GRAFT_REPO_ROOT/$OUT/prof$C.log 2>&1); echo "prof$C rc=$?" >> $OUT/summary.txt
          f=$(find $OUT/prof$C -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python tools/rocprof_csv_summary.py $f > $OUT/prof${C}_kernel_stats.txt && cp $f $OUT/prof${C}_kernel_stats.csv && head -14 $OUT/prof${C}_kernel_stats.txt ;;
    cmd:*) echo "running custom: ${s#cmd:}"; timeout 600 bash -c "${s#cmd:}" < /dev/null > $OUT/custom.log 2>&1; echo "custom rc=$?" >> $OUT/summary.txt; tail -40 $OUT/custom.log ;;
    *) echo "unknown step: $s" ;;
  esac
done
# rocprof output directories are large: keep only the summaries
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $OUT/summary.txt
