OUT=gpurun_out/r05w; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8
run() { n=$1; shift
  timeout 300 python tools/host_e2e_sweep.py "$@" > $OUT/$n.jsonl 2> $OUT/$n.err; echo "== $n $* rc=$?"; cat $OUT/$n.jsonl
  grep "knz \(out\|in\)" $OUT/$n.err | tail -2 | cut -c1-330
}
run c2 2 6,6 0
run c2old 2 6,6 0 KNZ_READ_AHEAD=1048576
run c2b 2 6 0 KNZ_READ_AHEAD=8388608
