OUT=gpurun_out/r05h; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8
run() { # name config lanes batch env...
  n=$1; shift
  timeout 300 python tools/host_e2e_sweep.py "$@" > $OUT/$n.jsonl 2> $OUT/$n.err; echo "== $n $* rc=$?"; cat $OUT/$n.jsonl
  grep "knz \(out\|in\)" $OUT/$n.err | tail -2 | cut -c1-330
}
run g3 3 6 2,3,4,6 KNZ_DEVICE_CONCURRENCY=3
run g2 3 6 3,4,6 KNZ_DEVICE_CONCURRENCY=2
run g2l4 3 4 3,4 KNZ_DEVICE_CONCURRENCY=2
run g1 3 4 4,6,9 KNZ_DEVICE_CONCURRENCY=1
