OUT=gpurun_out/r05u; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8
KNZ_HOST_TIMING=2 timeout 300 python tools/host_e2e_sweep.py 3 6 0 > $OUT/tl.jsonl 2> $OUT/tl.err
cat $OUT/tl.jsonl; grep "knz in" $OUT/tl.err | tail -34 | cut -c1-200
