#!/usr/bin/env python3
"""Benchmark of the kanzi block pipeline on MI355X.

Metric (BASELINE.json): encode+decode MB/s on silesia.tar, bit-exact; % of the HBM roofline.
Default workload = BASELINE config 3, the configuration the north-star target is stated on:
`-t BWT+MTFT+ZRLT -e ANS0 -b 8m` on silesia.tar (or its synthetic stand-in when the corpus is absent).

A "step" = one pass of the hot path over the corpus resident in HBM: encode every block to the kanzi
bit stream (device buffer -> device buffer) and decode that stream back (device -> device).
value = corpus bytes / (t_enc + t_dec) in MB/s (MB = 1e6 B).

Multi-GPU (--gpus N, one process per GPU under torch.distributed.run): blocks are independent, so a job is sharded over
the ranks by block index (kanzi-cpp_amd/sharded.py:block_ranges); every rank encodes and decodes its own
blocks, there is no collective in the data path (only the timing barrier and the MAX all-reduce of the elapsed time).
READ FIRST at N > 1: `one_corpus_sharded` (the ONE corpus in contiguous block ranges, with `block_count_ceiling` and
`efficiency_vs_ceiling`) and `many_blocks_sharded` (five corpora back to back = 127 blocks, ceiling 7.94 at N = 8): they answer the
metric's question, "MB/s on one input at 1/2/4/8 GPUs". The line's `value` is linear by construction:
The line's `value` is WEAK scaling, as the rule for paths that partition asks: the job grows with N -- N corpora, rank r
takes the r-th one (26 blocks each at config 3) -- and value = N * corpus bytes / MAX-over-ranks time. The same run then
also times the ONE corpus sharded over the N ranks (strong scaling: 26 blocks over 8 GPUs is 4,4,3,3,3,3,3,3, so the
best possible speed-up is 6.5x) and reports it as `one_corpus_sharded` in the same line, with that ceiling.
`--scaling strong` makes the sharded corpus the line's value instead (and the N corpora the extra object).

    python bench.py                      # N=1, config 3
    python bench.py --config 2           # -t NONE -e ANS0 -b 4m
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

# HIP maps the streams of a process onto 4 hardware queues unless told otherwise; a context of this library runs the BWT stages of a
# batch on three streams beside torch's, and the stream classes have six lanes with two copy streams each. Eight queues: +1.5 % on
# the device-resident step, +5 % end to end (profiles/r05_host_layer_sweeps.txt). An application that embeds the library sets the same
# variable before its first HIP call; the library does not touch the environment of its host process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = {
    # BASELINE.json configs (1-based)
    2: dict(transform="NONE", entropy="ANS0", block=4 << 20, corpus="silesia"),
    3: dict(transform="BWT+MTFT+ZRLT", entropy="ANS0", block=8 << 20, corpus="silesia"),
    4: dict(transform="BWT+SRT+ZRLT", entropy="FPAQ", block=32 << 20, corpus="enwik9"),
    5: dict(transform="LZX", entropy="ANS1", block=16 << 20, corpus="silesia"),
    # not BASELINE lines: config 1's codec on the device (BASELINE runs it on the CPU), config 5's entropy stage alone
    1: dict(transform="NONE", entropy="HUFFMAN", block=4 << 20, corpus="silesia"),
    6: dict(transform="NONE", entropy="ANS1", block=16 << 20, corpus="silesia"),
    # config 3's chain on text (use with --limit 211957760): how the suffix sorter does without the stand-in's periodic segments
    7: dict(transform="BWT+MTFT+ZRLT", entropy="ANS0", block=8 << 20, corpus="enwik9"),
    # config 3's chain on text with 30 % copied spans (1-64 KiB): long common prefixes, i.e. many doubling rounds of the suffix sorter
    8: dict(transform="BWT+MTFT+ZRLT", entropy="ANS0", block=8 << 20, corpus="repeats"),
    # config 3's chain on REAL bytes: a deterministic concatenation of files of this image (ELF objects, C/C++ headers, Python sources,
    # /usr/share; kanzi-cpp_amd/corpus.py:local) -- no corpus can be fetched, these files exist on every box
    9: dict(transform="BWT+MTFT+ZRLT", entropy="ANS0", block=8 << 20, corpus="local"),
}

# Kernel name (the KScope label of the launch) -> pipeline stage. First matching prefix wins.
STAGE_PREFIXES = [
    ("k_bwt_f", "bwt_forward"), ("bwt_f", "bwt_forward"),
    ("k_bwt_i", "bwt_inverse"), ("bwt_i", "bwt_inverse"), ("k_bwt_bases", "bwt_forward"),
    ("k_mtf_f", "mtft_forward"), ("k_mtf_i", "mtft_inverse"),
    ("k_zrlt_f", "zrlt_forward"), ("k_zrlt_i", "zrlt_inverse"), ("k_zero_dst", "zrlt_inverse"),
    ("k_srt_f", "srt_forward"), ("k_srt_zero", "srt_forward"), ("k_srt_i", "srt_inverse"),
    ("k_lz_inverse", "lz_inverse"), ("k_lz_i", "lz_inverse"), ("k_lz", "lz_forward"),
    ("k_ans0_stats", "ans0_encode"), ("k_ans0_encode", "ans0_encode"),
    ("k_ans0_scan", "ans0_decode"), ("k_ans0_decode", "ans0_decode"),
    ("k_ans1_hist", "ans1_encode"), ("k_ans1_ctx", "ans1_encode"), ("k_ans1_encode", "ans1_encode"),
    ("k_ans1_scan", "ans1_decode"), ("k_ans1_tables", "ans1_decode"), ("k_ans1_decode", "ans1_decode"),
    ("k_huff_encode", "huffman_encode"), ("k_huff_scan", "huffman_decode"), ("k_huff_decode", "huffman_decode"),
    ("k_fpaq_probs", "fpaq_encode"), ("k_fpaq_code", "fpaq_encode"), ("k_fpaq_e", "fpaq_encode"), ("k_fpaq_d", "fpaq_decode"),
    ("k_block_sum", "bit_assembly"), ("k_block_scan", "bit_assembly"), ("k_assemble", "bit_assembly"),
    ("memset_out", "bit_assembly"), ("k_put_prologue", "bit_assembly"), ("k_shift_bits", "bit_assembly"),
    ("k_walk_blocks", "framing_walk"), ("k_check_prelen", "framing_walk"),
]


def stage_of(name, direction):
    for pre, st in STAGE_PREFIXES:
        if name.startswith(pre):
            return st
    # helpers shared by several stages (tile scans, compaction, sequence bookkeeping) are charged to the direction
    return "other_" + direction


def stage_bytes(stage, N, Cc, nblocks):
    """Algorithmic HBM bytes of one pass of a stage (ideal one-pass traffic); N = uncompressed bytes
    of the batch, Cc = compressed bytes. Stages between BWT and the entropy coder see N-sized data up to ZRLT."""
    if stage in ("bwt_forward", "bwt_inverse"):
        return 2 * N + 32 * nblocks
    if stage in ("mtft_forward", "mtft_inverse", "srt_forward", "srt_inverse"):
        return 2 * N
    if stage in ("bit_assembly",):
        return 2 * Cc
    return N + Cc          # entropy coders, ZRLT, LZ: N + C per direction


def cpu_baseline(data, n_sample, cfg, cores):
    """Reference kanzi (oracle/_ref, the unmodified sources) on the host cores: compress + decompress a bounded
    sample through CompressedOutputStream/InputStream with -j min(cores, 64, #blocks). The clock runs INSIDE the C
    harness (oracle/ref_harness.cpp:ref_time_roundtrip) around the stream objects only; buffers are numpy arrays
    handed over by pointer, nothing is marshalled in the timed region."""
    import numpy as np
    import knzlib
    bs = cfg["block"]
    nblocks = max(1, (n_sample + bs - 1) // bs)
    src = np.frombuffer(data, dtype=np.uint8, count=n_sample)
    comp = np.empty(n_sample + n_sample // 2 + (1 << 20), dtype=np.uint8)
    back = np.empty(max(1, n_sample), dtype=np.uint8)
    u8p = C.POINTER(C.c_uint8)
    info = ""
    try:
        cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        cpu_model = "unknown"
    so = knzlib.ensure_ref()
    if so is not None:
        L = C.CDLL(so)
        L.ref_time_roundtrip.restype = C.c_int
        L.ref_time_roundtrip.argtypes = [u8p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_int, C.c_int, u8p, C.c_size_t,
                                         C.POINTER(C.c_size_t), u8p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        jobs = max(1, min(cores, 64, nblocks))
        clen, te, td = C.c_size_t(0), C.c_double(0), C.c_double(0)
        best = None
        for _ in range(3 if n_sample <= (512 << 20) else 1):          # best of 3, BASELINE.md's own method
            rc = L.ref_time_roundtrip(src.ctypes.data_as(u8p), n_sample, cfg["transform"].encode(), cfg["entropy"].encode(), bs, jobs,
                                      comp.ctypes.data_as(u8p), comp.size, C.byref(clen), back.ctypes.data_as(u8p), C.byref(te), C.byref(td))
            if rc != 0:
                raise RuntimeError("reference round trip failed: %d" % rc)
            if best is None or te.value + td.value < best[0] + best[1]:
                best = (te.value, td.value)
        info = ", best of 3 runs" if n_sample <= (512 << 20) else ""
        kind, t_enc, t_dec = "reference", best[0], best[1]
        enc = comp[:clen.value].tobytes()
    else:
        # Clean checkout without the reference build: the C restatement (oracle/), one independent stream per block on
        # every host core (ctypes releases the GIL), so that the figure is a multi-core one like the reference's.
        from concurrent.futures import ThreadPoolExecutor
        ora = knzlib.Oracle()
        jobs = max(1, min(cores, 64, nblocks))
        blocks = [bytes(src[i * bs:min(n_sample, (i + 1) * bs)]) for i in range(nblocks)]
        with ThreadPoolExecutor(jobs) as ex:
            t0 = time.perf_counter()
            encs = list(ex.map(lambda b: ora.compress(b, cfg["transform"], cfg["entropy"], bs)[1], blocks))
            t1 = time.perf_counter()
            decs = list(ex.map(lambda eb: ora.decompress(eb[0], len(eb[1]))[1], zip(encs, blocks)))
            t2 = time.perf_counter()
        assert decs == blocks
        kind, t_enc, t_dec, enc = "port", t1 - t0, t2 - t1, None
        info = "; oracle/_ref absent: C restatement, one stream per block on %d threads (marshalling included)" % jobs
    mbps = n_sample / (t_enc + t_dec) / 1e6
    return dict(value=round(mbps, 2), unit="MB/s", cores=jobs, kind=kind, cpu=cpu_model, host_cores=cores,
                sample="first %d bytes of the workload, %d blocks, -j %d, timed inside C: encode %.3f s + decode %.3f s%s" % (
                    n_sample, nblocks, jobs, t_enc, t_dec, info),
                enc_MBps=round(n_sample / t_enc / 1e6, 2), dec_MBps=round(n_sample / t_dec / 1e6, 2)), enc


def end_to_end(data, cfg):
    """What a C caller of the drop-in API gets (src/api/Compressor.hpp:92-116, Decompressor.hpp:63-117): host bytes ->
    initCompressor/compress per block -> .knz file on tmpfs -> initDecompressor/decompress per block -> host bytes in the
    caller's (already touched) buffer; PCIe, file I/O and every host copy included. Same entry points and structs as
    kanzi_amd/kanzi.py binds, called on numpy memory so that no Python object is built inside the timed region."""
    import numpy as np
    kz = importlib.import_module("kanzi_amd.kanzi")
    L = kz.lib()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    bs = cfg["block"]
    n = len(data)
    path = ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp") + "/knz_bench_%d.knz" % os.getpid()
    src = np.frombuffer(data, dtype=np.uint8)
    back = np.zeros(n + bs, dtype=np.uint8)
    sp, bp = src.ctypes.data, back.ctypes.data
    best = None
    try:
        for _ in range(3):
            back[:] = 0
            if os.path.exists(path):
                os.remove(path)                 # (truncating the previous repetition's file is tmpfs work, not the library's: 2-4 ms for 59 MB)
            t0 = time.perf_counter()
            f = libc.fopen(path.encode(), b"wb")
            cd = kz.cData(cfg["transform"].encode(), cfg["entropy"].encode(), bs, 8, 0, 0)
            ctx = C.c_void_p()
            if L.initCompressor(C.byref(cd), f, C.byref(ctx)) != 0:
                raise RuntimeError("initCompressor failed")
            out = C.c_size_t(0)
            written = 0
            for off in range(0, n, bs):
                if L.compress(ctx, sp + off, min(bs, n - off), C.byref(out)) != 0:
                    raise RuntimeError("compress failed")
                written += out.value
            if L.disposeCompressor(C.byref(ctx), C.byref(out)) != 0:
                raise RuntimeError("disposeCompressor failed")
            written += out.value
            libc.fclose(f)
            t1 = time.perf_counter()
            f = libc.fopen(path.encode(), b"rb")
            dd = kz.dData(bs, 8, 0, b"", b"", 0, 0, 0, 6)
            ctx = C.c_void_p()
            if L.initDecompressor(C.byref(dd), f, C.byref(ctx)) != 0:
                raise RuntimeError("initDecompressor failed")
            total = 0
            while True:
                ins, outs = C.c_size_t(0), C.c_size_t(bs)
                if L.decompress(ctx, bp + total, C.byref(ins), C.byref(outs)) != 0:
                    raise RuntimeError("decompress failed")
                if outs.value == 0:
                    break
                total += outs.value
            L.disposeDecompressor(C.byref(ctx))
            libc.fclose(f)
            t2 = time.perf_counter()
            if total != n or not np.array_equal(back[:n], src):      # checked outside the timed region
                raise RuntimeError("end-to-end round trip mismatch")
            cur = dict(value=round(n / (t2 - t0) / 1e6, 2), unit="MB/s", compress_MBps=round(n / (t1 - t0) / 1e6, 2),
                       decompress_MBps=round(n / (t2 - t1) / 1e6, 2), compressed_bytes=written, jobs=8,
                       host_layer="six lanes on the one GPU (three in the kernels at a time), 24 MiB batches, sink thread; GPU_MAX_HW_QUEUES=%s" % os.environ.get("GPU_MAX_HW_QUEUES", "unset"),
                       path="host bytes -> libkanzi_amd.so C API (initCompressor/compress, initDecompressor/decompress; one call per block) -> .knz on tmpfs -> host bytes")
            if best is None or cur["value"] > best["value"]:
                best = cur
    finally:
        if os.path.exists(path):
            os.remove(path)
    return best


def real_bytes_leg(ctx, torch, np, hipapi, corpus, framing, dev, steps=3, with_cpu=True):
    """Config 9 inside the default run: config 3's chain on REAL bytes (files of this image, kanzi-cpp_amd/corpus.py:local -- no corpus
    can be fetched and silesia.tar is on no box), `steps` timed encode+decode passes after one warm-up, device resident like `value`.
    The driver's line is measured on the synthetic stand-in, whose repeats are shorter than those of real files; this object is the same
    measurement on bytes nobody generated."""
    cfg = CONFIGS[9]
    data, desc = corpus.load(cfg["corpus"], None)
    n = len(data)
    bs = cfg["block"]
    import hashlib
    d_in = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    d_in[:n].copy_(torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()))
    p = ctx.params(cfg["transform"], cfg["entropy"], bs)
    cap = ctx.encode_bound(p, n)
    d_enc = torch.zeros(cap, dtype=torch.uint8, device=dev)
    d_dec = torch.empty(n + bs + 64, dtype=torch.uint8, device=dev)
    hdr, hb = framing.make_header(p.entropy_type, p.transform_type, bs, 0, 0)
    nb = (n + bs - 1) // bs
    st = {}

    def one():
        st["bits"] = ctx.encode_blocks(p, d_in.data_ptr(), n, d_enc.data_ptr(), cap, prologue=hdr, prologue_bits=hb)
        st["out"] = ctx.decode_blocks(p, d_enc.data_ptr(), st["bits"], hb, d_dec.data_ptr(), n + bs, max_blocks=max(nb, 1))[0]

    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    assert st["out"] == n and torch.equal(d_dec[:n], d_in[:n]), "round trip mismatch (real bytes)"
    # doubling rounds and stage time of the suffix sort, one profiled encode
    ctx.set_profiling(True)
    ctx.encode_blocks(p, d_in.data_ptr(), n, d_enc.data_ptr(), cap, prologue=hdr, prologue_bits=hb)
    kt = ctx.kernel_times()
    ctx.set_profiling(False)
    rounds = sum(l for nm, ms, l in kt if nm == "k_bwt_f_round")
    fwd_ms = sum(ms for nm, ms, l in kt if stage_of(nm, "encode") == "bwt_forward")
    out = {"value": round(n / el / 1e6, 2), "unit": "MB/s", "ms_per_step": round(el * 1e3, 4), "steps": steps, "warmup": 1,
           "workload": "-t %s -e %s -b %dm, %s" % (cfg["transform"], cfg["entropy"], bs >> 20, desc), "corpus_bytes": n, "blocks": nb,
           "input_md5": hashlib.md5(data).hexdigest(), "compressed_bytes": (st["bits"] + 7) // 8,
           "bwt_doubling_rounds": rounds, "bwt_forward_stage_ms": round(fwd_ms, 4), "bit_exact_vs_reference": None, "cpu_baseline": None}
    if with_cpu:
        cpu, ref_enc = cpu_baseline(data, n, cfg, os.cpu_count() or 1)
        out["cpu_baseline"] = cpu
        if ref_enc is not None:
            got = bytes(d_enc[:(st["bits"] + 7) // 8].cpu().numpy())
            out["bit_exact_vs_reference"] = (got == ref_enc)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="weak")
    ap.add_argument("--limit", type=int, default=0, help="use only the first LIMIT bytes of the corpus")
    ap.add_argument("--cpu-sample", type=int, default=256 << 20, help="bytes of the workload the CPU reference is timed on (the whole silesia corpus fits)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-real", action="store_true", help="skip the real_bytes object (config 9, 3 steps) of the default single-GPU run")
    ap.add_argument("--reps", type=int, default=0, help="repetitions of the separate encode / decode / per-kernel timing passes (default: 2..5 by --steps)")
    ap.add_argument("--dist-backend", default="nccl", help="gloo + --share-device: the N-rank code path on a 1-GPU box (developer check)")
    ap.add_argument("--share-device", action="store_true", help="every rank uses cuda:0")
    ap.add_argument("--many-blocks", type=int, default=-1, help="copies of the corpus in the many-block sharded job (default: 5 when N > 1, "
                    "off at N = 1; 0 = off): one job of that many corpora back to back, sharded over the ranks by block index")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)

    import __graft_entry__ as ge
    ge.load_package()
    hipapi = importlib.import_module("kanzi_amd.hipapi")
    corpus = importlib.import_module("kanzi_amd.corpus")
    framing = importlib.import_module("kanzi_amd.framing")
    sharded = importlib.import_module("kanzi_amd.sharded")

    cfg = CONFIGS[args.config]
    data, desc = corpus.load(cfg["corpus"], args.limit or None)
    n_total = len(data)
    import hashlib
    input_md5 = hashlib.md5(data).hexdigest()
    bs = cfg["block"]
    nblocks_total = (n_total + bs - 1) // bs
    # this rank's share of the corpus
    if world > 1 and args.scaling == "strong":
        first_block, cnt = sharded.block_ranges(n_total, bs, world)[rank]
    else:
        first_block, cnt = 0, nblocks_total
    lo = first_block * bs
    hi = min(n_total, (first_block + cnt) * bs)
    n = hi - lo
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.Stream(device=dev)  # the library launches (and HIP-event-times) on this stream
    ctx = hipapi.Context(local_rank, stream=stream.cuda_stream)

    h_in = torch.from_numpy(np.frombuffer(data, dtype=np.uint8, count=n, offset=lo).copy())
    d_in = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    d_in[:n].copy_(h_in)
    p = ctx.params(cfg["transform"], cfg["entropy"], bs)
    cap = ctx.encode_bound(p, n)
    d_enc = torch.zeros(cap, dtype=torch.uint8, device=dev)
    d_dec = torch.empty(n + bs + 64, dtype=torch.uint8, device=dev)
    # rank 0 carries the stream header, the last rank with blocks the end marker: the runs concatenate to one stream
    hdr, hdr_bits = framing.make_header(p.entropy_type, p.transform_type, bs, 0, n_total) if first_block == 0 else (b"", 0)
    finish = 1 if (first_block + cnt == nblocks_total) else 0

    state = {"bits": 0, "out_bytes": 0}

    def encode():
        state["bits"] = ctx.encode_blocks(p, d_in.data_ptr(), n, d_enc.data_ptr(), cap, prologue=hdr, prologue_bits=hdr_bits,
                                          first_block=first_block, finish=finish)

    def decode():
        ob, eb, nb = ctx.decode_blocks(p, d_enc.data_ptr(), state["bits"], hdr_bits, d_dec.data_ptr(), n + bs, max_blocks=max(cnt, 1))
        state["out_bytes"] = ob

    def step():
        if cnt:
            encode()
            decode()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- correctness of what was timed (outside the timed region)
    if cnt:
        assert state["out_bytes"] == n, "decoded %d of %d bytes" % (state["out_bytes"], n)
        assert torch.equal(d_dec[:n], d_in[:n]), "round trip mismatch"
    comp_bytes = (state["bits"] + 7) // 8
    if world > 1:
        t = torch.tensor([comp_bytes], dtype=torch.int64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        comp_total = int(t.item())
    else:
        comp_total = comp_bytes

    # ---- N > 1: the other way of giving N ranks work, timed the same way (every rank takes part: barrier and MAX reduce inside)
    other = None
    if world > 1:
        omode = "strong" if args.scaling == "weak" else "weak"
        fb2, c2 = sharded.block_ranges(n_total, bs, world)[rank] if omode == "strong" else (0, nblocks_total)
        lo2, hi2 = fb2 * bs, min(n_total, (fb2 + c2) * bs)
        n2 = max(hi2 - lo2, 0)
        d_in2 = torch.empty(n2 + 64, dtype=torch.uint8, device=dev)
        if n2:
            d_in2[:n2].copy_(torch.from_numpy(np.frombuffer(data, dtype=np.uint8, count=n2, offset=lo2).copy()))
        cap2 = ctx.encode_bound(p, n2)
        d_enc2 = torch.zeros(cap2, dtype=torch.uint8, device=dev)
        d_dec2 = torch.empty(n2 + bs + 64, dtype=torch.uint8, device=dev)
        hdr2, hdr_bits2 = framing.make_header(p.entropy_type, p.transform_type, bs, 0, n_total) if fb2 == 0 else (b"", 0)
        fin2 = 1 if (fb2 + c2 == nblocks_total) else 0
        st2 = {"bits": 0, "out": 0}

        def step2():
            if c2:
                st2["bits"] = ctx.encode_blocks(p, d_in2.data_ptr(), n2, d_enc2.data_ptr(), cap2, prologue=hdr2, prologue_bits=hdr_bits2,
                                                first_block=fb2, finish=fin2)
                st2["out"] = ctx.decode_blocks(p, d_enc2.data_ptr(), st2["bits"], hdr_bits2, d_dec2.data_ptr(), n2 + bs, max_blocks=max(c2, 1))[0]

        for _ in range(args.warmup):
            step2()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        a2 = time.perf_counter()
        for _ in range(args.steps):
            step2()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        e2 = time.perf_counter() - a2
        t = torch.tensor([e2], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e2 = float(t.item())
        if c2:
            assert st2["out"] == n2 and torch.equal(d_dec2[:n2], d_in2[:n2]), "round trip mismatch (second measurement)"
        most = max(c for _, c in sharded.block_ranges(n_total, bs, world))
        job2 = n_total if omode == "strong" else world * n_total
        other = {"scaling": omode, "value": round(job2 / (e2 / args.steps) / 1e6, 2), "unit": "MB/s", "ms_per_step": round(e2 / args.steps * 1e3, 4),
                 "steps": args.steps, "bytes_rank0": n2, "blocks_rank0": c2}
        if omode == "strong":
            other["block_count_ceiling"] = round(nblocks_total / most, 3) if most else None
            other["blocks_largest_share"] = most
            # speed-up over this run's own 26-block-per-rank step (the weak line's per-rank time is the N = 1 step time) and what part
            # of the block-count ceiling that is: the figure that answers "MB/s on one corpus at N GPUs"
            one = elapsed / args.steps
            other["speedup_vs_one_rank_step"] = round(one / (e2 / args.steps), 3)
            other["efficiency_vs_ceiling"] = round((one / (e2 / args.steps)) / (nblocks_total / most), 3) if most else None
        del d_in2, d_enc2, d_dec2

    # ---- a job with MANY blocks, sharded (strong scaling without the block-count ceiling in the way: 26 blocks over 8 ranks can
    # be at most 6.5x faster, 5 corpora back to back are 127 blocks of 8 MiB -> 7.94x). Rank r encodes and decodes blocks
    # [first, first + count) of the concatenation; the copies are identical bytes, which blocks that are independent cannot notice.
    many = None
    copies = args.many_blocks if args.many_blocks >= 0 else (5 if world > 1 else 0)
    if copies > 0:
        n_job = copies * n_total
        nb_job = (n_job + bs - 1) // bs
        fb3, c3 = sharded.block_ranges(n_job, bs, world)[rank]
        lo3, hi3 = fb3 * bs, min(n_job, (fb3 + c3) * bs)
        n3 = max(hi3 - lo3, 0)
        src_np = np.frombuffer(data, dtype=np.uint8)
        idx_lo = lo3 % n_total if n_total else 0
        reps_needed = (idx_lo + n3 + n_total - 1) // n_total if n3 else 0
        share = np.tile(src_np, reps_needed)[idx_lo:idx_lo + n3] if n3 else np.empty(0, dtype=np.uint8)
        d_in3 = torch.empty(n3 + 64, dtype=torch.uint8, device=dev)
        if n3:
            d_in3[:n3].copy_(torch.from_numpy(np.ascontiguousarray(share)))
        del share
        cap3 = ctx.encode_bound(p, n3)
        d_enc3 = torch.zeros(cap3, dtype=torch.uint8, device=dev)
        d_dec3 = torch.empty(n3 + bs + 64, dtype=torch.uint8, device=dev)
        hdr3, hdr_bits3 = framing.make_header(p.entropy_type, p.transform_type, bs, 0, n_job) if fb3 == 0 else (b"", 0)
        fin3 = 1 if (fb3 + c3 == nb_job) else 0
        st3 = {"bits": 0, "out": 0}

        def step3():
            if c3:
                st3["bits"] = ctx.encode_blocks(p, d_in3.data_ptr(), n3, d_enc3.data_ptr(), cap3, prologue=hdr3, prologue_bits=hdr_bits3,
                                                first_block=fb3, finish=fin3)
                st3["out"] = ctx.decode_blocks(p, d_enc3.data_ptr(), st3["bits"], hdr_bits3, d_dec3.data_ptr(), n3 + bs, max_blocks=max(c3, 1))[0]

        k3 = max(1, min(args.steps, 5))
        step3()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        a3 = time.perf_counter()
        for _ in range(k3):
            step3()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        e3 = time.perf_counter() - a3
        if world > 1:
            t = torch.tensor([e3], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e3 = float(t.item())
        if c3:
            assert st3["out"] == n3 and torch.equal(d_dec3[:n3], d_in3[:n3]), "round trip mismatch (many-block job)"
        most3 = max(c for _, c in sharded.block_ranges(n_job, bs, world))
        many = {"scaling": "strong", "job": "%d copies of the corpus back to back, %d blocks, sharded by block index" % (copies, nb_job),
                "job_bytes": n_job, "value": round(n_job / (e3 / k3) / 1e6, 2), "unit": "MB/s", "ms_per_step": round(e3 / k3 * 1e3, 4), "steps": k3,
                "blocks_rank0": c3, "blocks_largest_share": most3, "block_count_ceiling": round(nb_job / most3, 3) if most3 else None}
        del d_in3, d_enc3, d_dec3

    result = None
    if rank == 0:
        # ---- separate encode / decode timing + per-kernel HIP-event timing (same stream), rank 0's share
        def timed(fn, reps):
            torch.cuda.synchronize()
            a = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - a) / reps

        reps = args.reps if args.reps > 0 else max(2, min(args.steps, 5))
        t_enc = timed(encode, reps)
        t_dec = timed(decode, reps)
        kern, stages = {}, {}
        ctx.set_profiling(True)
        for direction, fn in (("encode", encode), ("decode", decode)):
            for _ in range(reps):
                fn()
            for nm, ms, launches in ctx.kernel_times():
                if not launches:
                    continue
                key = nm if nm not in kern else nm + "@" + direction
                kern[key] = dict(ms_per_step=ms / reps, launches_per_step=launches / reps, stage=stage_of(nm, direction))
                st = stages.setdefault(stage_of(nm, direction), dict(ms=0.0, launches=0.0, kernels=[]))
                st["ms"] += ms / reps
                st["launches"] += launches / reps
                st["kernels"].append(nm)
        ctx.set_profiling(False)
        nb_rank = max(cnt, 1)
        dom_stage = max(stages.items(), key=lambda kv: kv[1]["ms"])[0]
        dom = stages[dom_stage]
        alg = stage_bytes(dom_stage, n, comp_bytes, nb_rank)
        achieved = alg / (dom["ms"] * 1e-3) / 1e9
        dk_name, dk = max(((k, v) for k, v in kern.items() if v["stage"] == dom_stage), key=lambda kv: kv[1]["ms_per_step"])
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath)).get("config%d" % args.config, {})
                # measured with rocprofv3 --pmc on the same workload (tools/pmc_summary.py); only valid for the same input size
                if tj.get("n_bytes") == n and dom_stage in tj.get("stages", {}):
                    traffic = tj["stages"][dom_stage]
                    traffic_source = "profiles/pmc_traffic.json (%s)" % tj.get("source", "rocprofv3 --pmc")
            except Exception:
                traffic = None
        pipe_bytes = 2 * (n + comp_bytes)
        step_ms = (t_enc + t_dec) * 1e3
        roofline = dict(
            bound="hbm", stage=dom_stage, kernel=dk_name.split("@")[0], achieved=round(achieved, 2), peak=8000.0, unit="GB/s",
            frac=round(achieved / 8000.0, 5), traffic=traffic, traffic_source=traffic_source, algorithmic_bytes=alg,
            stage_ms=round(dom["ms"], 4), stage_launches_per_step=round(dom["launches"], 1),
            note="dominant STAGE: algorithmic bytes of one pass of the stage / summed HIP-event time of all its launches in one step; per-kernel "
                 "timing runs every launch of a step back to back on one stream, while the timed steps (value, enc_MBps, dec_MBps, pipeline) run the "
                 "BWT stages of a batch in 3 parts on 3 streams (KNZ_BWT_SPLIT), so the stage times add up to more than ms_per_step",
            dominant_kernel=dict(name=dk_name.split("@")[0], avg_ms=round(dk["ms_per_step"] / dk["launches_per_step"], 4),
                                 launches_per_step=round(dk["launches_per_step"], 1), ms_per_step=round(dk["ms_per_step"], 4),
                                 library=None),       # every launch of the pipeline is a kernel of this repository (no rocPRIM / hipCUB since round 3)
            pipeline=dict(algorithmic_bytes=pipe_bytes, ms=round(step_ms, 4), achieved=round(pipe_bytes / (step_ms * 1e-3) / 1e9, 2),
                          frac=round(pipe_bytes / (step_ms * 1e-3) / 1e9 / 8000.0, 5),
                          note="2*(N + C): uncompressed read + compressed write per direction, over encode + decode"),
            stages_ms={k: round(v["ms"], 4) for k, v in sorted(stages.items(), key=lambda kv: -kv[1]["ms"])},
            kernels_ms={k: round(v["ms_per_step"], 4) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms_per_step"])})

        cpu = None
        bit_exact = None
        e2e = None
        if world == 1:                           # the CPU baseline and the host-buffer path are single-GPU report lines
            if not args.no_cpu:
                # a corpus of at most 64 blocks is timed whole, one job per block (config 4: 30 blocks at -j 30, the figure the
                # reference would give a user of that file); larger ones on the first --cpu-sample bytes
                sample_n = n if nblocks_total <= 64 else min(n, args.cpu_sample)
                if sample_n < n:
                    sample_n -= sample_n % bs if sample_n >= bs else 0
                cpu, ref_enc = cpu_baseline(data, sample_n, cfg, os.cpu_count() or 1)
                if ref_enc is not None:
                    # bit-exactness of the device stream against the reference on the same sample
                    hdr2, hb2 = framing.make_header(p.entropy_type, p.transform_type, bs, 0, 0)
                    bits2 = ctx.encode_blocks(p, d_in.data_ptr(), sample_n, d_enc.data_ptr(), cap, prologue=hdr2, prologue_bits=hb2)
                    got = bytes(d_enc[:(bits2 + 7) // 8].cpu().numpy())
                    bit_exact = (got == ref_enc)
            if not args.no_e2e:
                try:
                    e2e = end_to_end(data, cfg)
                except Exception as ex:      # the host library is a separate .so; its absence must not hide the device line
                    e2e = dict(error=str(ex))
        if cfg["entropy"] == "FPAQ":
            # the floor the format sets: interval and probabilities carry through a whole block, so a block is ONE dependent chain
            roofline["format_floor"] = ("FPAQ codes a block as one dependent chain (entropy/FPAQEncoder.cpp:58-110, FPAQDecoder.cpp:62-120): one wave per block. "
                                        "Decoder: about 46 instructions per bit at 4.1 cycles per instruction of a lone wave = 51 ns per bit (a host core: about 8 ns), "
                                        "with nothing to take off the chain (the next context is the decoded bit). Encoder: the probabilities are computed by 32 family "
                                        "waves per block beside the coding wave (one launch, progress counters), the coding wave alone is 28 ns per bit. "
                                        "30 blocks are 30 chains against 30 host cores: this configuration stays below the reference on a many-core host.")
        real_bytes = None
        if world == 1 and args.config == 3 and not args.limit and not args.no_real:
            try:
                real_bytes = real_bytes_leg(ctx, torch, np, hipapi, corpus, framing, dev, steps=3, with_cpu=not args.no_cpu)
            except Exception as ex:          # (a box without the files of this image: the line must still come out)
                real_bytes = dict(error=str(ex))
        if "k_bwt_f_round" in kern:
            roofline["bwt_doubling_rounds"] = round(kern["k_bwt_f_round"]["launches_per_step"], 1)
        ms_per_step = elapsed / args.steps * 1e3
        job_bytes = n_total if (world == 1 or args.scaling == "strong") else world * n_total
        value = job_bytes / (elapsed / args.steps) / 1e6
        real = "stand-in" not in desc
        result = {
            "metric": "encode+decode MB/s on %s (bit-exact)" % ("silesia.tar" if real and cfg["corpus"] == "silesia" else desc.split(",")[0] if real else "the synthetic stand-in for %s" % ("silesia.tar" if cfg["corpus"] == "silesia" else cfg["corpus"])),
            "value": round(value, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak" if (world > 1 and args.scaling == "weak") else ("strong" if world > 1 else "weak"),
            "vs_baseline": None, "dtype": "u8", "data": ("real (files of this image, not silesia.tar)" if cfg["corpus"] == "local" else "real") if real else "synthetic",
            "config": {"workload": "-t %s -e %s -b %dm, %s" % (cfg["transform"], cfg["entropy"], bs >> 20, desc),
                       "corpus_bytes": n_total, "input_md5": input_md5, "blocks": nblocks_total, "compressed_bytes": comp_total,
                       "bytes_rank0": n, "blocks_rank0": cnt,
                       "parallelism": ("1 GPU" if world == 1 else
                                       "blocks of one corpus sharded over %d ranks in contiguous ranges, no collective" % world if args.scaling == "strong"
                                       else "a job of %d corpora sharded by block index: rank r takes the r-th corpus (%d blocks), no collective" % (world, nblocks_total))},
            "enc_MBps": round(n / t_enc / 1e6, 2), "dec_MBps": round(n / t_dec / 1e6, 2),
            "bit_exact_vs_reference": bit_exact,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "end_to_end": e2e,
            "real_bytes": real_bytes,
        }
        if world > 1 and args.scaling == "strong":
            # what block granularity allows: the largest share is ceil(blocks / N) blocks, so N ranks can be at most blocks / ceil(blocks / N)
            # times faster than one; the driver computes the efficiency from the per-N values, this is the bound to hold it against
            most = max(c for _, c in sharded.block_ranges(n_total, bs, world))
            result["config"]["block_count_ceiling"] = round(nblocks_total / most, 3) if most else None
            result["config"]["blocks_largest_share"] = most
        if other is not None:
            result["one_corpus_sharded" if other["scaling"] == "strong" else "n_corpora"] = other
        if many is not None:
            result["many_blocks_sharded"] = many
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
