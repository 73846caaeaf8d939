"""The drop-in boundary on the GPU: the reference-compatible C API (include/kanzi_api.h, through the
ctypes classes that mirror src/api/kanzi.py) and the C++ mirror classes (tests/cpp/host_mirror_test,
modelled on the reference's own test executables)."""
import importlib
import os
import subprocess

import pytest

import knzlib
import vectors

pytestmark = pytest.mark.gpu


def _kanzi():
    knzlib.load_pkg()
    kz = importlib.import_module("kanzi_amd.kanzi")
    # tests/test_host_stub.py re-runs this file on the CPU with the host library linked against a stand-in for the device
    # library (tests/stub/knz_hip_stub.c): the override lives here, in the tests -- the product binding has none
    alt = os.environ.get("KNZ_TEST_KANZI_LIB")
    if alt and kz._lib is None:
        kz.LIB_PATH = alt
    return kz


def test_cpp_host_mirror_suite():
    exe = os.environ.get("KNZ_TEST_HOST_MIRROR_EXE") or os.path.join(knzlib.ROOT, "tests", "cpp", "host_mirror_test")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    r = subprocess.run([exe, "all"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_c_api_roundtrip_and_bit_exact(tmp_path, oracle):
    # src/test/test_api.py / TestAPI.c: fill_buffer(size) = (i*17+3)&255, several blocks, parameter rewriting
    kz = _kanzi()
    assert kz.lib().getCompressorVersion() == 0x010000 and kz.lib().getDecompressorVersion() == 0x010000
    for transform, entropy, bs, jobs, n in [("none", "huffman", 1024, 1, 1024 * 3 + 100), ("BWT+MTFT+ZRLT", "ANS0", 65536, 2, 300000),
                                            ("bwt+srt+zrlt", "fpaq", 65530, 3, 200001), ("RLT", "NONE", 4096, 1, 0),
                                            ("srt", "ans1", 262144, 2, 600000)]:
        data = vectors.make(("fill17", n)) if n < 5000 else vectors.make(("mixed", n, 5))
        path = str(tmp_path / "x.knz")
        c = kz.Compressor(path, transform, entropy, bs, jobs)
        assert c.params.transform.decode() == transform.upper() and c.params.entropy.decode() == entropy.upper()
        block = c.params.blockSize
        assert block == (bs + 15) & -16
        total = 0
        for off in range(0, len(data), block):
            total += c.compress(data[off:off + block])
        with pytest.raises(kz.KanziError) as ei:
            c.compress(bytes(block + 1))                       # inSize > blockSize -> ERR_INVALID_PARAM
        assert ei.value.code == 18
        total = c.close()
        enc = open(path, "rb").read()
        assert total == len(enc)
        rc, ref = oracle.compress(data, transform.upper(), entropy.upper(), block, orig_size=0, jobs=jobs)
        assert rc == 0 and enc == ref, (transform, entropy)
        d = kz.Decompressor(path, buffer_size=block, jobs=jobs)
        out = bytearray()
        while True:
            chunk = d.decompress(block)
            out += chunk
            if len(chunk) < block:
                break
        d.close()
        assert bytes(out) == data


def test_streams_of_bitstream_versions_below_6(tmp_path, oracle):
    """the header of versions 3-5 (io/CompressedInputStream.cpp:541-558,606-645: one checksum bit, no padding, 16
    checksum bits) read by the host stream class, the blocks' old Huffman chunk, BWT header and LZ layouts read on the device. The
    streams come from the oracle's writers for the old layouts, which tests/test_old_bitstreams.py pins with the reference's decoder."""
    kz = _kanzi()
    for ver, transform, entropy, bs, ck, n in [(5, "BWT", "HUFFMAN", 65536, 0, 300000), (4, "NONE", "HUFFMAN", 4096, 32, 20000),
                                               (3, "BWT+MTFT+ZRLT", "ANS0", 1 << 20, 0, 1500000), (5, "BWT+SRT+ZRLT", "FPAQ", 16384, 32, 50000),
                                               (5, "LZ", "HUFFMAN", 65536, 0, 200000), (4, "LZX", "ANS1", 1 << 18, 32, 700000)]:
        data = vectors.make(("mixed", n, 7))
        oracle.set_bs_version(ver)
        try:
            rc, enc = oracle.compress(data, transform, entropy, bs, checksum=ck, orig_size=len(data))
        finally:
            oracle.set_bs_version(6)
        assert rc == 0 and enc[4] >> 4 == ver
        path = str(tmp_path / "old.knz")
        open(path, "wb").write(enc)
        d = kz.Decompressor(path, buffer_size=bs, jobs=1)
        out = bytearray()
        while True:
            chunk = d.decompress(bs)
            out += chunk
            if len(chunk) < bs:
                break
        d.close()
        assert bytes(out) == data, (ver, transform, entropy)
    # headerless: the caller names the version (dData.bsVersion, src/api/Decompressor.hpp:76 -> io/CompressedInputStream.cpp:97-98)
    data = vectors.make(("text", 90000, 6))
    oracle.set_bs_version(5)
    try:
        rc, enc = oracle.compress(data, "BWT", "HUFFMAN", 32768, headerless=1)
    finally:
        oracle.set_bs_version(6)
    assert rc == 0
    path = str(tmp_path / "old_headerless.knz")
    open(path, "wb").write(enc)
    d = kz.Decompressor(path, buffer_size=32768, headerless=True, transform="BWT", entropy="HUFFMAN", block_size=32768, bsVersion=5)
    out = bytearray()
    while True:
        chunk = d.decompress(32768)
        out += chunk
        if len(chunk) < 32768:
            break
    d.close()
    assert bytes(out) == data


def test_lanes_and_devices_give_the_single_device_stream(tmp_path, oracle, monkeypatch):
    """The stream classes spread consecutive batches over lanes -- one (device, context, worker thread) per entry of KNZ_DEVICES,
    two lanes on the default device when it is not set -- and append the independent bit runs in order (a run that does not start
    on a byte boundary is moved on its device). Whatever the lanes and the batch size, the file must be the reference's and must
    decode through the same lanes. KNZ_TEST_DEVICES (tests/test_host_stub.py: the stand-in device library accepts any device
    number) adds a list with two different devices."""
    kz = _kanzi()
    data = vectors.make(("mixed", 23 * 16384 + 777, 21))
    lists = ["0", "0,0", "0,0,0"] + ([os.environ["KNZ_TEST_DEVICES"]] if os.environ.get("KNZ_TEST_DEVICES") else [])
    for transform, entropy, bs, jobs, ck in [("BWT+MTFT+ZRLT", "ANS0", 16384, 3, 0), ("NONE", "HUFFMAN", 16384, 1, 32), ("RLT", "FPAQ", 32768, 2, 64)]:
        rc, ref = oracle.compress(data, transform, entropy, bs, orig_size=0, jobs=jobs, checksum=ck)
        assert rc == 0
        for devs in lists:
            for batch in ("1", "2", "5"):
                monkeypatch.setenv("KNZ_DEVICES", devs)
                monkeypatch.setenv("KNZ_BATCH_BLOCKS", batch)
                path = str(tmp_path / "lanes.knz")
                c = kz.Compressor(path, transform, entropy, bs, jobs, checksum=ck)
                for off in range(0, len(data), bs):
                    c.compress(data[off:off + bs])
                c.close()
                assert open(path, "rb").read() == ref, (transform, entropy, devs, batch)
                d = kz.Decompressor(path, buffer_size=bs, jobs=jobs)
                out = bytearray()
                while True:
                    chunk = d.decompress(bs)
                    out += chunk
                    if len(chunk) < bs:
                        break
                d.close()
                assert bytes(out) == data, (transform, entropy, devs, batch)
    # an empty stream and a stream of one short block go through the same machinery
    monkeypatch.setenv("KNZ_DEVICES", "0,0")
    for n in (0, 5, 1000):
        path = str(tmp_path / "tiny.knz")
        c = kz.Compressor(path, "BWT", "ANS0", 4096, 1)
        if n:
            c.compress(data[:n])
        c.close()
        assert open(path, "rb").read() == oracle.compress(data[:n], "BWT", "ANS0", 4096, orig_size=0)[1]


def test_reference_python_api_cases(tmp_path, oracle):
    """The cases of the reference's own ctypes test (src/test/test_api.py), written against the same class interface
    (bytes codec names, context managers, decompress_block, the reference's keyword names), plus what that test does
    not check: the file on disk is byte-identical to the reference's. Its headerless case declares bsVersion=1 for a stream
    written as version 6 (harmless there: LZ skips a 25-byte block and ANS0 chunks did not change with the version); it runs as
    written, and once more with bsVersion=0, which the reference also reads as an old layout (every version below 6)."""
    kz = _kanzi()
    fill = lambda size: bytes((i * 17 + 3) & 0xFF for i in range(size))
    lzx = dict(transform=b"LZX", entropy=b"HUFFMAN", block_size=1024, jobs=1, checksum=0, headerless=0)
    with pytest.raises(Exception):
        kz.Compressor(str(tmp_path / "bad.knz"), transform=None)                   # test_init_invalid
    with kz.Compressor(str(tmp_path / "a.knz"), **lzx) as c:                      # test_init_dispose
        assert c is not None
    with kz.Compressor(str(tmp_path / "b.knz"), **lzx) as c:                      # test_compress_small
        assert c.compress(fill(256)) >= 0
    with kz.Compressor(str(tmp_path / "c.knz"), **lzx) as c:                      # test_compress_too_big
        with pytest.raises(kz.KanziError):
            c.compress(fill(4096))
    with kz.Compressor(str(tmp_path / "d.knz"), **lzx) as c:                      # test_compress_two_blocks
        assert c.compress(fill(300)) >= 0 and c.compress(fill(500)) >= 0
    # test_basic_decompression
    msg = b"Hello Kanzi! Hello Compression!"
    path = str(tmp_path / "e.knz")
    with kz.Compressor(path, transform=b"LZ", entropy=b"ANS0", block_size=1 << 16, jobs=1, checksum=32, headerless=0) as c:
        c.compress(msg)
    assert open(path, "rb").read() == oracle.compress(msg, "LZ", "ANS0", 1 << 16, checksum=32)[1]
    with kz.Decompressor(path, buffer_size=1 << 16, jobs=1, headerless=0) as d:
        assert d.decompress_block(1024) == msg
    # test_large_multi_block
    size = 2 * 1024 * 1024
    data = bytes((i * 7) & 0xFF for i in range(size))
    path = str(tmp_path / "f.knz")
    with kz.Compressor(path, transform=b"LZ", entropy=b"FPAQ", block_size=256 * 1024, jobs=1, checksum=64, headerless=0) as c:
        for off in range(0, size, 256 * 1024):
            c.compress(data[off:off + 256 * 1024])
    assert open(path, "rb").read() == oracle.compress(data, "LZ", "FPAQ", 256 * 1024, checksum=64)[1]
    out = bytearray()
    with kz.Decompressor(path, buffer_size=256 * 1024, jobs=1, headerless=0) as d:
        while True:
            try:
                block = d.decompress_block(256 * 1024)
                if not block:
                    break
                out.extend(block)
            except kz.KanziError:
                break
    assert bytes(out) == data
    # test_headerless
    msg = b"HEADERLESS MODE IS ACTIVE"
    path = str(tmp_path / "g.knz")
    with kz.Compressor(path, transform=b"LZ", entropy=b"ANS0", block_size=1 << 15, jobs=1, checksum=0, headerless=1) as c:
        c.compress(msg)
    for ver in (1, 0, 6):
        with kz.Decompressor(path, buffer_size=1 << 15, jobs=1, headerless=1, transform=b"LZ", entropy=b"ANS0", blockSize=1 << 15,
                             originalSize=len(msg), checksum=0, bsVersion=ver) as d:
            assert d.decompress_block(256) == msg, ver


def test_c_api_parameter_validation(tmp_path):
    kz = _kanzi()
    with pytest.raises(kz.KanziError) as ei:
        kz.Compressor(str(tmp_path / "y.knz"), "BOGUS", "ANS0", 4096)
    assert ei.value.code == 4                                   # ERR_CREATE_COMPRESSOR (unknown names throw inside init)
    with pytest.raises(kz.KanziError):
        kz.Compressor(str(tmp_path / "y.knz"), "NONE", "ANS0", 100)     # block size below 1024


def test_c_api_headerless(tmp_path):
    kz = _kanzi()
    data = vectors.make(("text", 70000, 2))
    path = str(tmp_path / "h.knz")
    c = kz.Compressor(path, "ZRLT", "HUFFMAN", 16384, 1, headerless=True)
    for off in range(0, len(data), 16384):
        c.compress(data[off:off + 16384])
    c.close()
    d = kz.Decompressor(path, buffer_size=16384, headerless=True, transform="ZRLT", entropy="HUFFMAN", block_size=16384)
    out = bytearray()
    while True:
        chunk = d.decompress(16384)
        out += chunk
        if len(chunk) < 16384:
            break
    d.close()
    assert bytes(out) == data


def test_threads_share_the_device(tmp_path):
    # src/api: "separate contexts are independent" -- four threads with their own compressor/decompressor objects
    # share the device context of the process (tools/gpu_threads.py)
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_threads.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_multi_batch_stream_beyond_compaction_threshold(tmp_path, monkeypatch):
    # Regression (round-1 advisor finding): the input stream drops the consumed prefix of its fetch buffer; that must
    # never happen while the host walk holds bit cursors into it. 40 incompressible 1 MiB blocks = 40 MiB of
    # compressed data, decoded in batches of 3 blocks, so many batches start beyond the compaction threshold.
    import numpy as np
    monkeypatch.setenv("KNZ_BATCH_BLOCKS", "3")
    kz = _kanzi()
    bs = 1 << 20
    data = np.random.default_rng(5).integers(0, 256, 40 * bs + 12345, dtype=np.uint8).tobytes()
    path = str(tmp_path / "big.knz")
    with kz.Compressor(path, "NONE", "NONE", bs, 1) as c:
        for off in range(0, len(data), bs):
            c.compress(data[off:off + bs])
    assert os.path.getsize(path) > 40 * bs
    d = kz.Decompressor(path, buffer_size=bs, jobs=1)
    out = bytearray()
    while True:
        chunk = d.decompress(bs)
        out += chunk
        if len(chunk) < bs:
            break
    d.close()
    assert bytes(out) == data


def test_cli_interoperates_with_the_reference_cli_on_device(tmp_path):
    # kanzi_amd_cli (reference option names) on the real device path: byte-identical files, cross decoding, --from/--to
    import test_host_stub
    cli = os.path.join(knzlib.ROOT, "kanzi-cpp_amd", "kanzi_amd_cli")
    assert os.path.exists(cli), "run __graft_entry__.build()"
    if os.environ.get("KNZ_TEST_KANZI_LIB"):
        pytest.skip("covered by tests/test_host_stub.py in the stub run")
    if knzlib.ensure_ref() is None or not os.path.exists(knzlib.REF_BIN):
        pytest.skip("reference build not available")
    test_host_stub.run_cli_interop(cli, knzlib.REF_BIN, tmp_path)


def test_cli_levels_5_and_6_write_and_read_the_reference_files(tmp_path):
    """kanzi_amd_cli -c -l 5 / -l 6 (TEXT + UTF on the host, BWT + RANK / SRT + ZRLT and the entropy coder on the device) against what the
    reference's `kanzi -c -l N -j 1` writes (tests/golden/levels.json, from oracle/_ref): the same bytes, and back to the input. Text, the
    mixed stand-in (blocks TEXT refuses), UTF-8 (blocks UTF takes), CRLF and XML text, copied spans, tiny inputs, incompressible data,
    with and without block checksums."""
    import hashlib
    import json
    cli = os.environ.get("KNZ_TEST_CLI", os.path.join(knzlib.PKG, "kanzi_amd_cli"))
    recs = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "levels.json")))
    for r in recs:
        if os.environ.get("KNZ_TEST_KANZI_LIB") and r["input"][1] > (6 << 20):
            continue                  # (the CPU run against the stand-in device keeps to the smaller inputs)
        d = vectors.make(tuple(r["input"]))
        assert hashlib.md5(d).hexdigest() == r["input_md5"]
        src, out, back = str(tmp_path / "in.bin"), str(tmp_path / "out.knz"), str(tmp_path / "back.bin")
        open(src, "wb").write(d)
        p = subprocess.run([cli, "-c", "-i", src, "-o", out, "-f", "-l", str(r["level"])] + r["extra"], capture_output=True, text=True)
        assert p.returncode == 0, (r, p.stderr)
        enc = open(out, "rb").read()
        assert len(enc) == r["out"]["len"] and hashlib.md5(enc).hexdigest() == r["out"]["md5"], (r["level"], r["input"], len(enc))
        p = subprocess.run([cli, "-d", "-i", out, "-o", back, "-f"], capture_output=True, text=True)
        assert p.returncode == 0 and open(back, "rb").read() == d, (r, p.stderr)


def test_bench_line_contract(tmp_path):
    """bench.py's one JSON line on a small slice of the headline workload: the keys the driver reads, the roofline and cpu_baseline
    objects, the bit-exact flag; and the N-rank path (two ranks on this one GPU over gloo): weak scaling in `value`, the one sharded
    corpus beside it with its block-count ceiling."""
    import json
    import sys
    root = knzlib.ROOT
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "1", "--limit", str(16 << 20), "--no-e2e",
                        "--cpu-sample", str(16 << 20)], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["unit"] == "MB/s" and d["dtype"] == "u8" and d["value"] > 0
    assert d["bit_exact_vs_reference"] is True
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1 and rf["achieved"] > 0 and "traffic" in rf
    assert d["cpu_baseline"]["kind"] == "reference" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert "workload" in d["config"] and "model" not in d["config"]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
                        os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--dist-backend", "gloo", "--share-device",
                        "--limit", str(32 << 20)], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads([x for x in r.stdout.strip().splitlines() if x.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["blocks_rank0"] == 4
    o = d["one_corpus_sharded"]
    assert o["scaling"] == "strong" and o["blocks_rank0"] == 2 and o["block_count_ceiling"] == 2.0 and o["value"] > 0
    assert o["efficiency_vs_ceiling"] > 0 and o["speedup_vs_one_rank_step"] > 0
    m = d["many_blocks_sharded"]                 # five corpora back to back = 20 blocks, 10 per rank
    assert m["scaling"] == "strong" and m["blocks_rank0"] == 10 and m["block_count_ceiling"] == 2.0 and m["value"] > 0


def _device_count():
    knzlib.load_pkg()
    import ctypes as C
    import importlib
    n = C.c_int(0)
    importlib.import_module("kanzi_amd.hipapi").lib().knz_hip_device_count(C.byref(n))
    return n.value


def test_two_or_more_physical_devices(tmp_path, oracle, monkeypatch):
    """The multi-GPU code as it runs on a node with several GPUs, skipped with the reason on a one-GPU box:
    (1) the in-library lanes with DIFFERENT physical devices behind them -- KNZ_DEVICES=0,1 (and 0..7 when there are eight) through
    initCompressor / compress and initDecompressor / decompress must write and read the single-device file; (2) bench.py --gpus 2
    under its default backend (nccl = RCCL) with one rank per GPU: the line the driver's scaling run collects."""
    n = _device_count()
    if os.environ.get("KNZ_TEST_DEVICES"):
        pytest.skip("stand-in device library: covered by test_lanes_and_devices_give_the_single_device_stream")
    if n < 2:
        pytest.skip("one GPU on this box (knz_hip_device_count = %d): KNZ_DEVICES=0,1 and the nccl branch of bench.py need two; "
                    "the same code runs with several contexts on cuda:0 and over gloo in the tests above" % n)
    kz = _kanzi()
    data = vectors.make(("mixed", 37 * 65536 + 4321, 31))
    lists = ["0,1", "1,0,1"] + ([",".join(str(i) for i in range(n))] if n > 2 else [])
    for transform, entropy, bs, jobs, ck in [("BWT+MTFT+ZRLT", "ANS0", 65536, 3, 32), ("BWT+SRT+ZRLT", "FPAQ", 131072, 1, 0)]:
        rc, ref = oracle.compress(data, transform, entropy, bs, orig_size=0, jobs=jobs, checksum=ck)
        assert rc == 0
        for devs in lists:
            for batch in ("1", "3"):
                monkeypatch.setenv("KNZ_DEVICES", devs)
                monkeypatch.setenv("KNZ_BATCH_BLOCKS", batch)
                path = str(tmp_path / "multi.knz")
                c = kz.Compressor(path, transform, entropy, bs, jobs, checksum=ck)
                for off in range(0, len(data), bs):
                    c.compress(data[off:off + bs])
                c.close()
                assert open(path, "rb").read() == ref, (transform, entropy, devs, batch)
                d = kz.Decompressor(path, buffer_size=bs, jobs=jobs)
                out = bytearray()
                while True:
                    chunk = d.decompress(bs)
                    out += chunk
                    if len(chunk) < bs:
                        break
                d.close()
                assert bytes(out) == data, (transform, entropy, devs, batch)
    monkeypatch.delenv("KNZ_DEVICES")
    monkeypatch.delenv("KNZ_BATCH_BLOCKS")
    import json
    import sys
    root = knzlib.ROOT
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29633",
                        os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--limit", str(64 << 20)],
                       capture_output=True, text=True, timeout=1200, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads([x for x in r.stdout.strip().splitlines() if x.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    o = d["one_corpus_sharded"]
    assert o["scaling"] == "strong" and o["block_count_ceiling"] == 2.0 and 0 < o["efficiency_vs_ceiling"] <= 1.2
    assert d["many_blocks_sharded"]["value"] > 0


def test_large_transfers_through_the_host_layer(tmp_path, oracle, monkeypatch):
    """Pieces of several MiB take the host layer's parallel paths (round 5): staging copies and the C API's file writes / reads are cut
    into slices for the helper threads (pwrite / pread on the FILE's descriptor), the runs are appended by a sink thread, the source is
    read 16 MiB ahead. Whatever the knobs, the file is the reference's and comes back as the input; a FILE opened for appending and a
    file written in the middle of another (an offset that is not 0) take the same code."""
    kz = _kanzi()
    data = vectors.make(("mixed", 11 * (4 << 20) + 12345, 17))
    bs = 4 << 20
    rc, ref = oracle.compress(data, "NONE", "HUFFMAN", bs, orig_size=0, jobs=1)
    assert rc == 0
    for copy_threads, sink, ahead, lanes, gate in [("4", "1", None, "6", "3"), ("1", "0", None, "1", "1"), ("3", "1", str(3 << 20), "12", "2"),
                                                   ("8", "1", str(1 << 20), "3", "5")]:
        # (the helper pool and the device gate are created once per process with the first values; the other values still run the
        # slicing arithmetic, and the lane count is read per stream: 1, 3, 6 and 12 lanes behind the gate)
        monkeypatch.setenv("KNZ_COPY_THREADS", copy_threads)
        monkeypatch.setenv("KNZ_SINK_THREAD", sink)
        monkeypatch.setenv("KNZ_LANES", lanes)
        monkeypatch.setenv("KNZ_DEVICE_CONCURRENCY", gate)
        if ahead:
            monkeypatch.setenv("KNZ_READ_AHEAD", ahead)
        path = str(tmp_path / "big.knz")
        c = kz.Compressor(path, "NONE", "HUFFMAN", bs, 1)
        for off in range(0, len(data), bs):
            c.compress(data[off:off + bs])
        total = c.close()
        enc = open(path, "rb").read()
        assert total == len(enc) and enc == ref, (copy_threads, sink)
        d = kz.Decompressor(path, buffer_size=bs, jobs=1)
        out = bytearray()
        while True:
            chunk = d.decompress(bs)
            out += chunk
            if len(chunk) < bs:
                break
        d.close()
        assert bytes(out) == data, (copy_threads, sink)
    # the C API on a FILE that does not start at offset 0 and on one opened for appending (no positional writes there)
    import ctypes as C
    L = kz.lib()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    libc.fwrite.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p]
    for mode, lead in ((b"wb", b"0123456789"), (b"ab", b"")):
        path = str(tmp_path / "off.knz")
        if mode == b"ab":
            open(path, "wb").write(b"leading bytes")
            lead = b"leading bytes"
        f = libc.fopen(path.encode(), mode)
        if mode == b"wb":
            libc.fwrite(lead, 1, len(lead), f)
        cd = kz.cData(b"NONE", b"HUFFMAN", bs, 1, 0, 0)
        ctx = C.c_void_p()
        assert L.initCompressor(C.byref(cd), f, C.byref(ctx)) == 0
        out = C.c_size_t(0)
        for off in range(0, len(data), bs):
            blk = data[off:off + bs]
            assert L.compress(ctx, blk, len(blk), C.byref(out)) == 0
        assert L.disposeCompressor(C.byref(ctx), C.byref(out)) == 0
        libc.fclose(f)
        got = open(path, "rb").read()
        # (src/api/Compressor.cpp:218-224: the size field of the header is the size the DESTINATION had when the compressor was created)
        want = ref if mode == b"wb" else oracle.compress(data, "NONE", "HUFFMAN", bs, orig_size=len(lead), jobs=1)[1]
        assert got[:len(lead)] == lead and got[len(lead):] == want, mode
