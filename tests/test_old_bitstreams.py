"""Bitstream versions below 6: the reference still DECODES them -- stream header
(io/CompressedInputStream.cpp:541-558,606-645), Huffman chunks (entropy/HuffmanDecoder.cpp:349-459), BWT block header
(transform/BWTBlockCodec.cpp:140-164), LZ / LZX blocks (transform/LZCodec.cpp:614-760) -- but writes version 6 only, so there is nothing to take fixtures from. The oracle therefore
carries writers for the old layouts (oracle/huffman.c, transforms.c, stream.c under knzo_set_bs_version), and what pins them is the
unmodified reference decoding what they write. Everything else (the device kernels for the old layouts in tests/test_emu_kernels.py
and tests/test_gpu_parity.py, the host header parser in tests/test_gpu_host_api.py) is checked against streams made this way."""
import numpy as np
import pytest

import knzlib
import vectors

CHAINS = [("NONE", "HUFFMAN"), ("BWT", "HUFFMAN"), ("BWT+MTFT+ZRLT", "ANS0"), ("BWT", "NONE"), ("BWT+SRT+ZRLT", "FPAQ"), ("RLT", "HUFFMAN"),
          ("BWT+RANK+ZRLT", "ANS1"), ("LZ", "HUFFMAN"), ("LZX", "ANS0")]


def old_stream(oracle, ver, data, transform, entropy, bs, checksum=0, orig_size=None):
    oracle.set_bs_version(ver)
    try:
        rc, enc = oracle.compress(data, transform, entropy, bs, checksum=checksum, orig_size=len(data) if orig_size is None else orig_size)
    finally:
        oracle.set_bs_version(6)
    assert rc == 0
    return enc


def test_reference_decodes_the_oracles_old_streams():
    if knzlib.ensure_ref() is None:
        pytest.skip("reference build not available")
    oracle, ref = knzlib.Oracle(), knzlib.Ref()
    rng = np.random.default_rng(1)
    datas = [vectors.make(("text", 70000, 1)), vectors.make(("mixed", 300000, 2))[200000:290000], rng.integers(0, 256, 20000, dtype=np.uint8).tobytes(),
             b"a" * 40000, b"xyz" * 10, vectors.make(("text", 5000, 2)), b""]
    n = 0
    for ver in (3, 4, 5):
        for t, e in CHAINS:
            for d in datas:
                for bs, ck, osz in ((4096, 0, None), (65536, 32, None), (1 << 20, 0, 0)):
                    enc = old_stream(oracle, ver, d, t, e, bs, ck, osz)
                    assert enc[4] >> 4 == ver
                    rr, back_r = ref.decompress(enc, len(d) + 64)
                    ro, back_o = oracle.decompress(enc, len(d) + 64)
                    assert rr == 0 and back_r == d, (ver, t, e, len(d), bs, ck)
                    assert ro == 0 and back_o == d, (ver, t, e, len(d), bs, ck)
                    n += 1
    assert n == 3 * len(CHAINS) * len(datas) * 3


def test_old_headers_differ_where_the_reference_says():
    """one checksum bit instead of two, no padding, 16 checksum bits seeded with the bare version: 136 bits without a size field"""
    oracle = knzlib.Oracle()
    d = vectors.make(("text", 3000, 4))
    new = oracle.compress(d, "NONE", "NONE", 4096, orig_size=0)[1]
    old = old_stream(oracle, 5, d, "NONE", "NONE", 4096, orig_size=0)
    assert len(new) - len(old) == 3 and new[20:] == old[17:]          # 160 against 136 header bits, the same blocks behind them
    bad = bytearray(old); bad[16] ^= 1                                 # header checksum
    assert oracle.decompress(bytes(bad), 4096)[0] == 19                # ERR_CRC_CHECK
    v7 = bytearray(old); v7[4] = (7 << 4) | (v7[4] & 15)
    assert oracle.decompress(bytes(v7), 4096)[0] == 16                 # ERR_STREAM_VERSION


def test_python_header_parser_reads_old_headers():
    import importlib
    knzlib.load_pkg()
    fr = importlib.import_module("kanzi_amd.framing")
    oracle = knzlib.Oracle()
    d = vectors.make(("text", 9000, 4))
    for ver, ck, osz in ((3, 0, 0), (4, 32, None), (5, 0, None), (5, 32, 1 << 40)):
        enc = old_stream(oracle, ver, d, "BWT+MTFT+ZRLT", "HUFFMAN", 4096, ck, osz)
        h = fr.parse_header(enc)
        assert h["bs_version"] == ver and h["checksum_bits"] == ck and h["block_size"] == 4096 and h["etype"] == 1
        assert h["orig_size"] == (len(d) if osz is None else osz)
        assert h["bits"] == 32 + 4 + 1 + 5 + 48 + 28 + 2 + 16 * (0 if h["orig_size"] == 0 else (h["orig_size"].bit_length() - 1) // 16 + 1) + 16
        bad = bytearray(enc); bad[10] ^= 0x40
        with pytest.raises(fr.HeaderError) as ei:
            fr.parse_header(bytes(bad))
        assert ei.value.code in (16, 19, 2)
    new = oracle.compress(d, "NONE", "ANS0", 4096, orig_size=len(d))[1]
    assert fr.parse_header(new)["bs_version"] == 6
