"""Parity tests proper: the HIP path (through the C ABI, libknz_hip.so) against the oracle, the
golden vectors generated from the reference, and size-independent properties at full block sizes.
Bit-exact everywhere (integer/byte work)."""
import hashlib
import importlib
import os

import numpy as np
import pytest

import knzlib
import vectors

pytestmark = pytest.mark.gpu

ENTROPY_ON_DEVICE = ["NONE", "ANS0", "ANS1", "HUFFMAN", "FPAQ"]
TRANSFORMS_ON_DEVICE = ["ZRLT", "MTFT", "BWT", "SRT", "RLT", "LZ", "LZX", "RANK"]


def matches(packed, b):
    if packed["len"] != len(b):
        return False
    if "hex" in packed:
        return packed["hex"] == b.hex()
    return packed["md5"] == hashlib.md5(b).hexdigest()


def stream_supported(transform, entropy):
    return entropy in ENTROPY_ON_DEVICE and all(t in TRANSFORMS_ON_DEVICE or t == "NONE" for t in transform.split("+"))


def gpu_compress(hip, data, transform, entropy, bs, checksum=0, orig_size=0, headerless=0):
    knzlib.load_pkg()
    fr = importlib.import_module("kanzi_amd.framing")
    p = hip.params(transform, entropy, bs, checksum)
    hdr, hb = (b"", 0) if headerless else fr.make_header(p.entropy_type, p.transform_type, bs, checksum, orig_size)
    cap = hip.encode_bound(p, len(data)) + 64
    d_in, d_out = hip.malloc(len(data) + 64), hip.malloc(cap)
    hip.h2d(d_in, data)
    bits = hip.encode_blocks(p, d_in, len(data), d_out, cap, prologue=hdr, prologue_bits=hb)
    out = hip.d2h(d_out, (bits + 7) // 8)
    hip.free(d_in); hip.free(d_out)
    return out, bits, hb


def gpu_decompress(hip, enc, transform, entropy, bs, n, start_bit, checksum=0):
    p = hip.params(transform, entropy, bs, checksum)
    d_in, d_out = hip.malloc(len(enc) + 64), hip.malloc(n + bs + 64)
    hip.h2d(d_in, enc)
    ob, eb, nb = hip.decode_blocks(p, d_in, 8 * len(enc), start_bit, d_out, n + bs)
    out = hip.d2h(d_out, ob)
    hip.free(d_in); hip.free(d_out)
    return out


def test_entropy_stage_golden(hip, golden):
    n = 0
    for rec in golden["stages"]:
        if rec["kind"] != "entropy" or rec["name"] not in ENTROPY_ON_DEVICE:
            continue
        d = vectors.make(tuple(rec["input"]))
        enc, bits = hip.entropy_encode(rec["name"], d)
        assert bits == rec["bits"], (rec["name"], rec["input"])
        assert matches(rec["out"], enc), (rec["name"], rec["input"])
        dec, out, used = hip.entropy_decode(rec["name"], enc, len(d))
        assert dec == len(d) and out == d and used == bits, (rec["name"], rec["input"])
        n += 1
    assert n >= 2 * len(vectors.STAGE_INPUTS)


def test_reference_quirks(hip):
    """tests/golden/quirks.json: streams the reference emits but cannot decode. Bit-exact means the same stream out
    and the same refusal on the way back (a decoder that 'repairs' them would accept streams the reference rejects)."""
    import json
    recs = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quirks.json")))
    for rec in recs:
        d = bytes.fromhex(rec["input_hex"])
        enc, bits = hip.entropy_encode(rec["entropy"], d)
        assert bits == rec["bits"] and hashlib.md5(enc).hexdigest() == rec["enc_md5"], rec["name"]
        dec, out, used = hip.entropy_decode(rec["entropy"], enc, len(d))
        got = len(d) if (dec == len(d) and out == d) else -1
        assert got == rec["ref_decoded"], rec["name"]


def test_entropy_stage_vs_oracle_random(hip, oracle):
    rng = np.random.default_rng(77)
    for i in range(12):
        n = int(rng.integers(33, 200000))
        if i % 3 == 0:
            d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        elif i % 3 == 1:
            d = bytes((rng.geometric(0.02 + 0.08 * i, n) % 256).astype(np.uint8))
        else:
            d = vectors.make(("mixedslice", 2 << 20, 5 + i, 1000 * i, 1000 * i + n))
        for e in ENTROPY_ON_DEVICE:
            ref, rbits = oracle.entropy_encode(e, d)
            got, gbits = hip.entropy_encode(e, d)
            assert gbits == rbits and got == ref, (e, i, n)
            dec, out, used = hip.entropy_decode(e, ref, n)
            assert dec == n and out == d and used == rbits, (e, i, n)


def test_entropy_decode_at_bit_offset(hip, oracle):
    # the block framing hands the decoder arbitrary bit offsets
    d = vectors.make(("text", 50000, 9))
    for e in ENTROPY_ON_DEVICE:
        enc, bits = oracle.entropy_encode(e, d)
        for shift in (1, 5, 13):
            v = int.from_bytes(enc, "big") << (8 - (shift % 8)) % 8 if shift % 8 else int.from_bytes(enc, "big")
            shifted = (int.from_bytes(enc, "big") << ((-shift) % 8)).to_bytes(len(enc) + 1, "big")
            buf = bytes(shift // 8) + shifted
            start = 8 * (shift // 8) + (8 - ((-shift) % 8)) % 8 if False else None
            # simpler: build the bit string explicitly
            bitstr = "1" * shift + bin(int.from_bytes(enc, "big"))[2:].zfill(8 * len(enc))
            bitstr += "0" * ((-len(bitstr)) % 8)
            buf = int(bitstr, 2).to_bytes(len(bitstr) // 8, "big")
            dec, out, used = hip.entropy_decode(e, buf, len(d), start_bit=shift)
            assert dec == len(d) and out == d and used == bits, (e, shift)


def test_truncated_payload_rejected(hip, oracle):
    # src/test/TestEntropyCodec.cpp:230-274
    d = vectors.make(("formula13", 4096))
    for e in [x for x in ENTROPY_ON_DEVICE if x != "NONE"]:
        enc, bits = oracle.entropy_encode(e, d)
        dec, out, used = hip.entropy_decode(e, enc[:len(enc) // 2], len(d))
        assert dec != len(d) or out != d


def test_transform_stage_golden(hip, golden):
    n = 0
    for rec in golden["stages"]:
        if rec["kind"] != "transform" or rec["name"] not in TRANSFORMS_ON_DEVICE:
            continue
        d = vectors.make(tuple(rec["input"]))
        if len(d) == 0:
            continue
        ok, out = hip.transform_forward(rec["name"], d, rec["cap"], rec["entropy"] or None)
        assert int(bool(ok)) == rec["ok"], (rec["name"], rec["input"])
        if rec["ok"]:
            assert matches(rec["out"], out), (rec["name"], rec["input"])
            k, back = hip.transform_inverse(rec["name"], out, max(len(d), len(out)) + 64)
            assert k == 1 and back == d, (rec["name"], rec["input"])
        n += 1
    assert n > 100


def test_srt_inverse_of_damaged_blocks(hip, oracle):
    """SRT.cpp:111-204 checks nothing about the body, so a damaged one decodes to definite bytes: first bucket bytes that make the
    initial list no permutation, ranks beyond the live part of the list (stale entries come back), ranks for zeros and zeros for ranks.
    The fast chain (one list entry per lane, queue offsets riding with the symbols) has to notice and hand the block to the general
    one; accept / refuse and every byte as the oracle."""
    rng = np.random.default_rng(77)
    blocks = [vectors.make(("text", 60000, 3)), rng.integers(0, 4, 20000, dtype=np.uint8).tobytes(), b"xy" * 500,
              vectors.make(("mixed", 300000, 2))[250000:290000], rng.integers(0, 256, 9000, dtype=np.uint8).tobytes()]
    n = 0
    for d in blocks:
        ok, good = oracle.forward("SRT", d, len(d) + 2048, "FPAQ")
        assert ok
        at = 0
        for _ in range(256):                                      # the header: 256 var-ints
            while good[at] & 0x80:
                at += 1
            at += 1
        for variant in range(12):
            m = bytearray(good)
            body = len(m) - at
            if variant == 0:
                m[at] = (m[at] + 1) & 0xFF
            elif variant == 1:
                m[at] = 200
            elif variant < 5:
                for i in rng.integers(1, body, 4):
                    if m[at + int(i)]:
                        m[at + int(i)] = int(rng.integers(variant * 60, 256))
            elif variant < 9:
                for i in rng.integers(1, body, 1 << (variant - 4)):
                    m[at + int(i)] = int(rng.integers(0, 7))
            else:
                for i in rng.integers(1, body, 30):
                    m[at + int(i)] = int(rng.integers(0, 256))
            k1, b1 = oracle.inverse("SRT", bytes(m), len(d))
            k2, b2 = hip.transform_inverse("SRT", bytes(m), len(d))
            assert bool(k1) == bool(k2), (len(d), variant)
            if k1:
                assert b1 == b2, (len(d), variant)
            n += 1
    assert n == 60


def test_transform_capacity_semantics(hip, oracle):
    # destination capacity changes results (ZRLT/RLT): src/test/TestTransforms.cpp:371-498,500-757
    rng = np.random.default_rng(5)
    cases = [vectors.make(("mixed", 200000, 7)), rng.integers(0, 256, 30000, dtype=np.uint8).tobytes(),
             vectors.make(("ffmix", 500)), bytes(5000), vectors.make(("runs", 300, 40))]
    for d in cases:
        for t in ["ZRLT", "RLT", "MTFT", "SRT", "BWT", "LZ", "LZX"]:
            lzmax = ((len(d) + 16) if len(d) <= 1024 else len(d) + len(d) // 64) + 2      # LZCodec.hpp:91-95
            for cap in (len(d) // 2, len(d) - 1, len(d), len(d) + 1, len(d) + 33, len(d) + 1024, len(d) + 2048, lzmax - 1, lzmax):
                ok1, o1 = oracle.forward(t, d, cap, "ANS0")
                ok2, o2 = hip.transform_forward(t, d, cap, "ANS0")
                assert bool(ok1) == bool(ok2), (t, cap, len(d))
                if ok1:
                    assert o1 == o2, (t, cap)
                    for icap in (len(d) - 1, len(d), len(d) + 1, max(len(d), len(o1)) + 64):
                        k1, b1 = oracle.inverse(t, o1, icap)
                        k2, b2 = hip.transform_inverse(t, o1, icap)
                        assert bool(k1) == bool(k2), (t, cap, icap)
                        if k1:
                            assert b1 == b2
    # the reference's own ZRLT KATs
    assert hip.transform_inverse("ZRLT", bytes([2]), 1) == (1, bytes([1]))
    assert hip.transform_inverse("ZRLT", bytes([2, 2]), 1)[0] == 0
    assert hip.transform_inverse("ZRLT", bytes([0xFF]), 1)[0] == 0
    assert hip.transform_forward("ZRLT", bytes([0xFE]), 1)[0] == 0
    assert hip.transform_forward("ZRLT", bytes([0]), 1) == (1, bytes([0]))


def test_sbrt_rank_and_timestamp(hip, oracle):
    # SBRT modes 2 and 3 (transform/SBRT.cpp): typical input is BWT output; also raw data, tiny and 64-byte-edge sizes
    rng = np.random.default_rng(4)
    cases = [vectors.make(("text", 70000, 2)), vectors.make(("mixed", 200001, 9)), rng.integers(0, 256, 5000, dtype=np.uint8).tobytes(),
             bytes(1000), b"a", bytes(range(256)) * 3, vectors.make(("text", 64, 1)), vectors.make(("text", 65, 1)), vectors.make(("text", 63, 1))]
    for d in cases:
        x = oracle.forward("BWT", d, len(d) + 64)[1]
        for data in (d, x):
            if not data:
                continue
            for name in ("RANK", "TIMESTAMP"):
                ok1, o1 = oracle.forward(name, data, len(data))
                ok2, o2 = hip.transform_forward(name, data, len(data))
                assert ok1 == 1 and ok2 == 1 and o1 == o2, (name, len(data))
                k, back = hip.transform_inverse(name, o1, len(data))
                assert k == 1 and back == data, (name, len(data))
                assert hip.transform_forward(name, data, len(data) - 1)[0] == 0          # count > capacity -> false


def test_bwt_known_strings(hip):
    # src/test/TestBWT.cpp:42-60 ; header: mode byte + (primaryIndex-1)
    for src, exp, pidx in [(b"mississippi", b"ipssmpissii", 5),
                           (b"SIX.MIXED.PIXIES.SIFT.SIXTY.PIXIE.DUST.BOXES", b"STEXYDST.E.IXXIIXXSSMPPS.B..EE..USFXDIIOIIIT", 31)]:
        ok, out = hip.transform_forward("BWT", src, len(src) + 64)
        assert ok == 1 and out[0] == 0 and out[1] == pidx - 1 and out[2:] == exp
    # invalid primary index must fail (src/test/TestBWT.cpp:274-347)
    bad = bytes([0, 200]) + b"ipssmpissii"
    assert hip.transform_inverse("BWT", bad, 64)[0] == 0


def test_bwt_round0_key_shapes(hip, oracle):
    """launch_bwt_forward sorts the first round on as many symbols as fit a 64-bit key beside the position (5 for blocks up to
    16 MiB, 4 above); short suffixes take their place through the order in which the first pass is fed. Every key length the knob
    can force (knz_hip_tune "bwt_nsym") must give the reference's BWT on lone blocks, also around the ends of blocks that finish in
    zero bytes (where the padded keys of short suffixes collide with real ones)."""
    from kanzi_amd import hipapi
    rng = np.random.default_rng(11)
    cases = [bytes(5000), vectors.make(("text", 8000, 3)), vectors.make(("mixed", 70000, 4)), rng.integers(0, 256, 4097, dtype=np.uint8).tobytes(),
             vectors.make(("runs", 300, 40)), vectors.make(("text", 300000, 5)), bytes(2500) + b"\x01" + bytes(2499),
             b"ab\0\0\0\0ab\0\0\0\0\0\0", b"\0\0\0", b"xyzxyz\0xyzxy", bytes(3000) + b"ab" + bytes(4), bytes(7) + b"\x01" + bytes(3)]
    try:
        for d in cases:
            ok1, o1 = oracle.forward("BWT", d, len(d) + 64, "ANS0")
            assert ok1
            for nsym in (0, 1, 2, 3, 4):
                assert hipapi.lib().knz_hip_tune(b"bwt_nsym", nsym) == 0
                ok2, o2 = hip.transform_forward("BWT", d, len(d) + 64, "ANS0")
                assert ok2 and o1 == o2, (len(d), nsym)
    finally:
        hipapi.lib().knz_hip_tune(b"bwt_nsym", 0)


def test_mtft_tile_sizes(hip, oracle):
    """MTFT cuts a block into tiles of 4 KiB, or 1 KiB when the batch is small (shorter dependent chains); both sizes, forced through
    knz_hip_tune("mtf_tile"), must give the reference's bytes in both directions."""
    from kanzi_amd import hipapi
    d = vectors.make(("mixed", 300001, 6))
    ok1, o1 = oracle.forward("MTFT", d, len(d) + 64, "ANS0")
    assert ok1
    try:
        for tile in (1024, 4096, 0):
            assert hipapi.lib().knz_hip_tune(b"mtf_tile", tile) == 0
            ok2, o2 = hip.transform_forward("MTFT", d, len(d) + 64, "ANS0")
            assert ok2 and o1 == o2, tile
            ok3, back = hip.transform_inverse("MTFT", o1, len(d))
            assert ok3 and back == d, tile
    finally:
        hipapi.lib().knz_hip_tune(b"mtf_tile", 0)


def test_jobs_capacity_model(hip, oracle):
    # the reference's output depends on -j through buffer-slot capacities
    d = vectors.make(("mixed", 700001, 11))
    outs = {}
    for jobs in (1, 2, 3):
        rc, ref = oracle.compress(d, "BWT+SRT+ZRLT", "ANS0", 262144, headerless=1, jobs=jobs)
        p = hip.params("BWT+SRT+ZRLT", "ANS0", 262144, jobs=jobs)
        cap = hip.encode_bound(p, len(d)) + 64
        d_in, d_out = hip.malloc(len(d) + 64), hip.malloc(cap)
        hip.h2d(d_in, d)
        bits = hip.encode_blocks(p, d_in, len(d), d_out, cap)
        got = hip.d2h(d_out, (bits + 7) // 8)
        hip.free(d_in); hip.free(d_out)
        assert got == ref, jobs
        outs[jobs] = got
    assert outs[1] == outs[2] and outs[1] != outs[3]


def test_stream_golden(hip, golden):
    n = 0
    for rec in golden["streams"]:
        if not stream_supported(rec["transform"], rec["entropy"]):
            continue
        d = vectors.make(tuple(rec["input"]))
        out, bits, hb = gpu_compress(hip, d, rec["transform"], rec["entropy"], rec["block"], rec["checksum"],
                                     rec["orig_size"], rec["headerless"])
        assert matches(rec["out"], out), rec
        back = gpu_decompress(hip, out, rec["transform"], rec["entropy"], rec["block"], len(d), hb, rec["checksum"])
        assert back == d, rec
        n += 1
    assert n >= 15


def test_blocks_of_bitstream_versions_below_6(hip, oracle):
    """SURVEY.md 8(f)4 on the device: knz_params.bs_version 3..5 selects the old Huffman chunk layout (HuffmanDecoder.cpp:349-459: one
    code stream per chunk; k_huff_scan<true> / k_huff_decode<true>) and the old BWT block header (BWTBlockCodec.cpp:140-164: a mode byte
    per chunk; k_bwt_i_header<true>). The blocks come from the oracle's writers for those layouts, which tests/test_old_bitstreams.py
    pins with the reference's decoder. LZ / LZX blocks in their old token layout (LZCodec.cpp:614-760): k_lz_inverse<true>."""
    hipapi = importlib.import_module("kanzi_amd.hipapi")
    rng = np.random.default_rng(21)
    datas = [vectors.make(("text", 70000, 1)), vectors.make(("mixed", 300000, 2))[180000:290000], rng.integers(0, 256, 20000, dtype=np.uint8).tobytes(),
             b"a" * 40000, b"xyz" * 10, rng.integers(0, 3, 16385, dtype=np.uint8).tobytes()]
    n = 0
    for ver in (3, 5):
        for t, e in (("NONE", "HUFFMAN"), ("BWT", "HUFFMAN"), ("BWT+MTFT+ZRLT", "ANS0"), ("BWT", "NONE"), ("BWT+SRT+ZRLT", "HUFFMAN"), ("LZ", "HUFFMAN"),
                     ("LZX", "NONE")):
            for d in datas:
                for bs, ck in ((4096, 0), (65536, 32), (1 << 20, 0)):
                    oracle.set_bs_version(ver)
                    try:
                        rc, enc = oracle.compress(d, t, e, bs, checksum=ck, headerless=1)
                    finally:
                        oracle.set_bs_version(6)
                    assert rc == 0
                    p = hip.params(t, e, bs, ck, bs_version=ver)
                    d_in, d_out = hip.malloc(len(enc) + 64), hip.malloc(len(d) + bs + 64)
                    hip.h2d(d_in, enc)
                    ob, eb, nb = hip.decode_blocks(p, d_in, 8 * len(enc), 0, d_out, len(d) + bs)
                    out = hip.d2h(d_out, ob)
                    hip.free(d_in); hip.free(d_out)
                    assert out == d, (ver, t, e, len(d), bs, ck)
                    n += 1
    assert n == 2 * 7 * len(datas) * 3
    p = hip.params("NONE", "NONE", 65536, 0, bs_version=7)
    d_in, d_out = hip.malloc(4096), hip.malloc(70000)
    with pytest.raises(hipapi.KnzError) as ei:
        hip.decode_blocks(p, d_in, 8 * 1024, 0, d_out, 65536)
    assert ei.value.code == 16
    hip.free(d_in); hip.free(d_out)


def test_stream_vs_oracle_ragged(hip, oracle):
    for spec, bs in [(("mixed", 3 * 262144 + 777, 4), 262144), (("text", 100000, 3), 1024), (("rand", 40000, 1), 16384),
                     (("ramp", 15), 1024), (("ramp", 16), 1024), (("ramp", 33), 1024), (("const", 50000, 7), 4096)]:
        d = vectors.make(spec)
        for e in ENTROPY_ON_DEVICE:
            for t in ("NONE", "BWT+MTFT+ZRLT", "BWT+SRT+ZRLT"):
                rc, ref = oracle.compress(d, t, e, bs, headerless=1)
                out, bits, hb = gpu_compress(hip, d, t, e, bs, headerless=1)
                assert out == ref, (spec, t, e)
                assert gpu_decompress(hip, ref, t, e, bs, len(d), 0) == d, (spec, t, e)


def test_chains_of_more_than_four_stages(hip, oracle):
    """five, six and seven device stages (chains in which no even-numbered stage expands: those run into the reference's undefined behaviour): the block header carries the skip flags as a byte of its own (io/CompressedOutputStream.cpp:791-799);
    the oracle's streams for these chains are pinned by the reference in tests/test_oracle_vs_ref.py"""
    for spec, bs in [(("mixed", 700001, 11), 65536), (("text", 300000, 3), 262144), (("rand", 40000, 1), 16384)]:
        d = vectors.make(spec)
        for t, e, ck in [("BWT+RANK+ZRLT+RLT+MTFT", "ANS0", 0), ("RLT+BWT+RANK+ZRLT+MTFT+SRT", "HUFFMAN", 32), ("RLT+BWT+SRT+ZRLT+RLT+MTFT+ZRLT", "FPAQ", 64)]:
            rc, ref = oracle.compress(d, t, e, bs, ck, headerless=1)
            assert rc == 0
            out, bits, hb = gpu_compress(hip, d, t, e, bs, checksum=ck, headerless=1)
            assert out == ref, (spec, t, e)
            assert gpu_decompress(hip, ref, t, e, bs, len(d), 0, checksum=ck) == d, (spec, t, e)


def test_lz_block_groups_and_long_matches(hip, oracle):
    # the LZ encoder sorts (block, hash) keys in groups of at most 8190 (LZX) blocks: more blocks than one group
    d = vectors.make(("mixed", 9000 * 1024 + 321, 6))
    rc, ref = oracle.compress(d, "LZX", "NONE", 1024, headerless=1)
    out, bits, hb = gpu_compress(hip, d, "LZX", "NONE", 1024, headerless=1)
    assert rc == 0 and out == ref
    assert gpu_decompress(hip, ref, "LZX", "NONE", 1024, len(d), 0) == d
    # matches longer than the candidate lengths measured up front (248) and than MAX_MATCH (65793), overlapping copies
    rng = np.random.default_rng(8)
    unit = rng.integers(0, 256, 3000, dtype=np.uint8).tobytes()
    d = bytes(200000) + unit * 70 + b"ab" * 40000 + unit[:777] * 90 + rng.integers(0, 256, 70000, dtype=np.uint8).tobytes() + unit * 5
    for t in ("LZ", "LZX"):
        cap = len(d) + len(d) // 64 + 64
        ok1, o1 = oracle.forward(t, d, cap)
        ok2, o2 = hip.transform_forward(t, d, cap)
        assert ok1 == 1 and ok2 == 1 and o1 == o2, t
        k, back = hip.transform_inverse(t, o1, len(d) + 64)
        assert k == 1 and back == d, t


def _full_case(hip, config):
    """BASELINE config `config` at its own block size on 64 MiB: the device stream must be the reference's .knz
    (md5 + length from oracle/_ref, tests/golden/golden_full.json), and must decode back to the input."""
    import json
    recs = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_full.json")))
    rec = [r for r in recs if r["config"] == config][0]
    d = vectors.make(tuple(rec["input"]))
    assert hashlib.md5(d).hexdigest() == rec["input_md5"]
    out, bits, hb = gpu_compress(hip, d, rec["transform"], rec["entropy"], rec["block"], orig_size=rec["orig_size"])
    assert len(out) == rec["out"]["len"], (config, len(out), rec["out"]["len"])
    assert hashlib.md5(out).hexdigest() == rec["out"]["md5"], config
    assert gpu_decompress(hip, out, rec["transform"], rec["entropy"], rec["block"], len(d), hb) == d, config


def test_full_size_reference_stream_config1_huffman_4m(hip):
    _full_case(hip, 1)


def test_full_size_reference_stream_config2_ans0_4m(hip):
    _full_case(hip, 2)


def test_full_size_reference_stream_config3_bwt_mtft_zrlt_ans0_8m(hip):
    _full_case(hip, 3)


def test_full_size_reference_stream_config4_bwt_srt_zrlt_fpaq_32m(hip):
    _full_case(hip, 4)


def test_full_size_reference_stream_config5_lzx_ans1_16m(hip):
    _full_case(hip, 5)


def test_full_size_reference_stream_config4_four_blocks_of_32m(hip):
    _full_case(hip, "config4:4blocks")


def test_full_size_reference_stream_config4_six_blocks_of_copied_spans(hip):
    """block ids 4 and 5 of a 32 MiB stream (the slot model i % jobs / first_block_id beyond four blocks) on text with 30 % copied
    spans: long common prefixes through SRT and FPAQ"""
    _full_case(hip, "config4:6blocks_repeats")


def test_decode_in_ranges_side_by_side(hip, oracle):
    """knz_hip_decode_blocks runs a batch of 8 or more blocks with inverse stages as two or three block ranges on streams of their own (knob
    dec_parts, csrc/api.hip decode_impl; at least two blocks per range): every range is a smaller decode with its own workspaces. Whatever the number of ranges, the output is
    the input -- chains with BWT, SRT, RLT, checksums of both widths, a short last block, and a damaged block in the last range reported."""
    hipapi = importlib.import_module("kanzi_amd.hipapi")
    L = hipapi.lib()
    d = vectors.make(("mixed", 17 * 65536 + 12345, 41))
    try:
        for t, e, bs, ck in [("BWT+MTFT+ZRLT", "ANS0", 65536, 0), ("BWT+SRT+ZRLT", "FPAQ", 65536, 32), ("RLT", "HUFFMAN", 65536, 64), ("BWT", "ANS1", 131072, 0)]:
            rc, enc = oracle.compress(d, t, e, bs, checksum=ck, headerless=1)
            assert rc == 0
            for parts in (1, 2, 3):
                assert L.knz_hip_tune(b"dec_parts", parts) == 0
                assert gpu_decompress(hip, enc, t, e, bs, len(d), 0, checksum=ck) == d, (t, e, parts)
        # a flipped bit near the end of the stream: the range that holds the block reports it
        t, e, bs, ck = "BWT+MTFT+ZRLT", "ANS0", 65536, 32
        rc, enc = oracle.compress(d, t, e, bs, checksum=ck, headerless=1)
        bad = bytearray(enc); bad[len(bad) - 3000] ^= 0x10
        for parts in (1, 3):
            L.knz_hip_tune(b"dec_parts", parts)
            with pytest.raises(hipapi.KnzError):
                gpu_decompress(hip, bytes(bad), t, e, bs, len(d), 0, checksum=ck)
    finally:
        L.knz_hip_tune(b"dec_parts", 3)


def test_suffix_sort_label_paths_give_one_stream(hip):
    """Round 6: the suffix sort keeps its labels as versioned 64-bit entries and refines small groups in one kernel (k_bwt_f_small_fused); blocks
    above 256 MiB use 32-bit labels with separate key kernels (KNZ_BWT_PLAIN_LABELS), and KNZ_BWT_NO_FUSE keeps versioned labels with the
    two kernels; the fused kernel ranks the members of a group on packed unique keys where a block has at most 8 MiB and on plain keys
    otherwise (bwt_no_pack forces the plain form). All paths must write the same bytes -- the reference's (tests/golden/golden_full.json,
    hard:repeats, hard:dna, hard:fibword, all in blocks of 8 MiB: packed keys by default; the 32 MiB and 256 MiB blocks of the config-4 and
    big-block tests take the plain form by themselves)."""
    import json
    hipapi = importlib.import_module("kanzi_amd.hipapi")
    recs = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_full.json")))
    L = hipapi.lib()
    try:
        for config in ("hard:repeats", "hard:dna", "hard:fibword"):
            rec = [r for r in recs if r["config"] == config][0]
            d = vectors.make(tuple(rec["input"]))
            for knob in (None, "bwt_no_fuse", "bwt_plain_labels", "bwt_no_pack"):
                if knob:
                    assert L.knz_hip_tune(knob.encode(), 1) == 0
                try:
                    out, bits, hb = gpu_compress(hip, d, rec["transform"], rec["entropy"], rec["block"], orig_size=rec["orig_size"])
                finally:
                    if knob:
                        L.knz_hip_tune(knob.encode(), 0)
                assert len(out) == rec["out"]["len"] and hashlib.md5(out).hexdigest() == rec["out"]["md5"], (config, knob)
    finally:
        L.knz_hip_tune(b"bwt_no_fuse", 0)
        L.knz_hip_tune(b"bwt_plain_labels", 0)
        L.knz_hip_tune(b"bwt_no_pack", 0)


_BIG_INPUT = {}


def _big_case(hip, config):
    """ONE block of 256 MiB / 1 GiB (vectors.BIG_CASES): the device stream is the reference's .knz (md5 + length from oracle/_ref,
    tests/golden/golden_full.json) and decodes back to the input. The input of the three 1 GiB cases is generated once."""
    import json
    recs = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_full.json")))
    rec = [r for r in recs if r["config"] == config][0]
    spec = tuple(rec["input"])
    if spec not in _BIG_INPUT:
        _BIG_INPUT.clear()
        _BIG_INPUT[spec] = vectors.make(spec)
        assert hashlib.md5(_BIG_INPUT[spec]).hexdigest() == rec["input_md5"]
    d = _BIG_INPUT[spec]
    out, bits, hb = gpu_compress(hip, d, rec["transform"], rec["entropy"], rec["block"], orig_size=rec["orig_size"])
    assert len(out) == rec["out"]["len"], (config, len(out), rec["out"]["len"])
    assert hashlib.md5(out).hexdigest() == rec["out"]["md5"], config
    if "ref_decode_error" in rec:
        # a stream the reference writes and cannot read (vectors.BIG_CASES): the device refuses it with the reference's error code
        hipapi = importlib.import_module("kanzi_amd.hipapi")
        with pytest.raises(hipapi.KnzError) as ei:
            gpu_decompress(hip, out, rec["transform"], rec["entropy"], rec["block"], len(d), hb)
        assert ei.value.code == rec["ref_decode_error"], (config, ei.value.code)
        return
    back = gpu_decompress(hip, out, rec["transform"], rec["entropy"], rec["block"], len(d), hb)
    assert len(back) == len(d) and back == d, config


def test_big_block_256m_headline_chain(hip):
    """io/CompressedOutputStream.cpp:69-82 accepts blocks up to 1 GiB; one block of 256 MiB through BWT+MTFT+ZRLT / ANS0: 28-bit positions,
    32 sub-lists per inverse tile row, ANS0 on 16,384 chunks of one block"""
    _big_case(hip, "big:bwt_chain_256m")


def test_big_block_1g_ans0(hip):
    _big_case(hip, "big:ans0_1g")


def test_big_block_1g_huffman(hip):
    _big_case(hip, "big:huffman_1g")


def test_big_block_1g_suffix_sort(hip):
    """transform/BWT.cpp:32 (MAX_BLOCK_SIZE = 1 GiB): the suffix sorter and its inverse on one block of 2^30 bytes -- 30-bit positions
    beside four symbols in the round-0 keys, the count + scatter passes (the one-sweep look-back words hold 30-bit counts), labels and
    slots up to 2^30, run-round keys of 63 bits"""
    _big_case(hip, "big:bwt_1g")


def test_big_block_1g_less_64k_suffix_sort_round_trip(hip):
    """the largest BWT block the reference reads back (2^30 - 65,536 bytes: the stored length 2^30 - 65,503 stays below the decoder's limit):
    suffix sort and inverse on the device, stream and round trip against the reference"""
    _big_case(hip, "big:bwt_1g_less_64k")
    _BIG_INPUT.clear()


def test_real_files_of_this_image_against_the_reference(hip, oracle):
    """REAL bytes: 64 MiB of corpus.local() -- ELF shared objects with their zero padding and string tables, then
    C/C++ headers with the same licence text in front of thousands of them -- through the headline chain at 8 MiB blocks. The
    expected stream is computed on this box by the unmodified reference (oracle/_ref, which travels with the repository); a
    checkout without it falls back to the C restatement. No fixture: the files belong to the image, not to the repository."""
    corpus = importlib.import_module("kanzi_amd.corpus")
    data, files, desc = corpus.local(64 << 20)
    assert len(data) == 64 << 20 and files > 20, desc
    # 56 MiB of ELF, then headers: blocks 0-6 are machine code and tables, block 7 is source text
    t, e, bs = "BWT+MTFT+ZRLT", "ANS0", 8 << 20
    if knzlib.ensure_ref() is not None:
        rc, want = knzlib.Ref().compress(data, t, e, bs, jobs=1, orig_size=len(data))
    else:
        rc, want = oracle.compress(data, t, e, bs, orig_size=len(data))
    assert rc == 0
    out, bits, hb = gpu_compress(hip, data, t, e, bs, orig_size=len(data))
    assert len(out) == len(want) and out == want
    assert gpu_decompress(hip, out, t, e, bs, len(data), hb) == data
    # and the part of the concatenation where the file kinds change (headers, Python, /usr/share), as config 4's chain sees it
    data2, _, _ = corpus.local(130 << 20)
    part = data2[100 << 20:116 << 20]
    t2, e2 = "BWT+SRT+ZRLT", "ANS0"
    if knzlib.ensure_ref() is not None:
        rc, want = knzlib.Ref().compress(part, t2, e2, bs, jobs=1, orig_size=len(part))
    else:
        rc, want = oracle.compress(part, t2, e2, bs, orig_size=len(part))
    assert rc == 0
    out, bits, hb = gpu_compress(hip, part, t2, e2, bs, orig_size=len(part))
    assert out == want
    assert gpu_decompress(hip, out, t2, e2, bs, len(part), hb) == part


@pytest.mark.parametrize("name", [c[0] for c in vectors.HARD_CASES if c[0].startswith("hard:")])
def test_long_common_prefix_inputs_at_full_block_size(hip, name):
    """Inputs a prefix-doubling sorter finds hard (copies with edits, X || X, periods 3 / 5 / 7 / 768, the Fibonacci word, DNA with
    repeats, one constant block) at 8 MiB / 32 MiB blocks: the device stream must be the reference's .knz (digests from oracle/_ref)."""
    _full_case(hip, name)


def test_full_size_determinism(hip):
    # two encodes of the same batch give the same bytes (no dependence on what earlier calls left in the workspaces)
    d = vectors.make(("mixed", 32 << 20, 2))
    for t, e, bs in [("NONE", "ANS0", 4 << 20), ("BWT+MTFT+ZRLT", "ANS0", 8 << 20), ("NONE", "ANS1", 16 << 20)]:
        out1, bits1, _ = gpu_compress(hip, d, t, e, bs, headerless=1)
        out2, bits2, _ = gpu_compress(hip, d, t, e, bs, headerless=1)
        assert out1 == out2, (t, e)


def test_sharded_runs_concatenate_to_single_stream(hip, oracle):
    # multi-GPU path on one device: two block ranges encoded as separate runs (first_block_id honoured),
    # concatenated on the host, must equal the single-stream bytes (kanzi-cpp_amd/sharded.py)
    sh = importlib.import_module("kanzi_amd.sharded")
    d = vectors.make(("mixed", 700001, 11))
    for t, e, bs, jobs in [("BWT+MTFT+ZRLT", "ANS0", 65536, 1), ("BWT+SRT+ZRLT", "ANS0", 262144, 3)]:
        enc = sh.DeviceRunEncoder(0, t, e, bs, jobs=jobs, orig_size=len(d))
        ranges = sh.block_ranges(len(d), bs, 2)
        runs = []
        for r, (first, cnt) in enumerate(ranges):
            chunk = d[first * bs:min(len(d), (first + cnt) * bs)]
            runs.append(enc(chunk, first, r == 0, r == 1))
        got = sh.concat_bit_runs(runs)[0]
        rc, ref = oracle.compress(d, t, e, bs, orig_size=len(d), jobs=jobs)
        assert got == ref, (t, e, jobs)


def test_sharded_decode_ranges(hip, oracle):
    # decode side of the multi-GPU path on one device: the host walks the length prefixes, each "rank" decodes its own
    # contiguous range of blocks from its own slice of the stream, the ranges are placed in order
    sh = importlib.import_module("kanzi_amd.sharded")
    d = vectors.make(("mixed", 700001, 11))
    for t, e, bs, world in [("BWT+MTFT+ZRLT", "ANS0", 65536, 3), ("NONE", "HUFFMAN", 16384, 2), ("BWT+SRT+ZRLT", "ANS0", 262144, 4)]:
        rc, ref = oracle.compress(d, t, e, bs, orig_size=len(d))
        assert rc == 0
        dec = sh.DeviceRunDecoder(0)
        parts = [None] * world
        for r in range(world):
            def gather(obj, r=r):
                parts[r] = obj
                return None
            sh.decompress_sharded(ref, r, world, dec, gather)
        assert b"".join(parts) == d, (t, e, world)


def test_block_checksums(hip, oracle):
    # -x32 / -x64 (util/XXHash.hpp); a corrupted payload must be rejected with ERR_CRC_CHECK (19)
    from kanzi_amd.hipapi import KnzError
    d = vectors.make(("mixed", 300000, 4))
    for ck in (32, 64):
        for t, e, bs in [("NONE", "ANS0", 65536), ("BWT+MTFT+ZRLT", "HUFFMAN", 131072)]:
            rc, ref = oracle.compress(d, t, e, bs, checksum=ck, headerless=1)
            out, bits, hb = gpu_compress(hip, d, t, e, bs, checksum=ck, headerless=1)
            assert out == ref, (ck, t, e)
            assert gpu_decompress(hip, ref, t, e, bs, len(d), 0, checksum=ck) == d
    # flip one stored checksum bit of block 0: 5-bit lw-3 field gives the width of the length field
    rc, ref = oracle.compress(d, "NONE", "NONE", 65536, checksum=32, headerless=1)
    bad = bytearray(ref)
    lw = 3 + (bad[0] >> 3)
    ck_bit = 5 + lw + 8 + 24                                   # mode byte + 3-byte length (65536 -> dataSize 3)
    bad[(ck_bit + 7) // 8] ^= 0x10
    with pytest.raises(KnzError) as ei:
        gpu_decompress(hip, bytes(bad), "NONE", "NONE", 65536, len(d), 0, checksum=32)
    assert ei.value.code == 19


def test_corrupted_streams_fail_cleanly(hip, oracle):
    # src/test/TestMalformedStream.cpp in spirit: flipped bytes, overwritten ranges and truncation must come back as
    # an error code (or as wrong bytes when the damage is undetectable), never as a hang or a device fault
    rng = np.random.default_rng(11)
    d = vectors.make(("mixed", 200000, 3))
    cases = [("NONE", "ANS0", 65536), ("NONE", "ANS1", 65536), ("NONE", "HUFFMAN", 65536), ("NONE", "FPAQ", 16384),
             ("BWT+MTFT+ZRLT", "ANS0", 65536), ("BWT+SRT+ZRLT", "HUFFMAN", 65536), ("RLT+ZRLT", "ANS0", 65536),
             ("LZX", "NONE", 65536), ("LZ", "ANS0", 65536)]
    errors = 0
    for t, e, bs in cases:
        rc, ref = oracle.compress(d, t, e, bs, headerless=1)
        p = hip.params(t, e, bs)
        d_enc, d_dec = hip.malloc(len(ref) + 4096), hip.malloc(len(d) + 2 * bs + 64)
        for r in range(6):
            buf = bytearray(ref)
            if r % 3 == 0:
                for _ in range(1 + r):
                    buf[int(rng.integers(0, len(buf)))] ^= int(rng.integers(1, 256))
            elif r % 3 == 1:
                a = int(rng.integers(0, len(buf) - 64))
                buf[a:a + 64] = bytes(rng.integers(0, 256, 64, dtype=np.uint8))
            else:
                buf = buf[:int(rng.integers(8, len(buf)))]
            hip.h2d(d_enc, bytes(buf) + bytes(64))
            try:
                hip.decode_blocks(p, d_enc, 8 * len(buf), 0, d_dec, len(d) + bs)
            except Exception as ex:
                assert "knz_hip error" in str(ex)
                errors += 1
        hip.free(d_enc)
        hip.free(d_dec)
    assert errors > 0
    # the context is still usable afterwards
    out, bits, hb = gpu_compress(hip, d, "NONE", "ANS0", 65536, headerless=1)
    assert gpu_decompress(hip, out, "NONE", "ANS0", 65536, len(d), 0) == d


def test_repeated_round_trips_are_stable(hip):
    # intermittent faults (races, unsynchronised prefetches) show up as a differing stream or a failed decode
    d = vectors.make(("mixed", 24 << 20, 2))
    first = None
    for i in range(40):
        out, bits, hb = gpu_compress(hip, d, "NONE", "ANS0", 4 << 20, headerless=1)
        if first is None:
            first = out
        assert out == first, i
        if i % 8 == 0:
            assert gpu_decompress(hip, out, "NONE", "ANS0", 4 << 20, len(d), 0) == d, i


def test_randomised_soak(hip):
    """Random chains / codecs / block sizes / inputs / jobs / checksums against the oracle (tools/gpu_soak.py).
    What the soak leaves out, and why: (1) the chains BWT+ZRLT and BWT+RLT+ZRLT -- when their second stage
    is skipped or expands, the reference writes a stream that the reference itself cannot decode ("Block 1 incorrectly
    decompressed", checked with oracle/_ref/kanzi), so there is no reference behaviour to be bit-exact with; (2) chains that START
    with ZRLT / RLT / SRT on incompressible (`rand`) input -- an expanding even-indexed stage makes the reference write past the
    logical end of its own buffer (undefined behaviour, output depends on allocator state). Every BASELINE chain starts with BWT,
    whose 33-byte header keeps both cases away. Streams the reference emits but cannot decode for a third reason (the frequency
    normalisation corner pinned in tests/golden/quirks.json) are counted separately by the tool: refusing them is the parity."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_soak.py"), "5", "25"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
