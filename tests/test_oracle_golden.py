"""The oracle (oracle/*.c) against the golden vectors generated from the unmodified reference
(tests/golden/golden.json) and against the known-answer values the reference's own tests hold."""
import hashlib

import pytest

import knzlib
import vectors


def matches(packed, b):
    if packed["len"] != len(b):
        return False
    if "hex" in packed:
        return packed["hex"] == b.hex()
    return packed["md5"] == hashlib.md5(b).hexdigest()


def test_entropy_stage_vectors(oracle, golden):
    n = 0
    for rec in golden["stages"]:
        if rec["kind"] != "entropy":
            continue
        d = vectors.make(tuple(rec["input"]))
        enc, bits = oracle.entropy_encode(rec["name"], d)
        assert bits == rec["bits"], rec
        assert matches(rec["out"], enc), rec["input"]
        r, dec = oracle.entropy_decode(rec["name"], enc, len(d))
        assert r == len(d) and dec == d
        n += 1
    assert n > 100


def test_reference_quirks(oracle):
    """Inputs on which the reference misbehaves (tests/golden/quirks.json): the oracle reproduces the encoder's
    output bit for bit and fails to decode it exactly as the reference does."""
    import json, os
    recs = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quirks.json")))
    assert recs
    for rec in recs:
        d = bytes.fromhex(rec["input_hex"])
        enc, bits = oracle.entropy_encode(rec["entropy"], d)
        assert bits == rec["bits"] and hashlib.md5(enc).hexdigest() == rec["enc_md5"], rec["name"]
        r = oracle.entropy_decode(rec["entropy"], enc, len(d))
        got = len(d) if (r[0] == len(d) and r[1] == d) else -1
        assert got == rec["ref_decoded"], rec["name"]


def test_transform_stage_vectors(oracle, golden):
    n = 0
    for rec in golden["stages"]:
        if rec["kind"] != "transform":
            continue
        d = vectors.make(tuple(rec["input"]))
        ok, out = oracle.forward(rec["name"], d, rec["cap"], rec["entropy"] or None)
        assert int(bool(ok)) == rec["ok"], rec
        if rec["ok"]:
            assert matches(rec["out"], out), (rec["name"], rec["input"])
            k, back = oracle.inverse(rec["name"], out, len(d) + 64)
            assert k == 1 and back == d, (rec["name"], rec["input"])
        n += 1
    assert n > 100


def test_stream_vectors(oracle, golden):
    for rec in golden["streams"]:
        d = vectors.make(tuple(rec["input"]))
        rc, out = oracle.compress(d, rec["transform"], rec["entropy"], rec["block"], rec["checksum"], rec["orig_size"],
                                  rec["headerless"])
        assert rc == 0
        assert matches(rec["out"], out), rec
        if not rec["headerless"]:
            rc, back = oracle.decompress(out, len(d) + 16)
            assert rc == 0 and back == d


# ---- known answers held by the reference's own tests
def test_bwt_known_strings(oracle):
    # src/test/TestBWT.cpp:42-60 strings; expected values: transform/BWT.hpp:40-55 and SURVEY App. B
    cases = [(b"mississippi", b"ipssmpissii", 5),
             (b"3.14159265358979323846264338327950288419716939937510", b"03155.4743693098459329186897631422193923832557967183", 14),
             (b"SIX.MIXED.PIXIES.SIFT.SIXTY.PIXIE.DUST.BOXES", b"STEXYDST.E.IXXIIXXSSMPPS.B..EE..USFXDIIOIIIT", 31)]
    for src, exp, pidx in cases:
        ok, out, prim = oracle.bwt_raw(src)
        assert ok and out == exp and prim[0] == pidx


def test_kat32_transforms(oracle):
    # src/test/TestTransforms.cpp:889-899, expected outputs SURVEY App. B
    d = vectors.make(("kat32",))
    ok, out = oracle.forward("ZRLT", d, len(d))
    assert ok and out == bytes([0, 2, 3, 3, 3, 3, 8, 10, 10, 17, 17, 17, 2] + [4] * 19)
    ok, out = oracle.forward("MTFT", d)
    assert ok and out == bytes([0, 1, 2, 0, 0, 0, 7, 9, 0, 16, 0, 0, 4, 6] + [0] * 18)
    ok, out = oracle.forward("RLT", d)
    assert ok and out == bytes([4, 0, 1, 2, 4, 1, 7, 9, 9, 16, 16, 16, 1, 3, 4, 14, 3, 3])
    ok, out = oracle.forward("SRT", d, len(d) + 100)        # capacity < n + 1024 => refuses
    assert not ok


def test_zrlt_reference_kats(oracle):
    # src/test/TestTransforms.cpp:379-494 (testZRLTMalformed)
    ok, out = oracle.inverse("ZRLT", bytes([2]), 1)
    assert ok and out == bytes([1])
    ok, out = oracle.inverse("ZRLT", bytes([2, 2]), 1)           # capacity 1: second literal does not fit
    assert not ok
    ok, out = oracle.inverse("ZRLT", bytes([0xFF]), 1)           # truncated escape
    assert not ok
    ok, out = oracle.inverse("ZRLT", bytes([0]), 0)              # zero run into empty output
    assert not ok
    ok, out = oracle.forward("ZRLT", bytes([0xFE]), 1)           # escaped byte needs 2
    assert not ok
    ok, out = oracle.forward("ZRLT", bytes([0]), 1)
    assert ok and out == bytes([0])


def test_type_tables(oracle):
    # src/test/TestFactories.cpp ; SURVEY 8(a) a5: BWT+MTFT+ZRLT -> 0x47180000000
    assert oracle.L.knzo_transform_type(b"BWT+MTFT+ZRLT") == 0x47180000000
    assert oracle.L.knzo_transform_type(b"none") == 0
    assert oracle.L.knzo_transform_type(b"NONE+BWT") == 1 << 42
    assert oracle.L.knzo_transform_type(b"A+B") == 0xFFFFFFFFFFFFFFFF
    assert oracle.L.knzo_transform_type(b"BWT+BWT+BWT+BWT+BWT+BWT+BWT+BWT+BWT") == 0xFFFFFFFFFFFFFFFF
    assert oracle.L.knzo_entropy_type(b"ans0") == 5 and oracle.L.knzo_entropy_type(b"FPAQ") == 2
    assert oracle.L.knzo_entropy_type(b"bogus") == -1


def test_truncated_payloads_rejected(oracle):
    # src/test/TestEntropyCodec.cpp:230-307: truncated / shrunk payloads must not decode
    d = vectors.make(("formula13", 4096))
    for e in ["HUFFMAN", "ANS0", "FPAQ"]:
        enc, bits = oracle.entropy_encode(e, d)
        r, dec = oracle.entropy_decode(e, enc[:len(enc) // 2], len(d))
        assert r != len(d) or dec != d


def test_malformed_stream_headers(oracle):
    # src/test/TestMalformedStream.cpp: bad magic / version / checksum must raise the right code
    d = vectors.make(("text", 5000, 1))
    rc, s = oracle.compress(d, "NONE", "ANS0", 1024)
    assert rc == 0
    bad = bytearray(s); bad[0] ^= 1
    assert oracle.decompress(bytes(bad), 6000)[0] == 15          # ERR_INVALID_FILE
    bad = bytearray(s); bad[19] ^= 1
    assert oracle.decompress(bytes(bad), 6000)[0] == 19          # ERR_CRC_CHECK (header checksum)
    bad = bytearray(s); bad[4] = (bad[4] & 0x0F) | 0x70
    assert oracle.decompress(bytes(bad), 6000)[0] == 16          # ERR_STREAM_VERSION
