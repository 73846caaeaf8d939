"""The stages of the level presets 5 and 6 that run on the host (kanzi-cpp_amd/host/text_codec.cpp: TEXT in both encodings, UTF): their
output against the reference's (digests in tests/golden/host_stages.json, from oracle/_ref), their inverses, and -- where the compiled
reference is at hand -- a randomised comparison block by block. No GPU involved: these run in front of the device chain."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import knzlib
import vectors

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def stages():
    L = C.CDLL(os.path.join(knzlib.PKG, "libkanzi_amd.so"))
    u8p = C.POINTER(C.c_uint8)
    L.knz_host_text_forward.argtypes = [C.c_int, u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.knz_host_text_inverse.argtypes = [C.c_int, u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.knz_host_utf_forward.argtypes = [u8p, C.c_int, u8p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.knz_host_utf_inverse.argtypes = [u8p, C.c_int, u8p, C.c_int, C.POINTER(C.c_int)]

    class S:
        @staticmethod
        def text(variant, d, block_size=0, data_type=0):
            out = (C.c_uint8 * (len(d) + 64))(); dt = C.c_int(data_type); ol = C.c_int(0)
            ok = L.knz_host_text_forward(variant, knzlib._buf(d), len(d), out, len(d), block_size, 6, C.byref(dt), C.byref(ol))
            return ok, C.string_at(out, ol.value), dt.value

        @staticmethod
        def text_inv(variant, e, cap, block_size=0):
            out = (C.c_uint8 * (cap + 64))(); ol = C.c_int(0)
            ok = L.knz_host_text_inverse(variant, knzlib._buf(e), len(e), out, cap, block_size, 6, C.byref(ol))
            return ok, C.string_at(out, ol.value)

        @staticmethod
        def utf(d, data_type=0):
            out = (C.c_uint8 * (len(d) + 8192 + 64))(); dt = C.c_int(data_type); ol = C.c_int(0)
            ok = L.knz_host_utf_forward(knzlib._buf(d), len(d), out, len(d) + 8192, C.byref(dt), C.byref(ol))
            return ok, C.string_at(out, ol.value), dt.value

        @staticmethod
        def utf_inv(e, cap):
            out = (C.c_uint8 * (cap + 64))(); ol = C.c_int(0)
            ok = L.knz_host_utf_inverse(knzlib._buf(e), len(e), out, cap, C.byref(ol))
            return ok, C.string_at(out, ol.value)
    return S


def test_host_stages_match_the_reference_fixture(stages):
    recs = json.load(open(os.path.join(HERE, "golden", "host_stages.json")))
    applied = {"text1": 0, "text2": 0, "utf": 0}
    for rec in recs:
        d = vectors.make(tuple(rec["input"]))
        assert hashlib.md5(d).hexdigest() == rec["input_md5"], rec["input"]
        for name, want in rec["stages"].items():
            if name == "utf":
                ok, out, _ = stages.utf(d)
            else:
                ok, out, _ = stages.text(1 if name == "text1" else 2, d)
            assert ok == want["applied"], (rec["input"], name)
            if not ok:
                continue
            applied[name] += 1
            assert len(out) == want["len"] and hashlib.md5(out).hexdigest() == want["md5"], (rec["input"], name)
            if name == "utf":
                k, back = stages.utf_inv(out, len(d) + 8)
            else:
                k, back = stages.text_inv(1 if name == "text1" else 2, out, len(d))
            assert k == 1 and back == d, (rec["input"], name)
    assert min(applied.values()) >= 3, applied


def test_host_stage_data_types(stages):
    """what a refused block tells the next stage (Global::DataType): UTF-8, DNA, digits, base64, binary; and what a stage refuses to look at"""
    c = knzlib.corpus()
    assert stages.text(2, vectors.make(("utf8", 60000, 7)))[2] == 8                 # UTF8
    assert stages.text(2, vectors.make(("dna", 50000, 4)))[2] == 6                  # DNA
    assert stages.text(1, bytes(np.random.default_rng(1).integers(48, 58, 5000, dtype=np.uint8)))[2] == 4       # NUMERIC
    assert stages.text(1, vectors.make(("rand", 50000, 7)))[2] == 7                 # BIN (all 256 byte values)
    t = c.text(5000, 1)
    assert stages.text(2, t)[0] == 1 and stages.text(2, t)[2] == 1                  # TEXT
    assert stages.text(2, t, data_type=8)[0] == 0                                   # a block known to be UTF-8 is not tried as text
    assert stages.utf(t, data_type=1)[0] == 0                                       # nor text as UTF-8
    assert stages.text(2, b"\x89PNG" + t)[0] == 0 and stages.text(1, b"\x89PNG" + t)[0] == 1       # only the quick test looks at magic numbers


def test_host_stages_against_the_compiled_reference(stages):
    if knzlib.ensure_ref() is None:
        pytest.skip("oracle/_ref not built (no reference sources on this box)")
    R = knzlib.Ref()
    rng = np.random.default_rng(11)
    c = knzlib.corpus()
    words = [b"the", b"quick", b"Brown", b"fox", b"JUMPS", b"over", b"lazy", b"dog", b"People", b"because", b"xyzzy", b"compress", b"a", b"it's",
             b"\x0f", b"\x0e", b"na\xc3\xafve", b"\xe2\x82\xac", b"12345", b"foo_bar", b"<tag>", b"&amp;", b"</tag>"]
    seps = [b" ", b" ", b" ", b", ", b". ", b"\n", b"\r\n", b"  ", b"\t"]
    n_text = n_utf = 0
    for it in range(60):
        kind = it % 5
        n = int(rng.choice([1024, 1500, 4000, 30000, 120000]))
        if kind == 0:
            d = c.text(n, int(rng.integers(1, 999)))
        elif kind == 1:
            out = bytearray()
            while len(out) < n:
                w = words[int(rng.integers(0, len(words)))]
                if rng.random() < 0.3:
                    w = bytes(rng.integers(97, 123, int(rng.integers(2, 12)), dtype=np.uint8))
                out += w + seps[int(rng.integers(0, len(seps)))]
            d = bytes(out[:n])
        elif kind == 2:
            d = vectors.make(("utf8", n, int(rng.integers(1, 99))))
        elif kind == 3:
            d = vectors.make(("crlf", n, int(rng.integers(1, 99))))
        else:
            d = c.mixed(n + 1000, int(rng.integers(1, 50)))[500:500 + n]
        if len(d) < 1024:
            continue
        for variant, ent in ((1, "FPAQ"), (2, "ANS0")):
            _, ref_out, sk = R.forward("TEXT", d, dst_cap=len(d), entropy=ent)
            ok, out, _ = stages.text(variant, d)
            assert ok == (0 if sk & 0x80 else 1), (kind, n, variant)
            if ok:
                n_text += 1
                assert out == ref_out, (kind, n, variant)
        _, ref_out, sk = R.forward("UTF", d, dst_cap=len(d) + 8192)
        ok, out, _ = stages.utf(d)
        assert ok == (0 if sk & 0x80 else 1), (kind, n)
        if ok:
            n_utf += 1
            assert out == ref_out, (kind, n)
    assert n_text >= 40 and n_utf >= 8, (n_text, n_utf)


def test_text_inverse_refuses_zero_and_truncated_word_indexes(stages):
    """a 2- or 3-byte word index that decodes to 0 (C0 00, F0 00 00 in the top-bit encoding) must be refused -- the
    reference reads _dictList[-1] there (transform/TextCodec.cpp:1494-1521) -- and an index cut off by the end of the block must not
    read behind the caller's slice. The slices are exact-size heap copies so that an over-read would be caught by a sanitizer build
    and at least cannot be fed by padding."""
    t = knzlib.corpus().text(6000, 3)
    ok, enc, _ = stages.text(2, t)
    assert ok == 1
    k, back = stages.text_inv(2, enc, len(t))
    assert k == 1 and back == t
    head = enc[:40]
    for tail in (b"\xc0\x00", b"\xf0\x00\x00", b"\x80\xc0\x00", b"\x80\xf0\x00\x00"):
        k, _ = stages.text_inv(2, head + tail + b" abc", len(t))
        assert k == 0, tail
    for tail in (b"\xc0", b"\xf0", b"\xf0\x01", b"\x80", b"\x80\xf0", b"\xe5"):       # the block ends inside an index
        k, _ = stages.text_inv(2, head + tail, len(t))
        assert k == 0, tail
    ok, enc1, _ = stages.text(1, t)
    assert ok == 1
    for tail in (b"\x0f", b"\x0f\x80", b"\x0f\x80\x80", b"\x0e\xff\xff"):             # escape encoding: cut off / beyond the dictionary
        k, _ = stages.text_inv(1, enc1[:40] + tail, len(t))
        assert k == 0, tail
    rng = np.random.default_rng(5)
    for enc_v, v in ((enc, 2), (enc1, 1)):                                              # random damage: never a crash, refusal or some output
        for _ in range(300):
            b = bytearray(enc_v[:int(rng.integers(2, len(enc_v)))])
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(1, len(b)))] = int(rng.integers(0, 256))
            stages.text_inv(v, bytes(b), len(t) + 64)
