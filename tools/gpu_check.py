"""Ad-hoc GPU parity probe (developer tool): HIP path vs oracle, prints first mismatch."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import knzlib
import numpy as np

pkg = knzlib.load_pkg()
import importlib
hipapi = importlib.import_module("kanzi_amd.hipapi")
O = knzlib.Oracle()
c = knzlib.corpus()
ctx = hipapi.Context(0)
rng = np.random.default_rng(3)

def first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return i
    return n if len(a) != len(b) else -1

def datasets():
    yield "text64k", c.text(65536, 1)
    yield "text1m", c.text(1 << 20, 1)
    yield "mixed1m+5", c.mixed((1 << 20) + 5, 2)
    yield "rand100k", rng.integers(0, 256, 100000, dtype=np.uint8).tobytes()
    yield "zeros70k", bytes(70000)
    yield "small33", bytes(range(33))
    yield "small20", b"abcabcabcabcabcabcab"
    yield "n4096", c.text(4096, 3)
    yield "n16385", c.text(16385, 4)
    yield "n16387", c.text(16387, 4)
    yield "skew", bytes((rng.geometric(0.3, 200001) % 256).astype(np.uint8))
    yield "twosym", bytes((rng.integers(0, 2, 50000) * 7).astype(np.uint8))

ents = sys.argv[1].split(",") if len(sys.argv) > 1 else ["NONE", "ANS0"]
bad = 0
for name, d in datasets():
    for e in ents:
        ref, rbits = O.entropy_encode(e, d)
        try:
            got, gbits = ctx.entropy_encode(e, d)
        except Exception as ex:
            print(name, e, "ENC EXC", ex); bad += 1; continue
        ok = (got == ref and gbits == rbits)
        msg = ""
        if not ok:
            bad += 1
            msg = "first diff byte %d (bits %d vs %d)" % (first_diff(got, ref), gbits, rbits)
        try:
            dec, out, used = ctx.entropy_decode(e, ref, len(d))
            dok = (dec == len(d) and out == d and used == rbits)
            if not dok:
                bad += 1
                msg += " | dec=%d used=%d/%d firstdiff=%d" % (dec, used, rbits, first_diff(out, d))
        except Exception as ex:
            dok = False; bad += 1; msg += " | DEC EXC %s" % ex
        print("%-10s %-8s enc %s dec %s %s" % (name, e, ok, dok, msg))

# stream-level: blocks + framing
STREAMS = [("text4m", c.text(4 << 20, 1), 1 << 20), ("mixed3m+", c.mixed(3 * (1 << 20) + 12345, 2), 1 << 20),
           ("mixed4m", c.mixed(4 << 20, 2), 4 << 20), ("tiny", b"hello world!", 1024), ("mixed9m", c.mixed(9 * (1 << 20) + 3, 5), 16 << 20)]
for name, d, bs, e in [(nm, dd, b_, e_) for e_ in ents for (nm, dd, b_) in STREAMS]:
    rc, ref = O.compress(d, "NONE", e, bs, headerless=1)
    p = ctx.params("NONE", e, bs)
    cap = ctx.encode_bound(p, len(d))
    d_in = ctx.malloc(len(d) + 64); d_out = ctx.malloc(cap)
    ctx.h2d(d_in, d)
    t0 = time.time()
    bits = ctx.encode_blocks(p, d_in, len(d), d_out, cap)
    t1 = time.time()
    got = ctx.d2h(d_out, (bits + 7) // 8)
    ok = got == ref
    if not ok: bad += 1
    # decode the reference stream
    d_enc = ctx.malloc(len(ref) + 64); d_dec = ctx.malloc(len(d) + bs + 64)
    ctx.h2d(d_enc, ref)
    ob, eb, nb = ctx.decode_blocks(p, d_enc, 8 * len(ref), 0, d_dec, len(d) + bs)
    back = ctx.d2h(d_dec, ob)
    dok = back == d
    if not dok: bad += 1
    print("stream %-12s %s bs=%d enc %s (%d vs %d bytes, first diff %d) dec %s (blocks %d, endbit %d/%d) %.1f ms" % (
        name, e, bs, ok, len(got), len(ref), first_diff(got, ref), dok, nb, eb, 8 * len(ref), (t1 - t0) * 1e3))
    for pp in (d_in, d_out, d_enc, d_dec): ctx.free(pp)
print("BAD", bad)
sys.exit(1 if bad else 0)
