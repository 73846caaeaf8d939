"""Corrupt valid streams and decode them: the device path must answer with an error code or (rarely) wrong
bytes, never hang or fault (developer tool; a bounded version runs in tests/test_gpu_parity.py)."""
import sys, os, time, importlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import knzlib, vectors
import numpy as np

knzlib.load_pkg()
hipapi = importlib.import_module("kanzi_amd.hipapi")
O = knzlib.Oracle()
ctx = hipapi.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
CASES = [("NONE", "ANS0", 65536), ("NONE", "ANS1", 65536), ("NONE", "HUFFMAN", 65536), ("NONE", "FPAQ", 16384),
         ("BWT+MTFT+ZRLT", "ANS0", 65536), ("BWT+SRT+ZRLT", "HUFFMAN", 65536), ("RLT+ZRLT", "ANS0", 65536), ("SRT", "NONE", 65536),
         ("BWT", "NONE", 65536), ("MTFT", "NONE", 65536), ("ZRLT", "NONE", 65536), ("RLT", "NONE", 65536),
         ("LZX", "NONE", 65536), ("LZ", "ANS0", 65536), ("LZX", "ANS1", 262144)]
d = vectors.make(("mixed", 200000, 3))
stats = {"ok_same": 0, "ok_diff": 0, "error": 0}
t0 = time.time()
for t, e, bs in CASES:
    rc, ref = O.compress(d, t, e, bs, headerless=1)
    p = ctx.params(t, e, bs)
    d_enc = ctx.malloc(len(ref) + 4096); d_dec = ctx.malloc(len(d) + 2 * bs + 64)
    for r in range(rounds):
        buf = bytearray(ref)
        mode = r % 3
        if mode == 0:
            for _ in range(1 + r):
                buf[int(rng.integers(0, len(buf)))] ^= int(rng.integers(1, 256))
        elif mode == 1:
            a = int(rng.integers(0, len(buf) - 64)); buf[a:a + 64] = bytes(rng.integers(0, 256, 64, dtype=np.uint8))
        else:
            buf = buf[:int(rng.integers(8, len(buf)))]
        ctx.h2d(d_enc, bytes(buf) + bytes(64))
        t1 = time.time()
        try:
            ob, eb, nb = ctx.decode_blocks(p, d_enc, 8 * len(buf), 0, d_dec, len(d) + bs)
            back = ctx.d2h(d_dec, ob) if ob else b""
            stats["ok_same" if back == d else "ok_diff"] += 1
        except hipapi.KnzError as ex:
            stats["error"] += 1
        dt = time.time() - t1
        if dt > 2.0:
            print("SLOW", t, e, "round", r, "%.1f s" % dt)
    ctx.free(d_enc); ctx.free(d_dec)
    print("%-16s %-8s done" % (t, e), flush=True)
# the context must still work
rc, ref = O.compress(d, "NONE", "ANS0", 65536, headerless=1)
p = ctx.params("NONE", "ANS0", 65536)
d_enc = ctx.malloc(len(ref) + 64); d_dec = ctx.malloc(len(d) + 65536 + 64); ctx.h2d(d_enc, ref)
ob, eb, nb = ctx.decode_blocks(p, d_enc, 8 * len(ref), 0, d_dec, len(d) + 65536)
assert ctx.d2h(d_dec, ob) == d
print("fuzz", stats, "%.1f s" % (time.time() - t0), "context still healthy")
