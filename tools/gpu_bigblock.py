import sys, os, time, importlib, hashlib
sys.path.insert(0, "/root/repo/tests")
import knzlib, vectors
knzlib.load_pkg()
hipapi = importlib.import_module("kanzi_amd.hipapi")
O = knzlib.Oracle(); ctx = hipapi.Context(0)
def run(t, e, bs, n, seed=3):
    d = vectors.make(("mixed", n, seed))
    t0 = time.time(); rc, ref = O.compress(d, t, e, bs, headerless=1); t1 = time.time()
    p = ctx.params(t, e, bs)
    cap = ctx.encode_bound(p, len(d)) + 64
    d_in = ctx.malloc(len(d) + 64); d_out = ctx.malloc(cap); ctx.h2d(d_in, d)
    try:
        t2 = time.time(); bits = ctx.encode_blocks(p, d_in, len(d), d_out, cap); t3 = time.time()
        got = ctx.d2h(d_out, (bits + 7) // 8)
        ok = got == ref
        d_dec = ctx.malloc(len(d) + bs + 64)
        ob, eb, nb = ctx.decode_blocks(p, d_out, bits, 0, d_dec, len(d) + bs)
        dok = ctx.d2h(d_dec, ob) == d
        ctx.free(d_dec)
    except Exception as ex:
        ok = dok = False; t2 = t3 = 0; print("EXC", ex)
    ctx.free(d_in); ctx.free(d_out)
    print("%-14s %-7s bs=%d n=%d enc %s dec %s (oracle %.1f s, gpu enc %.2f s)" % (t, e, bs, n, ok, dok, t1 - t0, t3 - t2), flush=True)
which = sys.argv[1]
if which == "a":
    run("NONE", "NONE", 640 << 20, (640 << 20) + 1000)
    run("NONE", "ANS0", 640 << 20, 640 << 20)
    run("NONE", "HUFFMAN", 640 << 20, 600 << 20)
elif which == "b":
    run("BWT+MTFT+ZRLT", "ANS0", 256 << 20, 256 << 20)
elif which == "c":
    run("ZRLT", "NONE", 600 << 20, 600 << 20)
    run("MTFT", "ANS0", 600 << 20, 600 << 20)
