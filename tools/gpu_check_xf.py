"""Ad-hoc GPU parity probe for transforms: HIP path vs oracle."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import knzlib, vectors
import numpy as np
import importlib
knzlib.load_pkg()
hipapi = importlib.import_module("kanzi_amd.hipapi")
O = knzlib.Oracle()
ctx = hipapi.Context(0)

def first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]: return i
    return n if len(a) != len(b) else -1

names = sys.argv[1].split(",")
chains = sys.argv[2].split(",") if len(sys.argv) > 2 else []
specs = vectors.STAGE_INPUTS + [("mixed", 1 << 20, 2), ("text", (1 << 20) + 3, 1), ("const", 300000, 0), ("runs", 9000, 300)]
bad = 0
for spec in specs:
    d = vectors.make(spec)
    if len(d) == 0: continue
    for t in names:
        caps = [len(d), len(d) + 2048] if t in ("ZRLT", "RLT") else [len(d) + 2048]
        if t in ("LZ", "LZX"): caps = [len(d) + len(d) // 64 + 64, len(d) + 17]
        for cap in caps:
            ok1, o1 = O.forward(t, d, cap, "ANS0")
            try:
                t0 = time.time()
                ok2, o2 = ctx.transform_forward(t, d, cap, "ANS0")
                if "-t" in sys.argv: print("   %s %s n=%d forward %.1f ms" % (str(spec)[:30], t, len(d), 1e3 * (time.time() - t0)))
            except Exception as ex:
                print(spec, t, "FWD EXC", ex); bad += 1; continue
            good = (bool(ok1) == bool(ok2)) and (not ok1 or o1 == o2)
            msg = ""
            if not good:
                bad += 1; msg = "ok %d/%d len %d/%d firstdiff %d" % (ok1, ok2, len(o1), len(o2), first_diff(o1, o2))
            ig = True
            if ok1:
                icap = max(len(d), len(o1)) + 64
                k1, b1 = O.inverse(t, o1, icap)
                t0 = time.time()
                k2, b2 = ctx.transform_inverse(t, o1, icap)
                if "-t" in sys.argv: print("   %s %s n=%d inverse %.1f ms" % (str(spec)[:30], t, len(d), 1e3 * (time.time() - t0)))
                ig = (bool(k1) == bool(k2)) and (not k1 or b1 == b2) and (b1 == d)
                if not ig:
                    bad += 1; msg += " | inv ok %d/%d len %d/%d firstdiff %d" % (k1, k2, len(b1), len(b2), first_diff(b1, b2))
                # exact-capacity inverse
                k1, b1 = O.inverse(t, o1, len(d)); k2, b2 = ctx.transform_inverse(t, o1, len(d))
                if (bool(k1) != bool(k2)) or (k1 and b1 != b2):
                    bad += 1; msg += " | exactcap inv ok %d/%d" % (k1, k2)
            if msg or "-v" in sys.argv:
                print("%-40s %-5s cap=%d fwd %s inv %s %s" % (str(spec)[:40], t, cap, good, ig, msg))
for chain in chains:
    for spec, bs, e in [(("mixed", 3 * (1 << 20) + 12345, 2), 1 << 20, "ANS0"), (("text", 1 << 20, 1), 1 << 18, "HUFFMAN"),
                        (("rand", 300000, 9), 1 << 16, "ANS0"), (("const", 100000, 0), 65536, "ANS0"), (("ramp", 1025), 1024, "ANS0"),
                        (("mixed", 700001, 11), 262144, "ANS0")]:
        d = vectors.make(spec)
        for jobs in (1, 3):
            rc, ref = O.compress(d, chain, e, bs, headerless=1, jobs=jobs)
            p = ctx.params(chain, e, bs, jobs=jobs)
            cap = ctx.encode_bound(p, len(d)) + 64
            d_in = ctx.malloc(len(d) + 64); d_out = ctx.malloc(cap)
            ctx.h2d(d_in, d)
            try:
                bits = ctx.encode_blocks(p, d_in, len(d), d_out, cap)
                got = ctx.d2h(d_out, (bits + 7) // 8)
            except Exception as ex:
                print("stream", chain, spec, "ENC EXC", ex); bad += 1; continue
            ok = got == ref
            d_enc = ctx.malloc(len(ref) + 64); d_dec = ctx.malloc(len(d) + bs + 64)
            ctx.h2d(d_enc, ref)
            try:
                ob, eb, nb = ctx.decode_blocks(p, d_enc, 8 * len(ref), 0, d_dec, len(d) + bs)
                back = ctx.d2h(d_dec, ob)
                dok = back == d
            except Exception as ex:
                dok = False; print("  DEC EXC", ex)
            if not ok: bad += 1
            if not dok: bad += 1
            print("stream %-16s %-28s %s bs=%d j=%d enc %s (%d vs %d, firstdiff %d) dec %s" % (chain, str(spec)[:28], e, bs, jobs, ok, len(got), len(ref), first_diff(got, ref), dok))
            for pp in (d_in, d_out, d_enc, d_dec): ctx.free(pp)
print("BAD", bad)
