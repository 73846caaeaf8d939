"""Developer tool (GPU box): isolates which stage of a chain fails at full block size."""
import importlib, os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import knzlib
knzlib.load_pkg()
hipapi = importlib.import_module("kanzi_amd.hipapi")
c = knzlib.corpus()
O = knzlib.Oracle()
hip = hipapi.Context(0)
d = c.text(12 << 20, 1)
for n in (3 << 20, (4 << 20) + 5, 9 << 20):
    x = d[:n]
    enc, bits = hip.entropy_encode("FPAQ", x)
    ref, rbits = O.entropy_encode("FPAQ", x)
    print("fpaq encode n=%d bits %d ref %d same=%s" % (n, bits, rbits, enc == ref), flush=True)
    dec, out, used = hip.entropy_decode("FPAQ", ref, n)
    print("  decode: ret %d same=%s used %d" % (dec, out == x, used), flush=True)
    if out != x and dec == n:
        at = next(i for i in range(n) if out[i] != x[i]); print("  first diff at", at)
# transforms at 32 MiB
big = c.text(32 << 20, 1)
ok, t1 = hip.transform_forward("BWT", big, len(big) + 64)
print("bwt fwd ok", ok, len(t1), hashlib.md5(t1).hexdigest(), flush=True)
ok2, back = hip.transform_inverse("BWT", t1, len(big) + 64)
print("bwt inv ok", ok2, back == big, flush=True)
ok, t2 = hip.transform_forward("SRT", t1, len(t1) + 2048)
ok2, b2 = hip.transform_inverse("SRT", t2, len(t1) + 64)
print("srt", ok, ok2, b2 == t1, flush=True)
ok, t3 = hip.transform_forward("ZRLT", t2, len(t2))
ok2, b3 = hip.transform_inverse("ZRLT", t3, len(t2) + 64)
print("zrlt", ok, ok2, b3 == t2, len(t3), flush=True)
enc, bits = hip.entropy_encode("FPAQ", t3)
dec, out, used = hip.entropy_decode("FPAQ", enc, len(t3))
print("fpaq on chain data: dec", dec, out == t3, used, bits, flush=True)
