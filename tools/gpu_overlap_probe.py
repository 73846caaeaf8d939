"""How well do the stages of different blocks overlap when they run side by side? (developer tool) Two contexts on one GPU, one encoding
with -t BWT -e NONE, the other with -t MTFT+ZRLT -e ANS0 on BWT output of the same size: alone, and at the same time from two threads."""
import importlib, os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import knzlib
knzlib.load_pkg()
hipapi = importlib.import_module("kanzi_amd.hipapi")
c = knzlib.corpus()
n = 211957760
d = c.mixed(n, 2)
bs = 8 << 20
a, b = hipapi.Context(0), hipapi.Context(0)
def setup(ctx, data, t, e):
    p = ctx.params(t, e, bs)
    cap = ctx.encode_bound(p, len(data)) + 64
    di, do = ctx.malloc(len(data) + 64), ctx.malloc(cap)
    ctx.h2d(di, data)
    return p, di, do, cap
pa, ia, oa, ca = setup(a, d, "BWT", "NONE")
bits = a.encode_blocks(pa, ia, n, oa, ca)
bw = a.d2h(oa, (bits + 7) // 8)[64:64 + n]          # roughly BWT output (framing bytes inside do not matter for timing)
bw = bw + bytes(n - len(bw))
pb, ib, ob, cb = setup(b, bw, "MTFT+ZRLT", "ANS0")
def run(ctx, p, i, o, cap, reps, out):
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.encode_blocks(p, i, n, o, cap)
    out.append((time.perf_counter() - t0) / reps * 1e3)
for _ in range(2):
    ra, rb = [], []
    run(a, pa, ia, oa, ca, 3, ra); run(b, pb, ib, ob, cb, 3, rb)
    both_a, both_b = [], []
    t0 = time.perf_counter()
    ta = threading.Thread(target=run, args=(a, pa, ia, oa, ca, 3, both_a)); tb = threading.Thread(target=run, args=(b, pb, ib, ob, cb, 3, both_b))
    ta.start(); tb.start(); ta.join(); tb.join()
    wall = (time.perf_counter() - t0) / 3 * 1e3
    print("alone: BWT %.2f ms, MTFT+ZRLT+ANS0 %.2f ms (sum %.2f); together: %.2f ms per pair (BWT %.2f, rest %.2f)" % (ra[0], rb[0], ra[0] + rb[0], wall, both_a[0], both_b[0]), flush=True)
