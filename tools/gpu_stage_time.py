"""Per-stage transform timing through the host-buffer entry points (developer tool; includes staging copies)."""
import sys, os, time, importlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import knzlib, vectors
import numpy as np
knzlib.load_pkg()
hipapi = importlib.import_module("kanzi_amd.hipapi")
ctx = hipapi.Context(0)
rng = np.random.default_rng(1)
def runs(n):
    out = bytearray()
    while len(out) < n:
        out += bytes([int(rng.integers(0, 256))]) * int(rng.geometric(0.02 if rng.random() < 0.3 else 0.4))
    return bytes(out[:n])
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["RLT", "ZRLT", "SRT", "MTFT"]
for name, d in (("runs4m", runs(4 << 20)), ("text4m", vectors.make(("text", 4 << 20, 1)))):
    for t in names:
        ctx.transform_forward(t, d, len(d) + 2048, "ANS0")
        t0 = time.time(); ok, o = ctx.transform_forward(t, d, len(d) + 2048, "ANS0"); t1 = time.time()
        if ok:
            t2 = time.time(); k, b = ctx.transform_inverse(t, o, len(d) + 64); t3 = time.time()
        else: t2 = t3 = 0; k = False; b = b""
        print("%-8s %-5s fwd ok=%s %.1f ms  inv ok=%s %.1f ms (%s)" % (name, t, ok, (t1 - t0) * 1e3, k, (t3 - t2) * 1e3, b == d))
