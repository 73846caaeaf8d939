"""Study for DESIGN_HISTORY.md section 7: SBRT forward (MTF, RANK, TIMESTAMP) has a closed form that is independent across
positions -- the rank of c at i is the number of symbols whose (key, time of last update, -symbol) is larger, and
keys/times depend only on each symbol's last two occurrences before i. Checked here against the oracle (CPU only)."""
import sys
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "tests"))
import knzlib, vectors
O = knzlib.Oracle()
def par_rank(d, mode):
    # rank of c at i = number of symbols ahead of it in an order defined only by per-symbol history before i:
    # key descending, among equal keys the one updated later first, never-updated symbols by index at the end
    n = len(d); out = bytearray(n)
    last = [-1] * 256; prevl = [0] * 256      # p[c] is 0 before the first occurrence
    key = [0] * 256; upd = [-1] * 256
    for i in range(n):
        c = d[i]
        kc, uc = key[c], upd[c]
        r = 0
        for s in range(256):
            if s == c: continue
            ks, us = key[s], upd[s]
            ahead = (ks > kc) or (ks == kc and (us > uc or (us == uc and s < c)))
            r += ahead
        out[i] = r
        p = last[c] if last[c] >= 0 else 0
        if mode == 2: q = (i + p) >> 1
        elif mode == 3: q = p
        else: q = i
        key[c] = q; upd[c] = i; last[c] = i
    return bytes(out)
bad = 0
for spec in [("text", 3000, 1), ("mixed", 5000, 2), ("rand", 2000, 7), ("const", 500, 3), ("ramp", 600)]:
    d = vectors.make(spec)
    x = O.forward("BWT", d, len(d) + 64)[1]
    for data in (d, x):
        for name, mode in (("RANK", 2), ("TIMESTAMP", 3), ("MTFT", 1)):
            a = O.forward(name, data, len(data))[1]
            b = par_rank(data, mode)
            ok = a == b; bad += not ok
            if not ok:
                k = next(i for i in range(len(a)) if a[i] != b[i]); print(spec, name, "first diff", k, a[k], b[k])
print("bad", bad)
