"""Study for the LZ decoder plan (DESIGN_HISTORY.md section 7): how deep is the dependency graph of the match copies?
Level of a match = 1 + the highest level among the matches that wrote its source bytes (CPU only, uses the oracle)."""
import sys, bisect
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "..", "tests"))
import knzlib, vectors
O = knzlib.Oracle()
def study(name, d):
    ok, s = O.forward("LZX", d, len(d) + len(d) // 64 + 64)
    if ok != 1: print(name, "not compressible"); return
    litEnd = int.from_bytes(s[0:4], "little"); nTok = int.from_bytes(s[4:8], "little"); nDist = int.from_bytes(s[8:12], "little")
    mm = ((s[12] >> 1) & 7) + 2
    t = litEnd; m = litEnd + nTok; l = m + nDist; sp = 13; dpos = 0; rep0 = rep1 = len(s)
    def rdlen(p):
        b = s[p]
        if b < 254: return b, p + 1
        if b == 254: return 254 + ((s[p+1] << 8) | s[p+2]), p + 3
        return 255 + ((s[p+1] << 16) | (s[p+2] << 8) | s[p+3]), p + 4
    starts = []; levels = []   # match output ranges [start, end) with level
    ends = []
    import array
    maxlvl = 0; hist = {}
    while True:
        tok = s[t]; t += 1
        if (tok & 0x18) == 0:
            ml = tok & 3
            if ml == 3: e, l = rdlen(l); ml = 3 + mm + e
            else: ml += mm
            dist = rep1 if tok & 4 else rep0
        else:
            ml = tok & 7
            if ml == 7: e, l = rdlen(l); ml = 7 + mm + e
            else: ml += mm
            nb = (tok >> 3) & 3; dist = 0
            for _ in range(nb): dist = (dist << 8) | s[m]; m += 1
        if tok >= 32:
            if tok >= 0xE0: e, sp = rdlen(sp); lit = 7 + e
            else: lit = tok >> 5
            sp += lit; dpos += lit
            if sp >= litEnd - 13: break
        rep1 = rep0; rep0 = dist
        ref = dpos - dist; need_end = ref + min(ml, dist)
        # level = 1 + max level of matches overlapping [ref, need_end)
        i = bisect.bisect_right(starts, ref) - 1
        lv = 0
        if i >= 0 and ends[i] <= ref: i += 1
        i = max(i, 0)
        while i < len(starts) and starts[i] < need_end:
            if ends[i] > ref: lv = max(lv, levels[i])
            i += 1
        lv += 1
        starts.append(dpos); ends.append(dpos + ml); levels.append(lv)
        hist[lv] = hist.get(lv, 0) + 1
        dpos += ml
    ntok = len(levels); mx = max(levels)
    print("%-10s n=%d tokens=%d levels=%d  tokens/level=%.1f  (tokens at level<=8: %.1f%%)" % (name, len(d), ntok, mx, ntok / mx, 100.0 * sum(v for k, v in hist.items() if k <= 8) / ntok))
study("text4m", vectors.make(("text", 4 << 20, 1)))
study("mixed4m", vectors.make(("mixed", 4 << 20, 2)))
study("runs", vectors.make(("runs", 9000, 300)))
