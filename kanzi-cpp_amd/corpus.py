"""Deterministic synthetic corpora (integer-only) used when silesia.tar / enwik8 / enwik9 are absent.

Definitions follow SURVEY.md App. B (they were fixed there so that hashes of reference-kanzi
output could be recorded):
  text(N, seed)  -- Zipf-skewed 4096-word vocabulary over 26 letters, splitmix64 draws
  mixed(N, seed) -- 256 KiB segments cycling ramp / text / random / sparse zeros / runs
MD5(text(4194304,1)) = 533763267af795f681817771bd17d0cc, MD5(mixed(4194304,2)) = 4f3716bf4e8db9d931141d3c144dfc8c.
"""
import os

import numpy as np

_G = 0x9E3779B97F4A7C15
_M = (1 << 64) - 1
_LET = "etaoinshrdlucmfwypvbgkjqxz"
_SEG = 262144


def _sm(seed, count, start=1):
    """splitmix64 outputs number start..start+count-1 of the stream seeded with `seed`."""
    with np.errstate(over="ignore"):
        k = np.arange(start, start + count, dtype=np.uint64)
        x = np.uint64(seed & _M) + k * np.uint64(_G)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _vocab_table(seed):
    """8192 x 12 byte table: rows 0..4095 = word + ' ', rows 4096..8191 = word + '.\\n'; plus lengths."""
    s = (seed ^ 0x5EED) & _M
    draws = _sm(s, 4096 * 21 + 8)
    k = 0
    tab = np.zeros((8192, 12), dtype=np.uint8)
    lens = np.zeros(8192, dtype=np.int64)
    for w in range(4096):
        r = int(draws[k]); k += 1
        L = 2 + r % 9
        word = bytearray()
        for _ in range(L):
            r1 = int(draws[k]); r2 = int(draws[k + 1]); k += 2
            word.append(ord(_LET[(r1 % 26) * (r2 % 26) // 26]))
        a = bytes(word) + b" "
        b = bytes(word) + b".\n"
        tab[w, :len(a)] = np.frombuffer(a, dtype=np.uint8)
        tab[4096 + w, :len(b)] = np.frombuffer(b, dtype=np.uint8)
        lens[w] = len(a)
        lens[4096 + w] = len(b)
    return tab.reshape(-1), lens


def text(n, seed):
    tab, lens = _vocab_table(seed)
    out = np.empty(n + 16, dtype=np.uint8)
    pos = 0
    wi = 0  # words emitted so far
    step = 1 << 21
    while pos < n:
        a = _sm(seed, step, wi + 1)
        m = np.uint64(4095)
        idx = (((a & m) * ((a >> np.uint64(12)) & m) * ((a >> np.uint64(24)) & m)) >> np.uint64(24)).astype(np.int64)
        i = np.arange(wi, wi + step, dtype=np.int64)
        row = idx + 4096 * ((i % 16) == 15)
        ln = lens[row]
        ends = np.cumsum(ln)
        need = n - pos
        cnt = int(np.searchsorted(ends, need, side="left")) + 1
        cnt = min(cnt, step)
        row = row[:cnt]; ln = ln[:cnt]
        total = int(ends[cnt - 1])
        starts = ends[:cnt] - ln
        wid = np.repeat(np.arange(cnt, dtype=np.int64), ln)
        off = np.arange(total, dtype=np.int64) - np.repeat(starts, ln)
        chunk = tab[row[wid] * 12 + off]
        take = min(total, n - pos)
        out[pos:pos + take] = chunk[:take]
        pos += take
        wi += cnt
    return out[:n].tobytes()


def mixed(n, seed):
    S = _SEG
    out = np.empty(((n + S - 1) // S) * S, dtype=np.uint8)
    txt = np.frombuffer(text(n // 4 + 262144, seed + 1), dtype=np.uint8)
    tpos = 0
    k = 1
    nseg = (n + S - 1) // S
    for seg in range(nseg):
        m = seg % 5
        dst = out[seg * S:(seg + 1) * S]
        if m == 0:
            dst[:] = ((np.arange(S, dtype=np.int64) + seg) & 255).astype(np.uint8)
        elif m == 1:
            dst[:] = txt[tpos:tpos + S]
            tpos += S
        elif m == 2:
            a = _sm(seed, S // 8, k); k += S // 8
            dst[:] = np.frombuffer(a.astype("<u8").tobytes(), dtype=np.uint8)
        elif m == 3:
            dst[:] = 0
            a = _sm(seed, S // 64, k); k += S // 64
            ps = (a % np.uint64(S)).astype(np.int64)
            vs = ((a >> np.uint64(32)) & np.uint64(255)).astype(np.uint8)
            for p, v in zip(ps.tolist(), vs.tolist()):   # sequential: later pokes win
                dst[p] = v
        else:
            filled = 0
            while filled < S:
                a = _sm(seed, 4096, k)
                ln = (1 + ((a & np.uint64(255)) * ((a >> np.uint64(40)) & np.uint64(7))) // np.uint64(4)).astype(np.int64)
                ends = np.cumsum(ln)
                cnt = int(np.searchsorted(ends, S - filled, side="left")) + 1
                cnt = min(cnt, 4096)
                vals = ((a[:cnt] >> np.uint64(8)) & np.uint64(255)).astype(np.uint8)
                run = np.repeat(vals, ln[:cnt])
                take = min(len(run), S - filled)
                dst[filled:filled + take] = run[:take]
                filled += take
                k += cnt
    return out[:n].tobytes()


_REAL = {"silesia": "silesia.tar", "enwik9": "enwik9", "enwik8": "enwik8"}


def find_real(name):
    """Look for a real corpus file on this box ($KNZ_CORPUS_DIR, ~, /data, /datasets)."""
    fn = _REAL[name]
    dirs = [os.environ.get("KNZ_CORPUS_DIR"), os.path.expanduser("~"), "/data", "/datasets"]
    for d in dirs:
        if d and os.path.isfile(os.path.join(d, fn)):
            return os.path.join(d, fn)
    return None


def load(name, limit=None):
    """Returns (bytes, description). name in {silesia, enwik9, enwik8head}."""
    if name == "enwik8head":
        p = find_real("enwik8")
        if p:
            with open(p, "rb") as f:
                return f.read(4194304), "enwik8[:4MiB] (real)"
        return text(4194304, 1), "text(4194304,1) stand-in for enwik8 head"
    p = find_real(name)
    if p:
        with open(p, "rb") as f:
            d = f.read() if limit is None else f.read(limit)
        return d, "%s (real, %d B)" % (_REAL[name], len(d))
    if name == "silesia":
        n = 211957760 if limit is None else min(limit, 211957760)
        return mixed(n, 2), "mixed(%d,2) stand-in for silesia.tar" % n
    n = 1000000000 if limit is None else min(limit, 1000000000)
    return text(n, 1), "text(%d,1) stand-in for enwik9" % n
