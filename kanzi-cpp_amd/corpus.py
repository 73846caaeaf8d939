"""Deterministic synthetic corpora (integer-only) used when silesia.tar / enwik8 / enwik9 are absent.

Definitions follow SURVEY.md App. B (they were fixed there so that hashes of reference-kanzi
output could be recorded):
  text(N, seed)  -- Zipf-skewed 4096-word vocabulary over 26 letters, splitmix64 draws
  mixed(N, seed) -- 256 KiB segments cycling ramp / text / random / sparse zeros / runs
MD5(text(4194304,1)) = 533763267af795f681817771bd17d0cc, MD5(mixed(4194304,2)) = 4f3716bf4e8db9d931141d3c144dfc8c.

Round 4 adds inputs that are hard for a suffix sorter (long common prefixes; the shapes members of silesia.tar such as nci, xml
and mozilla have), used by the parity tests at full block size and by `bench.py --config 8`:
  repeats(N, seed)       -- text in which ~30 % of the bytes are copies of an earlier span (1-64 KiB, at most 4 MiB back, a few
                            single-byte edits per copy)
  tile(N, seed, period)  -- text(period, seed) repeated: with period = half a block every block is X || X
  periodic(N, seed, p)   -- a random unit of p bytes repeated (p = 3, 5, 7: periods no power of two divides)
  fibword(N)             -- the Fibonacci word over {a, b} (every prefix doubling round keeps groups alive)
  dna(N, seed)           -- random ACGT with ~25 % copied spans
  records(N, seed, r, k) -- rows of r bytes, each a copy of the row before with r - k bytes redrawn (a database table)
  gradient(N, seed, w)   -- rows of w samples of a slow ramp with two bits of noise (an image plane)
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


def _copy_spans(arr, seed, frac, min_len, max_len, back, alphabet):
    """Overwrites about frac * len(arr) bytes with copies of earlier spans (src within `back` bytes in front of the copy), then
    edits up to three bytes of each copy. Sequential: later copies see earlier ones."""
    n = len(arr)
    if n < 4 * min_len:
        return arr
    max_len = min(max_len, n // 4)
    want = int(frac * n)
    done = 0
    k = 1
    al = np.frombuffer(alphabet, dtype=np.uint8)
    while done < want:
        r = _sm(seed ^ 0xC0FFEE, 1024 * 6, k).reshape(1024, 6)
        k += 1024 * 6
        for r0, r1, r2, r3, r4, r5 in r.tolist():
            L = min_len + r0 % (max_len - min_len + 1)
            dst = L + r1 % (n - 2 * L + 1)
            gap = r2 % min(dst - L + 1, back)
            src = dst - L - gap
            arr[dst:dst + L] = arr[src:src + L]
            for e, rr in enumerate((r3, r4, r5)):
                if e < r3 % 4:
                    arr[dst + (rr >> 8) % L] = al[(rr >> 40) % len(al)]
            done += L
            if done >= want:
                break
    return arr


def repeats(n, seed):
    a = np.frombuffer(text(n, seed + 7), dtype=np.uint8).copy()
    return _copy_spans(a, seed, 0.30, 1024, 65536, 4 << 20, _LET.encode()).tobytes()


def tile(n, seed, period):
    unit = np.frombuffer(text(period, seed), dtype=np.uint8)
    return np.tile(unit, (n + period - 1) // period)[:n].tobytes()


def periodic(n, seed, p):
    unit = (_sm(seed ^ 0x9E71, p) % np.uint64(251)).astype(np.uint8)
    return np.tile(unit, (n + p - 1) // p)[:n].tobytes()


def fibword(n):
    a, b = b"a", b"b"              # s1 = a, s0 = b, s(k) = s(k-1) + s(k-2)
    while len(a) < n:
        a, b = a + b, a
    return a[:n]


def dna(n, seed):
    a = np.frombuffer(b"ACGT", dtype=np.uint8)[(_sm(seed ^ 0xD7A, n) >> np.uint64(33)) % np.uint64(4)].copy()
    return _copy_spans(a, seed + 1, 0.25, 256, 32768, 4 << 20, b"ACGT").tobytes()


def records(n, seed, rec, keep):
    """A table: rows of `rec` bytes, every row a copy of the one before with rec - keep of its bytes redrawn (which columns change
    is drawn per row) -- the fixed-length records of a database dump: a period that is no power of two, matches that break at
    different columns."""
    rows = (n + rec - 1) // rec
    r = _sm(seed ^ 0x7AB1E, rows * (rec - keep + 1) + rec)
    out = np.empty((rows, rec), dtype=np.uint8)
    cur = (r[:rec] % np.uint64(64) + np.uint64(32)).astype(np.uint8)
    k = rec
    for i in range(rows):
        d = r[k:k + rec - keep + 1]
        k += rec - keep + 1
        start = int(d[0] % np.uint64(rec))
        cols = (start + np.arange(rec - keep) * 7) % rec
        cur = cur.copy()
        cur[cols] = (d[1:] % np.uint64(64) + np.uint64(32)).astype(np.uint8)
        out[i] = cur
    return out.reshape(-1)[:n].tobytes()


def gradient(n, seed, width):
    """An image: rows of `width` samples of a slow ramp in x and y with two bits of noise (the smooth planes of a medical image or a
    scan: long near-matches one row back, at a distance that is no power of two)."""
    i = np.arange(n, dtype=np.int64)
    x, y = i % width, i // width
    noise = (_sm(seed ^ 0x6AAD, n) >> np.uint64(40)) & np.uint64(3)
    return (((x * 3 + y * 5) >> 2) + noise.astype(np.int64) & 255).astype(np.uint8).tobytes()


# Real bytes that exist on every box of this image (round 5): no corpus can be fetched, but ELF objects, C/C++ headers, Python
# sources and the mixed bag under /usr/share are real files with the zero padding, string tables, licence headers repeated a
# thousand times and near-duplicate functions the generators above lack. Sections in this order, each capped, files in sorted
# path order, regular files only (no symlinks), at most _LOCAL_FILE_CAP bytes from one file so that no single library dominates.
_LOCAL_SECTIONS = [
    ("/usr/lib/x86_64-linux-gnu", (".so",), 56 << 20),        # ELF shared objects (CPU code, symbol and string tables)
    ("/opt/rocm/include", None, 56 << 20),                     # C / C++ headers
    ("/usr/lib/python3.10", (".py",), 40 << 20),              # Python sources
    ("/usr/share", None, 48 << 20),                            # man pages (gzip), locale catalogues, icons, licences, terminfo, ...
    ("/usr/include", None, 48 << 20),
    ("/opt/rocm/lib", (".so",), 64 << 20),                    # ELF with embedded gfx code objects
]
_LOCAL_FILE_CAP = 8 << 20


def _local_files(root, suffixes):
    for d, dirs, files in os.walk(root):
        dirs.sort()
        for f in sorted(files):
            if suffixes is not None and not any(f.endswith(x) or (x + ".") in f for x in suffixes):
                continue
            p = os.path.join(d, f)
            if os.path.islink(p) or not os.path.isfile(p):
                continue
            yield p


def local(limit=211957760):
    """Deterministic concatenation of real files of this image, up to `limit` bytes. Returns (bytes, files used, description).
    The md5 is printed by bench.py (`config.input_md5`): two boxes of the same image give the same bytes."""
    parts, used, total = [], 0, 0
    for root, suffixes, cap in _LOCAL_SECTIONS:
        if total >= limit or not os.path.isdir(root):
            continue
        sec = 0
        for p in _local_files(root, suffixes):
            room = min(cap - sec, limit - total, _LOCAL_FILE_CAP)
            if room <= 0:
                break
            try:
                with open(p, "rb") as f:
                    b = f.read(room)
            except OSError:
                continue
            if not b:
                continue
            parts.append(b)
            sec += len(b); total += len(b); used += 1
    data = b"".join(parts)
    return data, used, "real files of this image (%d files: ELF .so + C/C++ headers + Python sources + /usr/share), %d B" % (used, len(data))


_REAL = {"silesia": "silesia.tar", "enwik9": "enwik9", "enwik8": "enwik8"}


def find_real(name):
    """Look for a real corpus file on this box ($KNZ_CORPUS_DIR, the working directory, ~, /root, /data, /datasets, /mnt, /workspace,
    each also with a corpus/ or corpora/ subdirectory)."""
    fn = _REAL[name]
    roots = [os.environ.get("KNZ_CORPUS_DIR"), os.getcwd(), os.path.expanduser("~"), "/root", "/data", "/datasets", "/mnt", "/workspace", "/tmp"]
    dirs = []
    for r in roots:
        if r:
            dirs += [r, os.path.join(r, "corpus"), os.path.join(r, "corpora"), os.path.join(r, "silesia")]
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
    if name == "local":                # real files of this image (bench.py --config 9)
        d, _, desc = local(211957760 if limit is None else limit)
        return d, desc
    if name == "repeats":              # (no real counterpart: the long-common-prefix stand-in, bench.py --config 8)
        n = 211957760 if limit is None else min(limit, 211957760)
        return repeats(n, 3), "repeats(%d,3) stand-in for long-common-prefix data (text with 30%% copied spans of 1-64 KiB)" % n
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
