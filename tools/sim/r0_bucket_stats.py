"""How many positions of an 8 MiB block sit in 2- and 3-symbol buckets too large for one workgroup's LDS (developer study, round 6:
the MSD first pass + in-LDS finish proposed for round 0 of the suffix sort leaves those on the fallback path).
usage: r0_bucket_stats.py            (mixed, text and three blocks of the real-file corpus)"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import knzlib
corpus = knzlib.corpus()
B = 8 << 20


def frac(name, d):
    a = np.frombuffer(d, dtype=np.uint8).astype(np.int64)
    dg = a[:-1] * 256 + a[1:]
    h = np.bincount(dg, minlength=65536)
    for cap in (8192, 32768):
        print("%-12s 2-symbol buckets: %6d non-empty, positions in buckets > %5d: %5.1f %%  (largest %d)" % (name, (h > 0).sum(), cap, 100.0 * h[h > cap].sum() / len(dg), h.max()))
    tr = dg[:-1] * 256 + a[2:]
    h3 = np.bincount(tr, minlength=1 << 24)
    print("%-12s 3-symbol buckets: %6d non-empty, positions in buckets > %5d: %5.1f %%  (largest %d)" % (name, (h3 > 0).sum(), 8192, 100.0 * h3[h3 > 8192].sum() / len(tr), h3.max()))


frac("mixed", corpus.mixed(B, 2))
frac("text", corpus.text(B, 1))
loc = corpus.local()[0]
for k in (0, 9, 20):
    frac("local blk%d" % k, loc[k * B:(k + 1) * B])
