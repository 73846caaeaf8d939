import numpy as np, sys
def groups(lcp, h):
    # boundaries: slot i starts a group if lcp[i] < h
    n = len(lcp)
    st = np.flatnonzero(lcp < h)
    sizes = np.diff(np.append(st, n))
    return st, sizes
for name in sys.argv[1:]:
    lcp = np.fromfile(name + '.lcp', dtype=np.int32)
    n = len(lcp)
    print(name, 'n', n, 'mean lcp %.1f' % lcp.mean(), 'max', lcp.max())
    h = 12
    while True:
        st, sz = groups(lcp, h)
        tied = sz[sz > 1]
        if len(tied) == 0: break
        tot = tied.sum()
        b = [int(tied[tied == 2].sum()), int(tied[(tied > 2) & (tied <= 4)].sum()), int(tied[(tied > 4) & (tied <= 8)].sum()),
             int(tied[(tied > 8) & (tied <= 32)].sum()), int(tied[(tied > 32) & (tied <= 256)].sum()), int(tied[tied > 256].sum())]
        print('  h=%7d tied members %8d groups %8d | in pairs %8d, 3-4 %8d, 5-8 %8d, 9-32 %8d, 33-256 %8d, >256 %8d' % (h, tot, len(tied), *b))
        h *= 2
