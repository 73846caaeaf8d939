import numpy as np, sys
def chain_round(sa, lcp, h0, kmax):
    n = len(sa)
    start = lcp < h0
    gid = np.maximum.accumulate(np.where(start, np.arange(n), 0))
    st = np.flatnonzero(start); sz = np.diff(np.append(st, n))
    szslot = np.repeat(sz, sz)
    rank = np.empty(n, dtype=np.int64); rank[sa] = np.arange(n)
    G = np.empty(n + 1, dtype=np.int64); G[sa] = gid; G[n] = -1
    tied = (szslot > 1) & (szslot <= kmax)
    slots = np.flatnonzero(tied)
    pos = sa[slots].astype(np.int64); g = gid[slots]
    o = np.lexsort((pos, g)); pos = pos[o]; g = g[o]
    same = g[1:] == g[:-1]
    p = pos[:-1][same]; q = pos[1:][same]; gp = g[:-1][same]
    d = q - p
    D = np.zeros(n + 2, dtype=np.int64); D[p] = d
    brk = np.flatnonzero(D[1:] != D[:-1])          # x such that D[x+1] != D[x]
    e = brk[np.searchsorted(brk, p, side='left')]  # first break position >= p
    s = e + 1 - p
    a = e + 1; b = e + 1 + d
    res = (b >= n) | (G[np.minimum(a, n)] != G[np.minimum(b, n)])
    npairs = len(p)
    # groups fully resolved
    unres_per_group = np.bincount(np.searchsorted(st, gp), weights=(~res).astype(np.float64), minlength=len(st))
    grp_tied = (sz > 1) & (sz <= kmax)
    full = grp_tied & (unres_per_group == 0)
    return dict(members=int(sz[grp_tied].sum()), pairs=npairs, resolved=int(res.sum()), members_full=int(sz[full].sum()),
                pair_groups=int((sz == 2).sum()), pair_groups_res=int(((sz == 2) & (unres_per_group == 0)).sum()),
                mean_s=float(s.mean()), runs=int(len(np.unique(e))))
for name in sys.argv[1:]:
    sa = np.fromfile(name + '.sa', dtype=np.int32).astype(np.int64); lcp = np.fromfile(name + '.lcp', dtype=np.int32)
    print(name)
    for h0 in (16, 32, 64, 128, 256, 1024):
        for kmax in (2, 8, 256):
            r = chain_round(sa, lcp, h0, kmax)
            print('  h0=%5d kmax=%3d members %8d pairs %8d resolved %8d (%.1f%%)  members in fully resolved groups %8d (%.1f%%)  pair groups %7d resolved %7d  mean chain %.0f  distinct chain ends %d' % (
                h0, kmax, r['members'], r['pairs'], r['resolved'], 100.0 * r['resolved'] / max(1, r['pairs']), r['members_full'], 100.0 * r['members_full'] / max(1, r['members']),
                r['pair_groups'], r['pair_groups_res'], r['mean_s'], r['runs']))
