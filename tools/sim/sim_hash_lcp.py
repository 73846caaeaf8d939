import numpy as np, sys
def phase_cost(g):
    # g = true group LCP minus known c (>=0), vectorised. returns (reads per member, offset of the final T-step relative to c)
    g = g.astype(np.int64)
    cost = np.ones_like(g)            # first T-step at c
    off = np.zeros_like(g)
    tie = g >= 8
    # after the tie: c += 8
    r = np.where(tie, g - 8, 0)
    # gallop: s = 8, 16, ...; succeed while r >= s (cumulative): after k successes the shared extension is 8*(2^k - 1)
    # find k = max such that 8*(2^k - 1) <= r  -> 2^k <= r/8 + 1
    k = np.floor(np.log2(r // 8 + 1)).astype(np.int64)
    ext = 8 * ((1 << k) - 1)
    # k successes + 1 failure, then descent from s_fail/2 = 8*2^(k-1) down to 8: k steps (for k >= 1; the test at size 8 ... ) -> k-? 
    desc = np.maximum(k, 0)           # sizes 8*2^(k-1) ... 8 : k steps
    rem = r - ext                     # remaining after gallop, < 8*2^k
    # descent lands on the largest multiple of 8 <= rem
    land = (rem // 8) * 8
    cost = cost + np.where(tie, k + 1 + desc + 1, 0)
    off = np.where(tie, 8 + ext + land, 0)
    return cost, off
def simulate(lcp, h0, kmax, kmin=2):
    n = len(lcp)
    st = np.flatnonzero(lcp < h0)
    sz = np.diff(np.append(st, n))
    sel = (sz >= kmin) & (sz <= kmax)
    a = st[sel]; b = a + sz[sel]
    c = np.full(len(a), h0, dtype=np.int64)
    total_reads = 0; members0 = int(sz[sel].sum()); phases = 0; maxphase = 0
    lcpx = np.append(lcp, 0).astype(np.int64)
    while len(a):
        phases += 1
        idx = np.empty(2 * len(a), dtype=np.int64); idx[0::2] = a + 1; idx[1::2] = b
        m = np.minimum.reduceat(lcpx, idx)[0::2]
        cost, off = phase_cost(m - c)
        size = b - a
        total_reads += int((cost * size).sum())
        cnew = c + off + 8
        # split
        rep = np.repeat(np.arange(len(a)), size)
        slots = np.concatenate([np.arange(x, y) for x, y in zip(a[:0], b[:0])]) if False else (np.repeat(a, size) + (np.arange(size.sum()) - np.repeat(np.cumsum(size) - size, size)))
        isb = (lcpx[slots] < cnew[rep]) | (slots == a[rep])
        bpos = slots[isb]; brep = rep[isb]
        # ends: next boundary in the same group or group end
        nxt = np.append(bpos[1:], 0); same = np.append(brep[1:] == brep[:-1], False)
        e = np.where(same, nxt, b[brep])
        s2 = e - bpos
        keep = s2 > 1
        a = bpos[keep]; b = e[keep]; c = cnew[brep[keep]]
    return members0, total_reads, phases
def doubling_cost(lcp, h0, kmax, kmin=2):
    # member visits in rounds h0, 2h0, ... for the members that are in groups of size kmin..kmax at depth h0 (they only get smaller)
    n = len(lcp)
    st = np.flatnonzero(lcp < h0)
    sz = np.diff(np.append(st, n))
    sel = (sz >= kmin) & (sz <= kmax)
    mask = np.repeat(sel, sz)
    visits = 0; h = h0; rounds = 0
    while True:
        st = np.flatnonzero(lcp < h); s = np.diff(np.append(st, n))
        tied = np.repeat(s > 1, s) & mask
        v = int(tied.sum())
        if v == 0: break
        visits += v; rounds += 1; h *= 2
    return visits, rounds
for name in sys.argv[1:]:
    lcp = np.fromfile(name + '.lcp', dtype=np.int32)
    print(name)
    for h0 in (12, 16, 32, 64, 128):
        for kmax in (2, 4, 8, 32, 256):
            m, reads, ph = simulate(lcp, h0, kmax)
            v, r = doubling_cost(lcp, h0, kmax)
            print('  h0=%4d kmax=%3d members %8d  hash reads %9d (%.1f/member, %d phases)   doubling visits %9d (%d rounds)  ratio visits*6/reads %.2f' % (h0, kmax, m, reads, reads / max(m, 1), ph, v, r, 6 * v / max(reads, 1)))
