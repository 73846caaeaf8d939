// LZ / LZX (Lempel-Ziv with repeat distances) on gfx950: one wave per block.
//
// Reference being replaced: transform/LZCodec.cpp:119-456 (LZXCodec<T>::forward), :470-640 (inverseV6),
// LZCodec.hpp:187-246 (hash, emitLength, readLength, findMatch), constants LZCodec.cpp:66-114.
// "LZ" = 16-bit hash, one look-ahead position; "LZX" = 19-bit hash, two look-ahead positions.
//
// The encoder is a greedy parse whose every decision depends on the hash table left by all earlier decisions
// and on the last two distances, so the parse itself is one dependent chain per block (like FPAQ).  The wave
// walks that chain with wave-uniform control flow and spends its 64 lanes on everything under a decision:
//   * match lengths: 64 x 8-byte compares per step (the reference's whole-word findMatch, 512 bytes at a time);
//   * backward extension: 64 byte compares per step;
//   * the hash-table fill behind a match: 64 positions per step.  Positions only ever grow, so "last writer
//     wins" is an atomic max and the table ends up exactly as the serial loop leaves it;
//   * literal runs and the final section copies: 8 bytes per lane.
// The table is only touched with device-scope atomics (served by L2), so the max updates and the lookups agree.
// The decoder resolves a token with uniform loads and copies literals / matches with the whole wave; an
// overlapping match (distance < length) is a modulo gather from the bytes already written.
#include "common.hpp"
#include "stages.hpp"

namespace knz {

constexpr int LZ_MAXD1 = (1 << 16) - 2;
constexpr int LZ_MAXD2 = (1 << 24) - 2;
constexpr int LZ_MM = 4;                                  // data type hint is always "undefined" on this path
constexpr int LZ_MAXMATCH = 65535 + 254 + 4;
constexpr int LZ_MINBLOCK = 24;

__host__ __device__ inline int lz_max_encoded(int n) { return ((n <= 1024) ? n + 16 : n + n / 64) + 2; }

__device__ __forceinline__ u64 ld64u(const u8* p) { u64 v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ u32 ld32u(const u8* p) { u32 v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ void st64u(u8* p, u64 v) { __builtin_memcpy(p, &v, 8); }
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ u64 sgpr64(u64 v)
{
    return ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)v);
}

template <int HASH_LOG>
__device__ __forceinline__ u32 lz_hash(u64 w) { return (u32)(((w << 24) * 0x1E35A7BDull) >> (64 - HASH_LOG)); }

// The table belongs to one wave, so plain loads and stores are coherent (same CU, same L1) and no access has to
// travel to the device-wide coherence point. Positions only grow, so a store is the reference's overwrite.
__device__ __forceinline__ int tab_get(const int* t, u32 h) { return t[h]; }
__device__ __forceinline__ void tab_put(int* t, u32 h, int p) { t[h] = p; }

// 64 Kbit LDS filter over the low 16 hash bits: returns true when some lane of `act` may share its hash with
// another one (never misses a real duplicate); leaves the filter clear
__device__ __forceinline__ bool lz_maybe_dups(u32* seen, u32 h, bool act)
{
    const u32 fbit = 1u << (h & 31), fidx = (h >> 5) & 2047;
    const u32 old = act ? atomicOr(&seen[fidx], fbit) : 0u;
    if (act) atomicAnd(&seen[fidx], ~fbit);
    return __ballot(act && (old & fbit)) != 0;
}

// table[h] = p for every lane of `act`, in lane order: with equal hashes the highest lane (latest position) stays
__device__ __forceinline__ void tab_put_wave(int* t, u32* seen, u32 h, int p, bool act, int lane)
{
    bool lose = false;
    if (lz_maybe_dups(seen, h, act)) {
#pragma unroll 9
        for (int j = 1; j < 64; j++) {
            const u32 hj = (u32)__builtin_amdgcn_readlane((int)h, j);
            const int aj = __builtin_amdgcn_readlane(act ? 1 : 0, j);
            if (aj && j > lane && hj == h) lose = true;
        }
    }
    if (act && !lose) t[h] = p;
}

// LZCodec.hpp:227-246 with the whole wave: compares whole 8-byte words only (so the result can stop up to 7 bytes
// short of `limit`), lane l looks at word n/8 + l
__device__ int lz_match(const u8* s, int a, int b, int limit, int lane)
{
    int n = 0;
    for (;;) {
        const int o = n + 8 * lane;
        const bool valid = o + 8 <= limit;
        u64 x = 0;
        if (valid) x = ld64u(s + a + o) ^ ld64u(s + b + o);
        const u64 stop = __ballot(!valid || x != 0);
        if (stop) {
            const int f = __ffsll((long long)stop) - 1;
            const u32 xl = (u32)__builtin_amdgcn_readlane((int)(u32)x, f);
            const u32 xh = (u32)__builtin_amdgcn_readlane((int)(u32)(x >> 32), f);
            const int vf = __builtin_amdgcn_readlane(valid ? 1 : 0, f);
            if (!vf) return n + 8 * f;
            return n + 8 * f + (xl ? (__ffs((int)xl) - 1) >> 3 : 4 + ((__ffs((int)xh) - 1) >> 3));
        }
        n += 512;
    }
}

// non-overlapping copy: 8 bytes per lane for the bulk, one byte per lane for the last < 8 bytes, so that even a
// 5-byte copy is one load and one store deep (a byte loop in one lane would be a chain of load -> store pairs)
__device__ __forceinline__ void wave_copy(u8* dst, const u8* src, int len, int lane)
{
    const int bulk = len & ~7;
    for (int i = 8 * lane; i < bulk; i += 512) st64u(dst + i, ld64u(src + i));
    if (lane < len - bulk) dst[bulk + lane] = src[bulk + lane];
}

// LZCodec.hpp:192-210 (the reference's 3-byte form also stores a zero byte that the next write covers)
__device__ __forceinline__ int lz_put_len(u8* p, int len, int lane)
{
    if (len < 254) { if (lane == 0) p[0] = (u8)len; return 1; }
    if (len < 65536 + 254) {
        const int v = len - 254;
        if (lane == 0) { p[0] = 0xFE; p[1] = (u8)(v >> 8); p[2] = (u8)v; }
        return 3;
    }
    const u32 v = (u32)(len - 255);
    if (lane == 0) { p[0] = 0xFF; p[1] = (u8)(v >> 16); p[2] = (u8)(v >> 8); p[3] = (u8)v; }
    return 4;
}

struct LzScratch { int* tables; u8* side; size_t secStride; };

__device__ __forceinline__ u64 readlane64(u64 v, int l)
{
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (u32)__builtin_amdgcn_readlane((int)v, l);
}

// The positions the reference visits while it finds nothing (a literal run) do not depend on one another except
// through the hash table, so up to 64 of them are evaluated at once: lane i takes the i-th position the serial loop
// would visit, looks up its bucket, and tests the three candidates (bucket, two repeat distances) for the 4-byte
// equality that precedes every findMatch.  A lane whose bucket was filled by an earlier lane of the same batch
// takes that lane's position instead (found exactly; a 64 Kbit LDS filter says when the search is needed at all).
// Lanes before the first one with an equality are misses for sure: their buckets are written and the serial state
// (position, skip counter) jumps over them.  The first lane with an equality is then handled by the serial code
// below, with the candidates already known for it and for the next two positions.
template <int HASH_LOG, bool EXTRA>
__global__ __launch_bounds__(64) void k_lz_forward(XfStage st, LzScratch ws)
{
    const int b = blockIdx.x;
    const int lane = lane_id();
    const int n = (int)st.len[b];
    if (n == 0) return;
    const u8* __restrict__ src = st.src[b];
    u8* __restrict__ dst = st.dst[b];
    int ok = 0, outLen = 0;
    int* table = ws.tables + ((size_t)b << HASH_LOG);
    u8* tk = ws.side + (size_t)b * 3 * ws.secStride;
    u8* mb = tk + ws.secStride;
    u8* ml = mb + ws.secStride;
    __shared__ u32 seen[2048];
    for (int i = lane; i < 2048; i += 64) seen[i] = 0;
    __syncthreads();
    if (st.cap[b] >= (u32)lz_max_encoded(n) && n >= LZ_MINBLOCK) {
        const int srcEnd = n - 16 - 2;
        const int maxDist = (srcEnd < 4 * LZ_MAXD1) ? LZ_MAXD1 : LZ_MAXD2;
        int pos = 0, d = 13, anchor = 0, nm = 0, nl = 0, nt = 0;
        int rep0 = n, rep1 = n, recent = 0, skip = 0;
        bool reject = false;
        while (pos < srcEnd) {
            // ---- the next (up to 64) positions of a literal run, all at once
            const int stride = 1 + (skip >> 6);
            const int cnt = 64 - (skip & 63);                 // visits until the stride of the reference changes
            const int bp = pos + lane * stride;
            const bool act = (lane < cnt) && (bp < srcEnd);
            const u64 bw = act ? ld64u(src + bp) : 0ull;
            const u32 bh = lz_hash<HASH_LOG>(bw);
            int cd = act ? tab_get(table, bh) : 0;
            const bool dups = lz_maybe_dups(seen, bh, act);
            if (dups) {
#pragma unroll 9
                for (int j = 0; j < 63; j++) {
                    const u32 hj = (u32)__builtin_amdgcn_readlane((int)bh, j);
                    const int aj = __builtin_amdgcn_readlane(act ? 1 : 0, j);
                    if (aj && lane > j && hj == bh) cd = pos + j * stride;
                }
            }
            bool hit = false;
            u32 cw = 0, xw = 0, yw = 0;
            if (act) {
                const int blo = (bp - maxDist > 0) ? bp - maxDist : 0;
                const int rx = bp + 1 - rep0, ry = bp + 1 - rep1;
                cw = ld32u(src + (cd > blo ? cd : 0));
                xw = ld32u(src + (rx > blo ? rx : 0));
                yw = ld32u(src + (ry > blo ? ry : 0));
                const u32 n4 = (u32)(bw >> 8);
                hit = (cd > blo && cw == (u32)bw) || (rx > blo && xw == n4) || (ry > blo && yw == n4);
            }
            const u64 hitMask = __ballot(hit);
            const u64 endMask = hitMask | __ballot(!act);
            const int f = endMask ? __ffsll((long long)endMask) - 1 : 64;       // lanes below f found nothing
            if (!dups) { if (lane < f) tab_put(table, bh, bp); }
            else {
                bool lose = false;
#pragma unroll 9
                for (int j = 1; j < 64; j++) {
                    const u32 hj = (u32)__builtin_amdgcn_readlane((int)bh, j);
                    if (j < f && j > lane && hj == bh) lose = true;
                }
                if (lane < f && !lose) tab_put(table, bh, bp);
            }
            if (f > 0) { skip += f; recent = 0; pos += f * stride; }
            if (f == 64 || !((hitMask >> f) & 1)) continue;

            // ---- the serial step of the reference at `pos`
            const u64 w0 = readlane64(bw, f);
            const int cand = __builtin_amdgcn_readlane(cd, f);
            const bool near1 = (stride == 1) && (f + 1 < 64) && ((__ballot(act) >> ((f + 1) & 63)) & 1);
            const bool near2 = (stride == 1) && (f + 2 < 64) && ((__ballot(act) >> ((f + 2) & 63)) & 1);
            const int bc1 = __builtin_amdgcn_readlane(cd, (f + 1) & 63);
            const int bc2 = __builtin_amdgcn_readlane(cd, (f + 2) & 63);
            const u64 bw1 = readlane64(bw, (f + 1) & 63), bw2 = readlane64(bw, (f + 2) & 63);
            if (lane == 0) tab_put(table, lz_hash<HASH_LOG>(w0), pos);
            const int nxt = pos + 1;
            const int lo = (pos - maxDist > 0) ? pos - maxDist : 0;
            const u32 nx4 = (u32)(w0 >> 8);
            const int refA = nxt - (recent ? rep1 : rep0);
            const int refB = nxt - (recent ? rep0 : rep1);
            // the candidate words were fetched by the batch (same clamped addresses)
            const u32 x4 = (u32)__builtin_amdgcn_readlane((int)xw, f), y4 = (u32)__builtin_amdgcn_readlane((int)yw, f);
            const u32 a4 = recent ? y4 : x4;
            const u32 b4 = recent ? x4 : y4;
            const u32 c4 = (u32)__builtin_amdgcn_readlane((int)cw, f);
            int best = 0, ref = refA;
            if (refA > lo && a4 == nx4) best = lz_match(src, nxt, refA, min(srcEnd - nxt, LZ_MAXMATCH), lane);
            else {
                ref = refB;
                if (refB > lo && b4 == nx4) best = lz_match(src, nxt, refB, min(srcEnd - nxt, LZ_MAXMATCH), lane);
            }
            if (best < LZ_MM) {
                ref = cand;
                if (cand > lo && c4 == (u32)w0) best = lz_match(src, pos, cand, min(srcEnd - pos, LZ_MAXMATCH), lane);
                if (best < LZ_MM) { pos = nxt + (skip >> 6); skip++; recent = 0; continue; }
                if (pos - ref != rep0 && pos - ref != rep1) {
                    // new distance: is the match one (LZX: two) position(s) further at least as long?
                    const int p1 = nxt, p2 = nxt + 1;
                    const u32 h1 = lz_hash<HASH_LOG>(near1 ? bw1 : sgpr64(ld64u(src + p1)));
                    const int c1 = near1 ? bc1 : sgpr(tab_get(table, h1));
                    if (lane == 0) tab_put(table, h1, p1);
                    int c2 = 0;
                    if (EXTRA) {
                        const u32 h2 = lz_hash<HASH_LOG>(near2 ? bw2 : sgpr64(ld64u(src + p2)));
                        c2 = near2 ? bc2 : sgpr(tab_get(table, h2));      // p1 is in the table, as in the serial order
                        if (lane == 0) tab_put(table, h2, p2);
                    }
                    if (c1 > lo + 1 && sgpr((int)ld32u(src + p1 + best - 3)) == sgpr((int)ld32u(src + c1 + best - 3))) {
                        const int b1 = lz_match(src, p1, c1, min(srcEnd - p1, LZ_MAXMATCH), lane);
                        if (b1 >= best) { ref = c1; best = b1; pos = p1; }
                    }
                    if (EXTRA) {
                        if (c2 > lo + 2 && sgpr((int)ld32u(src + p2 + best - 3)) == sgpr((int)ld32u(src + c2 + best - 3))) {
                            const int b2 = lz_match(src, p2, c2, min(srcEnd - p2, LZ_MAXMATCH), lane);
                            if (b2 >= best) { ref = c2; best = b2; pos = p2; }
                        }
                    }
                }
                // extend backwards, 64 bytes per step
                for (;;) {
                    const bool c = (pos - lane > anchor) && (ref - lane > lo) && (src[pos - 1 - lane] == src[ref - 1 - lane]);
                    const u64 fail = __ballot(!c);
                    const int k = fail ? __ffsll((long long)fail) - 1 : 64;
                    best += k; ref -= k; pos -= k;
                    if (k < 64) break;
                }
                if (best > LZ_MAXMATCH) { ref += best - LZ_MAXMATCH; pos += best - LZ_MAXMATCH; best = LZ_MAXMATCH; }
            } else {
                // repeat match at pos + 1: take the byte at pos with it when it matches too
                if (best >= LZ_MAXMATCH || sgpr((int)src[pos]) != sgpr((int)src[ref - 1])) {
                    pos++;
                    if (lane == 0) tab_put(table, lz_hash<HASH_LOG>(ld64u(src + pos)), pos);
                } else { best++; ref--; }
            }
            skip = 0;
            const int dist = pos - ref;
            int token, th;
            if (dist == rep0) { token = 0x00; th = 3; }
            else if (dist == rep1) { token = 0x04; th = 3; }
            else {
                const int w3 = dist >= 65536, w2 = dist >= 256;
                if (lane == 0) {
                    int q = nm;
                    if (w3) mb[q++] = (u8)(dist >> 16);
                    if (w2) mb[q++] = (u8)(dist >> 8);
                    mb[q] = (u8)dist;
                }
                nm += w3 + w2 + 1;
                token = (w3 + w2 + 1) << 3; th = 7;
            }
            const int mlen = best - LZ_MM;
            if (mlen >= th) { token += th; nl += lz_put_len(ml + nl, mlen - th, lane); }
            else token += mlen;
            rep1 = rep0; rep0 = dist; recent = 1;
            const int lit = pos - anchor;
            if (lit == 0) { if (lane == 0) tk[nt] = (u8)token; nt++; }
            else {
                if (lit >= 7) {
                    if (lit >= (1 << 24)) { reject = true; break; }
                    if (lane == 0) tk[nt] = (u8)((7 << 5) | token);
                    nt++;
                    d += lz_put_len(dst + d, lit - 7, lane);
                } else { if (lane == 0) tk[nt] = (u8)((lit << 5) | token); nt++; }
                wave_copy(dst + d, src + anchor, lit, lane);
                d += lit;
            }
            // the sections only grow: once they cannot fit, the final size test of the reference fails as well
            if (d + nt + nm + nl >= n) { reject = true; break; }
            anchor = pos + best;
            for (int p0 = pos + 1; p0 < anchor; p0 += 64) {
                const int p = p0 + lane;
                const bool in = p < anchor;
                tab_put_wave(table, seen, lz_hash<HASH_LOG>(in ? ld64u(src + p) : 0ull), p, in, lane);
            }
            pos = anchor;
        }
        const int lit = n - anchor;
        if (!reject && d + lit + nt + nm + nl < n) {
            if (lit >= 7) { if (lane == 0) tk[nt] = (u8)(7 << 5); nt++; d += lz_put_len(dst + d, lit - 7, lane); }
            else { if (lane == 0) tk[nt] = (u8)(lit << 5); nt++; }
            wave_copy(dst + d, src + anchor, lit, lane);
            d += lit;
            if (lane == 0) {
                const u32 hd[3] = { (u32)d, (u32)nt, (u32)nm };
                for (int k = 0; k < 3; k++) for (int j = 0; j < 4; j++) dst[4 * k + j] = (u8)(hd[k] >> (8 * j));
                dst[12] = (u8)((maxDist == LZ_MAXD1 ? 0 : 1) | (((LZ_MM - 2) & 7) << 1));
            }
            __threadfence_block();                     // the sections were written by lane 0
            wave_copy(dst + d, tk, nt, lane); d += nt;
            wave_copy(dst + d, mb, nm, lane); d += nm;
            wave_copy(dst + d, ml, nl, lane); d += nl;
            outLen = d;
            ok = (d <= n - n / 100) ? 1 : 0;
        }
    }
    if (lane == 0) { st.ok[b] = (u8)ok; st.newLen[b] = (u32)outLen; }
}

// A 256-byte window over one of the four byte sequences the decoder walks (tokens, distances, length extensions,
// literal-run extensions), held 4 bytes per lane: the token chain reads it with readlane, so no load sits on it.
// Bytes past the block read as zero (the reference relies on two bytes of padding, LZCodec.cpp:486-490).
struct LzByteWin {
    u32 reg;
    int base;
    __device__ __forceinline__ void refill(const u8* s, int from, int limit, int lane)
    {
        base = from;
        const int o = from + 4 * lane;
        if (o + 4 <= limit) reg = ld32u(s + o);
        else {
            u32 v = 0;
            for (int k = 0; k < 4; k++) if (o + k < limit) v |= (u32)s[o + k] << (8 * k);
            reg = v;
        }
    }
    __device__ __forceinline__ u32 at(const u8* s, int x, int limit, int lane)
    {
        if (x < base || x >= base + 256) refill(s, x, limit, lane);
        const int o = x - base;
        const u32 wv = (u32)__builtin_amdgcn_readlane((int)reg, sgpr(o >> 2));
        return (wv >> (8 * (o & 3))) & 0xFFu;
    }
};

// LZCodec.hpp:212-225
__device__ __forceinline__ u32 lz_get_len(LzByteWin& w, const u8* s, int& pos, int limit, int lane)
{
    const u32 b0 = w.at(s, pos, limit, lane);
    if (b0 < 254) { pos += 1; return b0; }
    const u32 b1 = w.at(s, pos + 1, limit, lane), b2 = w.at(s, pos + 2, limit, lane);
    if (b0 == 254) { pos += 3; return 254 + ((b1 << 8) | b2); }
    const u32 b3 = w.at(s, pos + 3, limit, lane);
    pos += 4;
    return 255 + ((b1 << 16) | (b2 << 8) | b3);
}

__global__ __launch_bounds__(64) void k_lz_inverse(XfStage st)
{
    const int b = blockIdx.x;
    const int lane = lane_id();
    const int n = (int)st.len[b];
    if (n == 0) return;
    const u8* __restrict__ src = st.src[b];
    u8* dst = st.dst[b];
    const int cap = (st.cap[b] > 0x7FFFFFFFu) ? 0x7FFFFFFF : (int)st.cap[b];
    int ok = 0, d = 0;
    do {
        if (n < 13) break;
        const int litEnd = sgpr((int)ld32u(src)), nTok = sgpr((int)ld32u(src + 4)), nDist = sgpr((int)ld32u(src + 8));
        if (litEnd < 0 || nTok < 0 || nDist < 0) break;
        if (litEnd < 13 || litEnd > n || nTok > n - litEnd || nDist > n - litEnd - nTok) break;
        int t = litEnd, m = litEnd + nTok, l = m + nDist;
        const int flags = sgpr((int)src[12]);
        const int maxDist = (flags & 1) ? LZ_MAXD2 : LZ_MAXD1;
        const int mm = ((flags >> 1) & 7) + 2;
        int s = 13, rep0 = n, rep1 = n;
        int settled = 0;                         // output bytes below this index are known to have left the wave
        LzByteWin wt, wm, wl, ws;
        wt.refill(src, t, n, lane); wm.refill(src, m, n, lane); wl.refill(src, l, n, lane); ws.refill(src, s, n, lane);
        ok = 1;
        for (;;) {
            const int token = (int)wt.at(src, t, n, lane);
            t++;
            int mlen, dist;
            if ((token & 0x18) == 0) {
                mlen = token & 3;
                mlen = (mlen == 3) ? 3 + mm + (int)lz_get_len(wl, src, l, n, lane) : mlen + mm;
                dist = (token & 4) ? rep1 : rep0;
            } else {
                mlen = token & 7;
                mlen = (mlen == 7) ? 7 + mm + (int)lz_get_len(wl, src, l, n, lane) : mlen + mm;
                const int nb = (token >> 3) & 3;          // 1, 2 or 3 distance bytes, most significant first
                dist = (int)wm.at(src, m, n, lane);
                if (nb >= 2) dist = (dist << 8) | (int)wm.at(src, m + 1, n, lane);
                if (nb == 3) dist = (dist << 8) | (int)wm.at(src, m + 2, n, lane);
                m += nb;
            }
            if (token >= 32) {
                const u32 lit = (token >= 0xE0) ? 7u + lz_get_len(ws, src, s, n, lane) : (u32)(token >> 5);
                if (lit > (u32)(cap - d) || lit > (u32)(litEnd - s)) { ok = 0; break; }
                wave_copy(dst + d, src + s, (int)lit, lane);
                s += (int)lit; d += (int)lit;
                if (s >= litEnd - 13) break;
            }
            rep1 = rep0; rep0 = dist;
            const int end = d + mlen;
            const int ref = d - dist;
            if (ref < 0 || dist > maxDist || end > cap) { ok = 0; break; }
            // bytes this match reads may still be in flight from an earlier copy of this wave
            if (ref + (mlen < dist ? mlen : dist) > settled) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); settled = d; }
            if (dist >= mlen) wave_copy(dst + d, dst + ref, mlen, lane);
            else {
                const u32 ud = dist ? (u32)dist : 1u;          // distance 0 only occurs in corrupt input
                for (int i = lane; i < mlen; i += 64) dst[d + i] = dst[ref + (int)((u32)i % ud)];
            }
            d = end;
        }
        ok = ok && (s == litEnd);
    } while (0);
    if (lane == 0) { st.ok[b] = (u8)(ok ? 1 : 0); st.newLen[b] = (u32)d; }
}

size_t lz_scratch_u32(int ttype, int nBlocks, u32 maxLen)
{
    const size_t tab = (size_t)nBlocks << (ttype == KNZ_T_LZX ? 19 : 16);
    const size_t sec = ((size_t)maxLen + 64 + 15) & ~(size_t)15;
    return tab + (size_t)nBlocks * 3 * sec / 4 + 64;
}

void launch_lz_forward(hipStream_t s, const XfStage& st, int ttype)
{
    const int hashLog = (ttype == KNZ_T_LZX) ? 19 : 16;
    LzScratch ws;
    ws.tables = reinterpret_cast<int*>(st.scratchU32);
    ws.side = reinterpret_cast<u8*>(st.scratchU32 + ((size_t)st.nBlocks << hashLog));
    ws.secStride = ((size_t)st.maxLen + 64 + 15) & ~(size_t)15;
    hipMemsetAsync(ws.tables, 0, ((size_t)st.nBlocks << hashLog) * sizeof(int), s);
    KScope ks_("k_lz_forward");
    if (ttype == KNZ_T_LZX) hipLaunchKernelGGL((k_lz_forward<19, true>), dim3(st.nBlocks), dim3(64), 0, s, st, ws);
    else hipLaunchKernelGGL((k_lz_forward<16, false>), dim3(st.nBlocks), dim3(64), 0, s, st, ws);
}

void launch_lz_inverse(hipStream_t s, const XfStage& st) { KScope ks_("k_lz_inverse"); hipLaunchKernelGGL(k_lz_inverse, dim3(st.nBlocks), dim3(64), 0, s, st); }

}  // namespace knz
