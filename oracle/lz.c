/*
 * lz.c -- oracle (TEST INFRASTRUCTURE ONLY, see knz_oracle.h): the LZ / LZX transform.
 *
 *   forward   transform/LZCodec.cpp:119-456   (LZXCodec<T>::forward; T=false "LZ": 16-bit hash, one look-ahead
 *                                               position; T=true "LZX": 19-bit hash, two look-ahead positions)
 *   inverse   transform/LZCodec.cpp:470-640   (inverseV6, bitstream version 6)
 *   helpers   transform/LZCodec.hpp:187-246   (hash, emitLength, readLength, findMatch), constants
 *             LZCodec.cpp:66-114, getMaxEncodedLength LZCodec.hpp:91-95
 *
 * Output of forward: 13-byte header (LE32 end of the literal section, LE32 #tokens, LE32 #distance bytes, flags) |
 * literal section (literal-run length extensions interleaved with the literal bytes) | tokens | distance bytes |
 * match-length extensions.  The data type hint of the reference's Context is always "undefined" on this path
 * (DESIGN.md section 4), so the minimum match is 4.
 */
#include "knz_oracle.h"
#include <stdlib.h>
#include <string.h>

enum { LZ_MAXD1 = (1 << 16) - 2, LZ_MAXD2 = (1 << 24) - 2, LZ_MINMATCH = 4, LZ_MAXMATCH = 65535 + 254 + 4, LZ_MINBLOCK = 24 };

static uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }   /* little-endian hosts only */
static uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

int knzo_lz_max_encoded(int n) { return ((n <= 1024) ? n + 16 : n + n / 64) + 2; }

/* LZCodec.hpp:187-190 */
static uint32_t lz_hash(const uint8_t* p, unsigned hashLog) { return (uint32_t)(((ld64(p) << 24) * 0x1E35A7BDull) >> (64 - hashLog)); }

/* LZCodec.hpp:227-246: whole 8-byte words only, so up to 7 bytes short of `limit` */
static int lz_match(const uint8_t* s, int a, int b, int limit)
{
    int n = 0;
    while (n + 8 <= limit) {
        const uint64_t x = ld64(s + a + n) ^ ld64(s + b + n);
        if (x) return n + (__builtin_ctzll(x) >> 3);
        n += 8;
    }
    return n;
}

/* LZCodec.hpp:192-210. The 3-byte form stores a 4th (zero) byte that the next write covers. */
static int lz_put_len(uint8_t* p, int len)
{
    if (len < 254) { p[0] = (uint8_t)len; return 1; }
    if (len < 65536 + 254) { const int v = len - 254; p[0] = 0xFE; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)v; p[3] = 0; return 3; }
    { const uint32_t v = (uint32_t)(len - 255); p[0] = 0xFF; p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; return 4; }
}

static int imin(int a, int b) { return a < b ? a : b; }

static int lz_forward_v6(const uint8_t* src, int n, uint8_t* dst, int dstCap, int extra, int* outLen)
{
    *outLen = 0;
    if (n == 0) return 1;
    if (dstCap < knzo_lz_max_encoded(n)) return 0;
    if (n < LZ_MINBLOCK) return 0;
    const unsigned hashLog = extra ? 19 : 16;
    int32_t* table = (int32_t*)calloc((size_t)1 << hashLog, sizeof(int32_t));
    /* token / distance / length-extension sections: their total stays below n or the block is rejected */
    uint8_t* tk = (uint8_t*)malloc((size_t)n + 16);
    uint8_t* mb = (uint8_t*)malloc((size_t)n + 16);
    uint8_t* ml = (uint8_t*)malloc((size_t)n + 16);
    int ok = 0;
    const int srcEnd = n - 16 - 2;
    const int maxDist = (srcEnd < 4 * LZ_MAXD1) ? LZ_MAXD1 : LZ_MAXD2;
    const int mm = LZ_MINMATCH;
    int pos = 0, d = 13, anchor = 0, nm = 0, nl = 0, nt = 0;
    int rep[2] = { n, n };
    int recent = 0, skip = 0;

    while (pos < srcEnd) {
        const uint32_t h = lz_hash(src + pos, hashLog);
        const int cand = table[h];
        table[h] = pos;
        const int nxt = pos + 1;
        const int lo = (pos - maxDist > 0) ? pos - maxDist : 0;
        int best = 0;
        int ref = nxt - rep[recent];
        if (ref > lo && ld32(src + nxt) == ld32(src + ref)) {
            best = lz_match(src, nxt, ref, imin(srcEnd - nxt, LZ_MAXMATCH));
        } else {
            ref = nxt - rep[recent ^ 1];
            if (ref > lo && ld32(src + nxt) == ld32(src + ref)) best = lz_match(src, nxt, ref, imin(srcEnd - nxt, LZ_MAXMATCH));
        }
        if (best < mm) {
            /* no usable repeat: the hash candidate at pos, then (new distances only) the next one or two positions */
            ref = cand;
            if (ref > lo && ld32(src + pos) == ld32(src + ref)) best = lz_match(src, pos, ref, imin(srcEnd - pos, LZ_MAXMATCH));
            if (best < mm) { pos = nxt + (skip >> 6); skip++; recent = 0; continue; }
            if (pos - ref != rep[0] && pos - ref != rep[1]) {
                const int p1 = nxt, p2 = nxt + 1;
                const uint32_t h1 = lz_hash(src + p1, hashLog);
                const int c1 = table[h1];
                table[h1] = p1;
                if (c1 > lo + 1 && ld32(src + p1 + best - 3) == ld32(src + c1 + best - 3)) {
                    const int b1 = lz_match(src, p1, c1, imin(srcEnd - p1, LZ_MAXMATCH));
                    if (b1 >= best) { ref = c1; best = b1; pos = p1; }
                }
                if (extra) {
                    const uint32_t h2 = lz_hash(src + p2, hashLog);
                    const int c2 = table[h2];
                    table[h2] = p2;
                    if (c2 > lo + 2 && ld32(src + p2 + best - 3) == ld32(src + c2 + best - 3)) {
                        const int b2 = lz_match(src, p2, c2, imin(srcEnd - p2, LZ_MAXMATCH));
                        if (b2 >= best) { ref = c2; best = b2; pos = p2; }
                    }
                }
            }
            while (pos > anchor && ref > lo && src[pos - 1] == src[ref - 1]) { best++; ref--; pos--; }
            if (best > LZ_MAXMATCH) { ref += best - LZ_MAXMATCH; pos += best - LZ_MAXMATCH; best = LZ_MAXMATCH; }
        } else {
            /* repeat match found at pos + 1: take the byte at pos with it when it matches too */
            if (best >= LZ_MAXMATCH || src[pos] != src[ref - 1]) { pos++; table[lz_hash(src + pos, hashLog)] = pos; }
            else { best++; ref--; }
        }
        skip = 0;
        const int dist = pos - ref;
        int token, th;
        if (dist == rep[0]) { token = 0x00; th = 3; }
        else if (dist == rep[1]) { token = 0x04; th = 3; }
        else {
            const int w3 = dist >= 65536, w2 = dist >= 256;
            mb[nm] = (uint8_t)(dist >> 16); nm += w3;
            mb[nm] = (uint8_t)(dist >> 8); nm += w2;
            mb[nm++] = (uint8_t)dist;
            token = (w3 + w2 + 1) << 3; th = 7;
        }
        const int mlen = best - mm;
        if (mlen >= th) { token += th; nl += lz_put_len(ml + nl, mlen - th); }
        else token += mlen;
        rep[1] = rep[0]; rep[0] = dist; recent = 1;
        const int lit = pos - anchor;
        if (lit == 0) tk[nt++] = (uint8_t)token;
        else {
            if (lit >= 7) {
                if (lit >= (1 << 24)) goto done;
                tk[nt++] = (uint8_t)((7 << 5) | token);
                d += lz_put_len(dst + d, lit - 7);
            } else tk[nt++] = (uint8_t)((lit << 5) | token);
            memcpy(dst + d, src + anchor, (size_t)lit);
            d += lit;
        }
        /* the three side sections only grow; once they and the literals cannot fit the block is rejected below */
        if (d + nt + nm + nl >= n) goto done;
        anchor = pos + best;
        for (int p = pos + 1; p < anchor; p++) table[lz_hash(src + p, hashLog)] = p;
        pos = anchor;
    }
    {
        const int lit = n - anchor;
        if (d + lit + nt + nm + nl >= n) goto done;
        if (lit >= 7) { tk[nt++] = (uint8_t)(7 << 5); d += lz_put_len(dst + d, lit - 7); }
        else tk[nt++] = (uint8_t)(lit << 5);
        memcpy(dst + d, src + anchor, (size_t)lit);
        d += lit;
        const uint32_t hd[3] = { (uint32_t)d, (uint32_t)nt, (uint32_t)nm };
        for (int k = 0; k < 3; k++) for (int j = 0; j < 4; j++) dst[4 * k + j] = (uint8_t)(hd[k] >> (8 * j));
        dst[12] = (uint8_t)((maxDist == LZ_MAXD1 ? 0 : 1) | (((mm - 2) & 7) << 1));
        memcpy(dst + d, tk, (size_t)nt); d += nt;
        memcpy(dst + d, mb, (size_t)nm); d += nm;
        memcpy(dst + d, ml, (size_t)nl); d += nl;
        *outLen = d;
        ok = d <= n - n / 100;
    }
done:
    free(table); free(tk); free(mb); free(ml);
    return ok;
}

/* LZCodec.hpp:212-225; pos/limit guard added (the reference reads up to 2 bytes past the block) */
static uint32_t lz_get_len(const uint8_t* s, int* pos, int limit)
{
    uint32_t b[4] = { 0, 0, 0, 0 };
    for (int k = 0; k < 4; k++) if (*pos + k < limit) b[k] = s[*pos + k];
    if (b[0] < 254) { *pos += 1; return b[0]; }
    if (b[0] == 254) { *pos += 3; return 254 + ((b[1] << 8) | b[2]); }
    *pos += 4;
    return 255 + ((b[1] << 16) | (b[2] << 8) | b[3]);
}

static int lz_inverse_v6(const uint8_t* src, int n, uint8_t* dst, int dstCap, int* outLen)
{
    *outLen = 0;
    if (n == 0) return 1;
    if (n < 13) return 0;
    const int32_t litEnd = (int32_t)ld32(src), nTok = (int32_t)ld32(src + 4), nDist = (int32_t)ld32(src + 8);
    if (litEnd < 0 || nTok < 0 || nDist < 0) return 0;
    if (litEnd < 13 || litEnd > n || nTok > n - litEnd || nDist > n - litEnd - nTok) return 0;
    int t = litEnd;                  /* cursor in the token section */
    int m = litEnd + nTok;           /* ... distance section */
    int l = m + nDist;               /* ... match-length extension section */
    const int maxDist = (src[12] & 1) ? LZ_MAXD2 : LZ_MAXD1;
    const int mm = ((src[12] >> 1) & 7) + 2;
    int s = 13, d = 0, rep0 = n, rep1 = n, ok = 1;
    for (;;) {
        /* reads past the block are undefined in the reference (it relies on 2 bytes of padding): zeros here */
        const int token = (t < n) ? src[t] : 0;
        t++;
        int mlen, dist;
        if ((token & 0x18) == 0) {
            mlen = token & 3;
            mlen = (mlen == 3) ? 3 + mm + (int)lz_get_len(src, &l, n) : mlen + mm;
            dist = (token & 4) ? rep1 : rep0;
        } else {
            mlen = token & 7;
            mlen = (mlen == 7) ? 7 + mm + (int)lz_get_len(src, &l, n) : mlen + mm;
            const int nb = (token >> 3) & 3;          /* 1, 2 or 3 distance bytes, most significant first */
            dist = 0;
            for (int k = 0; k < nb; k++) { dist = (dist << 8) | ((m < n) ? src[m] : 0); m++; }
        }
        if (token >= 32) {
            const uint32_t lit = (token >= 0xE0) ? 7u + lz_get_len(src, &s, n) : (uint32_t)(token >> 5);
            if (lit > (uint32_t)(dstCap - d) || lit > (uint32_t)(litEnd - s)) { ok = 0; break; }
            memcpy(dst + d, src + s, lit);
            s += (int)lit; d += (int)lit;
            if (s >= litEnd - 13) break;
        }
        rep1 = rep0; rep0 = dist;
        const int end = d + mlen;
        int ref = d - dist;
        if (ref < 0 || dist > maxDist || end > dstCap) { ok = 0; break; }
        while (d < end) dst[d++] = dst[ref++];
    }
    *outLen = d;
    return ok && s == litEnd;
}

/* ------------------------------------------------------------------ the block layout of bitstream versions below 6
 * LZCodec.cpp:614-760 (inverseV5) reads it; the reference has no writer for it any more. Same four sections behind the same 13-byte
 * header, other token: bits 7-5 literal run (7: + extension), bits 3-0 match length - minMatch (14: + extension; 15: a repeat
 * distance, length always in the extension, bit 4 picks the older one), bit 4 otherwise: one more distance byte than the flag byte's
 * bit 0 gives (1 or 2); flag bits 2-1 index the minimum match { 4, 9, 6, 6 }. */
static int lz_inverse_v5(const uint8_t* src, int n, uint8_t* dst, int dstCap, int* outLen)
{
    *outLen = 0;
    if (n == 0) return 1;
    if (n < 13) return 0;
    const int32_t tk0 = (int32_t)ld32(src), nTok = (int32_t)ld32(src + 4), nDist = (int32_t)ld32(src + 8);
    if (tk0 < 0 || nTok < 0 || nDist < 0) return 0;
    if (tk0 < 13 || tk0 > n || nTok > n - tk0 || nDist > n - tk0 - nTok) return 0;
    int t = tk0, m = tk0 + nTok, l = m + nDist;
    const int srcEnd = tk0 - 13, litEnd = tk0;
    const int mFlag = src[12] & 1;
    const int maxDist = mFlag ? LZ_MAXD2 : LZ_MAXD1;
    static const int MINM[4] = { 4, 9, 6, 6 };
    const int mm = MINM[(src[12] >> 1) & 3];
    int s = 13, d = 0, rep0 = 0, rep1 = 0, ok = 1;
    for (;;) {
        const int token = (t < n) ? src[t] : 0;
        t++;
        if (token >= 32) {
            const uint32_t lit = (token >= 0xE0) ? 7u + lz_get_len(src, &s, n) : (uint32_t)(token >> 5);
            if (lit > (uint32_t)(dstCap - d) || lit > (uint32_t)(litEnd - s)) { ok = 0; break; }
            memcpy(dst + d, src + s, lit);
            s += (int)lit; d += (int)lit;
            if (s >= srcEnd) break;
        }
        int mlen = token & 0x0F, dist;
        if (mlen == 15) {
            mlen = mm + (int)lz_get_len(src, &l, n);
            dist = (token & 0x10) ? rep1 : rep0;
        } else {
            mlen = (mlen == 14) ? 14 + mm + (int)lz_get_len(src, &l, n) : mlen + mm;
            dist = (m < n) ? src[m] : 0; m++;
            if (mFlag) { dist = (dist << 8) | ((m < n) ? src[m] : 0); m++; }
            if (token & 0x10) { dist = (dist << 8) | ((m < n) ? src[m] : 0); m++; }
        }
        rep1 = rep0; rep0 = dist;
        const int end = d + mlen;
        int ref = d - dist;
        if (ref < 0 || dist > maxDist || end > dstCap) { ok = 0; break; }
        while (d < end) dst[d++] = dst[ref++];
    }
    *outLen = d;
    return ok && s == srcEnd + 13;
}

/* test writer: the current layout of a block, sequence by sequence, in the old tokens */
static int lz_current_to_v5(const uint8_t* src, int n, uint8_t* dst, int dstCap, int* outLen)
{
    *outLen = 0;
    if (n < 13) return 0;
    const int litEnd = (int)ld32(src), nTok = (int)ld32(src + 4), nDist = (int)ld32(src + 8);
    int t = litEnd, m = litEnd + nTok, l = m + nDist;
    const int mFlag = src[12] & 1;
    const int mm = ((src[12] >> 1) & 7) + 2;
    int mmIdx = -1;
    if (mm == 4) mmIdx = 0; else if (mm == 9) mmIdx = 1; else if (mm == 6) mmIdx = 2;
    if (mmIdx < 0) return 0;
    uint8_t* tk = (uint8_t*)malloc((size_t)n + 16);
    uint8_t* mb = (uint8_t*)malloc((size_t)3 * n + 16);
    uint8_t* ml = (uint8_t*)malloc((size_t)4 * n + 16);
    int nt = 0, nm = 0, nl = 0, s = 13, ok = 1;
    for (;;) {
        const int token = src[t++];
        int mlen, dist = 0, rep = -1;
        if ((token & 0x18) == 0) {
            mlen = token & 3;
            mlen = (mlen == 3) ? 3 + mm + (int)lz_get_len(src, &l, n) : mlen + mm;
            rep = (token & 4) ? 1 : 0;
        } else {
            mlen = token & 7;
            mlen = (mlen == 7) ? 7 + mm + (int)lz_get_len(src, &l, n) : mlen + mm;
            const int nb = (token >> 3) & 3;
            for (int k = 0; k < nb; k++) dist = (dist << 8) | src[m++];
        }
        int out = token & 0xE0;
        if (token >= 32) {
            const uint32_t lit = (token >= 0xE0) ? 7u + lz_get_len(src, &s, n) : (uint32_t)(token >> 5);
            s += (int)lit;
            if (s >= litEnd - 13) { tk[nt++] = (uint8_t)out; break; }     /* the last token: literals only */
        }
        if (rep >= 0) {
            out |= 15 | (rep ? 0x10 : 0);
            nl += lz_put_len(ml + nl, mlen - mm);
        } else {
            const int wide = mFlag ? (dist >= 65536) : (dist >= 256);
            if (wide) { out |= 0x10; }
            if (mFlag && wide) mb[nm++] = (uint8_t)(dist >> 16);
            if (mFlag || wide) mb[nm++] = (uint8_t)(dist >> 8);
            mb[nm++] = (uint8_t)dist;
            const int c = mlen - mm;
            if (c >= 14) { out |= 14; nl += lz_put_len(ml + nl, c - 14); } else out |= c;
        }
        tk[nt++] = (uint8_t)out;
    }
    const int total = litEnd + nt + nm + nl;
    if (total > dstCap) ok = 0;
    if (ok) {
        memcpy(dst, src, (size_t)litEnd);                   /* header + literal section: the same bytes */
        const uint32_t hd[3] = { (uint32_t)litEnd, (uint32_t)nt, (uint32_t)nm };
        for (int k = 0; k < 3; k++) for (int j = 0; j < 4; j++) dst[4 * k + j] = (uint8_t)(hd[k] >> (8 * j));
        dst[12] = (uint8_t)(mFlag | (mmIdx << 1));
        memcpy(dst + litEnd, tk, (size_t)nt);
        memcpy(dst + litEnd + nt, mb, (size_t)nm);
        memcpy(dst + litEnd + nt + nm, ml, (size_t)nl);
        *outLen = total;
    }
    free(tk); free(mb); free(ml);
    return ok;
}

int knzo_lz_forward(const uint8_t* src, int n, uint8_t* dst, int dstCap, int extra, int* outLen)
{
    if (knzo_get_bs_version() >= 6) return lz_forward_v6(src, n, dst, dstCap, extra, outLen);
    *outLen = 0;
    if (n == 0) return 1;
    uint8_t* cur = (uint8_t*)malloc((size_t)knzo_lz_max_encoded(n) + 64);
    int cl = 0, ok = lz_forward_v6(src, n, cur, knzo_lz_max_encoded(n), extra, &cl);
    if (ok) ok = lz_current_to_v5(cur, cl, dst, dstCap, outLen) && *outLen <= n - n / 100;
    if (!ok) *outLen = 0;
    free(cur);
    return ok;
}

int knzo_lz_inverse(const uint8_t* src, int n, uint8_t* dst, int dstCap, int* outLen)
{
    return knzo_get_bs_version() < 6 ? lz_inverse_v5(src, n, dst, dstCap, outLen) : lz_inverse_v6(src, n, dst, dstCap, outLen);
}
