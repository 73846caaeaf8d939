/* Study for the speculative parallel LZ parse: how soon does a parse started at a segment
 * boundary from a guessed state meet the true parse?  Not part of the product or of the oracle. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
enum { MAXD1 = 65534, MAXD2 = (1 << 24) - 2, MM = 4, MAXMATCH = 65535 + 254 + 4 };
static uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint32_t hsh(const uint8_t* p, unsigned hl) { return (uint32_t)(((ld64(p) << 24) * 0x1E35A7BDull) >> (64 - hl)); }
static int mlen(const uint8_t* s, int a, int b, int limit) { int n = 0; while (n + 8 <= limit) { uint64_t x = ld64(s + a + n) ^ ld64(s + b + n); if (x) return n + (__builtin_ctzll(x) >> 3); n += 8; } return n; }
static int imin(int a, int b) { return a < b ? a : b; }
typedef struct { int anchor, rep0, rep1; } Ev;

/* the reference's parse from an arbitrary state; candidates from prev[] when table == NULL, else from the table.
 * stops at pos >= end, or (stopAccel) when the skip counter reaches 64. returns #events; *why = 1 if stopped by accel */
static uint8_t* g_ins;   /* prev-based mode: 1 = position is in the table */
static int look(const int* prev, int q) { int c = prev[q]; while (c > 0 && !g_ins[c]) c = prev[c]; return c; }
static int parse(const uint8_t* src, int n, const int* prev, int32_t* table, unsigned hl, int extra,
                 int pos, int end, int stopAccel, Ev* ev, int maxEv, int* why)
{
    const int srcEnd = n - 18;
    const int maxDist = (srcEnd < 4 * MAXD1) ? MAXD1 : MAXD2;
    int anchor = pos, rep[2] = { n, n }, recent = 0, skip = 0, ne = 0;
    *why = 0;
    if (end > srcEnd) end = srcEnd;
    while (pos < end) {
        if (stopAccel && skip >= 64) { *why = 1; break; }
        int cand;
        if (table) { const uint32_t h = hsh(src + pos, hl); cand = table[h]; table[h] = pos; } else { cand = look(prev, pos); g_ins[pos] = 1; }
        const int nxt = pos + 1;
        const int lo = (pos - maxDist > 0) ? pos - maxDist : 0;
        int best = 0, ref = nxt - rep[recent];
        if (ref > lo && ld32(src + nxt) == ld32(src + ref)) best = mlen(src, nxt, ref, imin(srcEnd - nxt, MAXMATCH));
        else { ref = nxt - rep[recent ^ 1]; if (ref > lo && ld32(src + nxt) == ld32(src + ref)) best = mlen(src, nxt, ref, imin(srcEnd - nxt, MAXMATCH)); }
        if (best < MM) {
            ref = cand;
            if (ref > lo && ld32(src + pos) == ld32(src + ref)) best = mlen(src, pos, ref, imin(srcEnd - pos, MAXMATCH));
            if (best < MM) { pos = nxt + (skip >> 6); skip++; recent = 0; continue; }
            if (pos - ref != rep[0] && pos - ref != rep[1]) {
                const int origin = pos;
                for (int k = 1; k <= (extra ? 2 : 1); k++) {
                    const int pk = origin + k;
                    int ck;
                    if (table) { const uint32_t h = hsh(src + pk, hl); ck = table[h]; table[h] = pk; } else { ck = look(prev, pk); g_ins[pk] = 1; }
                    if (ck > lo + k && ld32(src + pk + best - 3) == ld32(src + ck + best - 3)) {
                        const int bk = mlen(src, pk, ck, imin(srcEnd - pk, MAXMATCH));
                        if (bk >= best) { ref = ck; best = bk; pos = pk; }
                    }
                }
            }
            while (pos > anchor && ref > lo && src[pos - 1] == src[ref - 1]) { best++; ref--; pos--; }
            if (best > MAXMATCH) { ref += best - MAXMATCH; pos += best - MAXMATCH; best = MAXMATCH; }
        } else {
            if (best >= MAXMATCH || src[pos] != src[ref - 1]) { pos++; if (table) table[hsh(src + pos, hl)] = pos; else g_ins[pos] = 1; }
            else { best++; ref--; }
        }
        skip = 0;
        const int dist = pos - ref;
        rep[1] = rep[0]; rep[0] = dist; recent = 1;
        anchor = pos + best;
        if (table) for (int p = pos + 1; p < anchor; p++) table[hsh(src + p, hl)] = p; else for (int p = pos + 1; p < anchor; p++) g_ins[p] = 1;
        pos = anchor;
        if (ne < maxEv) { ev[ne].anchor = anchor; ev[ne].rep0 = rep[0]; ev[ne].rep1 = rep[1]; ne++; }
    }
    return ne;
}

int main(int argc, char** argv)
{
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long fl = ftell(f); fseek(f, 0, SEEK_SET);
    const int bs = atoi(argv[2]), seg = atoi(argv[3]), extra = 1; const unsigned hl = 19; const int stopAccelFlag = argc > 4 ? atoi(argv[4]) : 0;
    uint8_t* all = malloc(fl + 64); if (fread(all, 1, fl, f) != (size_t)fl) return 2; fclose(f);
    long totSeg = 0, conv = 0, accel = 0, noconv = 0, wrong = 0; double sumOff = 0; long maxOff = 0, serialBytes = 0, totalBytes = 0;
    for (long off = 0; off < fl; off += bs) {
        const int n = (fl - off < bs) ? (int)(fl - off) : bs; if (n < 1024) break;
        const uint8_t* src = all + off; const int srcEnd = n - 18;
        int* prev = calloc((size_t)n + 8, sizeof(int)); int32_t* last = calloc((size_t)1 << hl, 4);
        for (int q = 0; q < srcEnd; q++) { uint32_t h = hsh(src + q, hl); prev[q] = last[h]; last[h] = q; }
        memset(last, 0, sizeof(int32_t) << hl);
        Ev* T = malloc(sizeof(Ev) * (size_t)(n / 4 + 16)); int why;
        const int nT = parse(src, n, prev, last, hl, extra, 0, srcEnd, 0, T, n / 4 + 16, &why);   /* the true parse */
        Ev* S = malloc(sizeof(Ev) * (size_t)(seg / 4 + 64));
        g_ins = malloc((size_t)n + 8);
        totalBytes += n; serialBytes += imin(seg, n);
        for (int s = seg; s < srcEnd; s += seg) {
            const int e = imin(s + seg, srcEnd);
            memset(g_ins, 1, (size_t)s); memset(g_ins + s, 0, (size_t)(n - s) + 8);
            const int nS = parse(src, n, prev, NULL, hl, extra, s, e, stopAccelFlag, S, seg / 4 + 64, &why);
            totSeg++;
            if (why) accel++;
            /* walk the true events of this segment; adopt runs of speculative events that coincide with them */
            int m = 0; while (m < nT && T[m].anchor < s) m++;
            int i = 0; long covered = 0; int firstMeet = -1, runs = 0;
            while (m < nT && T[m].anchor <= e && i < nS) {
                if (S[i].anchor < T[m].anchor) { i++; continue; }
                if (S[i].anchor > T[m].anchor) { m++; continue; }
                if (S[i].rep0 != T[m].rep0 || S[i].rep1 != T[m].rep1) { i++; m++; continue; }
                /* same state: follow both while they agree */
                if (firstMeet < 0) firstMeet = S[i].anchor - s;
                const int a0 = S[i].anchor; runs++;
                while (i + 1 < nS && m + 1 < nT && S[i + 1].anchor == T[m + 1].anchor && S[i + 1].rep0 == T[m + 1].rep0 && S[i + 1].rep1 == T[m + 1].rep1) { i++; m++; }
                covered += S[i].anchor - a0;
                i++; m++;
            }
            if (firstMeet < 0) noconv++; else { conv++; sumOff += firstMeet; if (firstMeet > maxOff) maxOff = firstMeet; }
            if (runs > 1) wrong++;
            serialBytes += (e - s) - covered;
        }
        free(g_ins);
        free(prev); free(last); free(T); free(S);
    }
    printf("%s bs=%d seg=%d: segments %ld met %ld (first meeting after %.0f B on average, max %ld B) needed more than one run %ld never-met %ld | serial share %.1f%%\n",
           argv[1], bs, seg, totSeg, conv, conv ? sumOff / conv : 0.0, maxOff, wrong, noconv, 100.0 * serialBytes / totalBytes);
    return 0;
}
