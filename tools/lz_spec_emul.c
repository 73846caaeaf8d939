/* Executable specification of the speculative parallel LZ/LZX parse planned in DESIGN.md section 7 (round 2).
 * Sequential emulation of what three kernels would do -- (1) cold parses of all segments, (2) an in-order stitch
 * that re-parses from the true state until it meets a segment's speculative trajectory, validates the lookups of
 * that trajectory against the positions really skipped so far, takes it over, and falls back / rejoins where a
 * lookup is inconsistent, (3) emission of the byte sections from the final event list -- checked against the plain
 * serial parse: identical events, identical output bytes.  Not part of the product or of the oracle.
 *   gcc -O2 -o lz_spec_emul tools/lz_spec_emul.c && ./lz_spec_emul FILE BLOCK SEGMENT [lzx=1]                   */
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

typedef struct { int pos, anchor, rep0, rep1, recent, skip; } State;
typedef struct { int lit0, pos, best, dist, rep0, rep1, lk0, lk1; } Event;      /* after the event: anchor = pos + best */
typedef struct { int q, cand; } Lookup;
typedef struct {                       /* which positions are in the table */
    uint8_t* ins; int base, span;      /* local: ins[p - base] for base <= p < base + span; below base: assumed present */
    int global;                        /* global: ins[p] for every p */
    int overflow;
} View;
typedef struct { const uint8_t* src; int n, srcEnd, maxDist, extra; const int* prev; } Ctx;

static int present(const View* v, int c) { return v->global ? v->ins[c] : (c < v->base ? 1 : v->ins[c - v->base]); }
static void mark(View* v, int p) { if (v->global) v->ins[p] = 1; else if (p - v->base < v->span) v->ins[p - v->base] = 1; else v->overflow = 1; }
static int look(const Ctx* cx, const View* v, int q) { int c = cx->prev[q]; while (c > 0 && !present(v, c)) c = cx->prev[c]; return c; }

/* The reference's loop from state *st until one match has been emitted (returns 1, *ev filled, lookups appended) or
 * pos >= limit (returns 0; the state is then in the middle of a literal run). */
static int step(const Ctx* cx, View* v, State* st, int limit, Event* ev, Lookup* lk, int* nlk, int maxlk)
{
    const uint8_t* src = cx->src;
    const int srcEnd = cx->srcEnd, maxDist = cx->maxDist;
    int pos = st->pos, anchor = st->anchor, recent = st->recent, skip = st->skip;
    int rep[2] = { st->rep0, st->rep1 };
    const int lk0 = *nlk;
    while (pos < srcEnd && pos < limit && !v->overflow) {
        const int cand = look(cx, v, pos);
        if (*nlk < maxlk) { lk[*nlk].q = pos; lk[*nlk].cand = cand; (*nlk)++; }
        mark(v, pos);
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
                for (int k = 1; k <= (cx->extra ? 2 : 1); k++) {
                    const int pk = origin + k;
                    const int ck = look(cx, v, pk);
                    if (*nlk < maxlk) { lk[*nlk].q = pk; lk[*nlk].cand = ck; (*nlk)++; }
                    mark(v, pk);
                    if (ck > lo + k && ld32(src + pk + best - 3) == ld32(src + ck + best - 3)) {
                        const int bk = mlen(src, pk, ck, imin(srcEnd - pk, MAXMATCH));
                        if (bk >= best) { ref = ck; best = bk; pos = pk; }
                    }
                }
            }
            while (pos > anchor && ref > lo && src[pos - 1] == src[ref - 1]) { best++; ref--; pos--; }
            if (best > MAXMATCH) { ref += best - MAXMATCH; pos += best - MAXMATCH; best = MAXMATCH; }
        } else {
            if (best >= MAXMATCH || src[pos] != src[ref - 1]) { pos++; mark(v, pos); }
            else { best++; ref--; }
        }
        const int dist = pos - ref;
        rep[1] = rep[0]; rep[0] = dist;
        for (int p = pos + 1; p < pos + best; p++) mark(v, p);
        ev->lit0 = anchor; ev->pos = pos; ev->best = best; ev->dist = dist; ev->rep0 = rep[0]; ev->rep1 = rep[1]; ev->lk0 = lk0; ev->lk1 = *nlk;
        st->pos = st->anchor = pos + best; st->rep0 = rep[0]; st->rep1 = rep[1]; st->recent = 1; st->skip = 0;
        return 1;
    }
    st->pos = pos; st->anchor = anchor; st->rep0 = rep[0]; st->rep1 = rep[1]; st->recent = recent; st->skip = skip;
    return 0;
}

/* emission of the byte sections from an event list (LZCodec.cpp:316-454 in event form). returns output length or -1 */
static int put_len(uint8_t* p, int len)
{
    if (len < 254) { p[0] = (uint8_t)len; return 1; }
    if (len < 65536 + 254) { const int v = len - 254; p[0] = 0xFE; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)v; return 3; }
    { const uint32_t v = (uint32_t)(len - 255); p[0] = 0xFF; p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; return 4; }
}
static int emit(const Ctx* cx, const Event* ev, int nev, uint8_t* dst)
{
    const int n = cx->n; const uint8_t* src = cx->src;
    uint8_t* tk = malloc((size_t)n + 16); uint8_t* mb = malloc((size_t)n + 16); uint8_t* ml = malloc((size_t)n + 16);
    int d = 13, nt = 0, nm = 0, nl = 0, rep0 = n, rep1 = n, anchor = 0, res = -1;
    for (int i = 0; i < nev; i++) {
        const int dist = ev[i].dist; int token, th;
        if (dist == rep0) { token = 0; th = 3; } else if (dist == rep1) { token = 4; th = 3; }
        else { const int w3 = dist >= 65536, w2 = dist >= 256; if (w3) mb[nm++] = (uint8_t)(dist >> 16); if (w2) mb[nm++] = (uint8_t)(dist >> 8); mb[nm++] = (uint8_t)dist; token = (w3 + w2 + 1) << 3; th = 7; }
        const int ml_ = ev[i].best - MM;
        if (ml_ >= th) { token += th; nl += put_len(ml + nl, ml_ - th); } else token += ml_;
        rep1 = rep0; rep0 = dist;
        const int lit = ev[i].pos - anchor;
        if (lit == 0) tk[nt++] = (uint8_t)token;
        else {
            if (lit >= 7) { if (lit >= (1 << 24)) goto out; tk[nt++] = (uint8_t)(0xE0 | token); d += put_len(dst + d, lit - 7); }
            else tk[nt++] = (uint8_t)((lit << 5) | token);
            memcpy(dst + d, src + anchor, (size_t)lit); d += lit;
        }
        if (d + nt + nm + nl >= n) goto out;
        anchor = ev[i].pos + ev[i].best;
    }
    {
        const int lit = n - anchor;
        if (d + lit + nt + nm + nl >= n) goto out;
        if (lit >= 7) { tk[nt++] = 0xE0; d += put_len(dst + d, lit - 7); } else tk[nt++] = (uint8_t)(lit << 5);
        memcpy(dst + d, src + anchor, (size_t)lit); d += lit;
        const uint32_t hd[3] = { (uint32_t)d, (uint32_t)nt, (uint32_t)nm };
        for (int k = 0; k < 3; k++) for (int j = 0; j < 4; j++) dst[4 * k + j] = (uint8_t)(hd[k] >> (8 * j));
        dst[12] = (uint8_t)((cx->maxDist == MAXD1 ? 0 : 1) | (((MM - 2) & 7) << 1));
        memcpy(dst + d, tk, (size_t)nt); d += nt; memcpy(dst + d, mb, (size_t)nm); d += nm; memcpy(dst + d, ml, (size_t)nl); d += nl;
        res = d;
    }
out:
    free(tk); free(mb); free(ml);
    return res;
}

typedef struct { Event* ev; int nev; Lookup* lk; int nlk; uint8_t* ins; int base, span; } Seg;

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s FILE BLOCK SEGMENT [lzx=1]\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb"); if (!f) return 2;
    fseek(f, 0, SEEK_END); const long fl = ftell(f); fseek(f, 0, SEEK_SET);
    const int bs = atoi(argv[2]), SEG = atoi(argv[3]), extra = argc > 4 ? atoi(argv[4]) : 1; const unsigned hl = extra ? 19 : 16;
    uint8_t* all = malloc((size_t)fl + 64); if (fread(all, 1, (size_t)fl, f) != (size_t)fl) return 2; fclose(f);
    long tEvents = 0, tAdopted = 0, tSerialSteps = 0, tVisitsSerial = 0, tVisitsAll = 0, tCuts = 0, blocks = 0; int bad = 0;
    for (long off = 0; off < fl; off += bs) {
        const int n = (fl - off < bs) ? (int)(fl - off) : bs; if (n < 1024) break;
        blocks++;
        Ctx cx; cx.src = all + off; cx.n = n; cx.srcEnd = n - 18; cx.maxDist = (cx.srcEnd < 4 * MAXD1) ? MAXD1 : MAXD2; cx.extra = extra;
        int* prev = calloc((size_t)n + 8, sizeof(int)); int32_t* last = calloc((size_t)1 << hl, 4);
        for (int q = 0; q < cx.srcEnd; q++) { uint32_t h = hsh(cx.src + q, hl); prev[q] = last[h]; last[h] = q; }
        free(last); cx.prev = prev;
        const int maxEv = n / 4 + 16;
        /* ---- the plain serial parse: the truth to compare with */
        Event* T = malloc(sizeof(Event) * (size_t)maxEv); int nT = 0;
        {
            View g; g.ins = calloc((size_t)n + 8, 1); g.global = 1; g.overflow = 0; g.base = 0; g.span = n;
            State st = { 0, 0, n, n, 0, 0 }; Lookup* lk = malloc(sizeof(Lookup) * ((size_t)n + 16)); int nlk = 0;
            while (step(&cx, &g, &st, cx.srcEnd, &T[nT], lk, &nlk, n + 16)) { nT++; nlk = 0; }
            tVisitsAll += 0; free(lk); free(g.ins);
        }
        /* ---- (1) cold parses of all segments */
        const int K = (cx.srcEnd + SEG - 1) / SEG;
        Seg* sg = calloc((size_t)K, sizeof(Seg));
        const int span = 2 * SEG + MAXMATCH + 64;
        for (int k = 1; k < K; k++) {
            Seg* s = &sg[k]; s->base = k * SEG; s->span = span; s->ins = calloc((size_t)span + 8, 1);
            const int cap = SEG / 2 + 64; s->ev = malloc(sizeof(Event) * (size_t)cap); s->lk = malloc(sizeof(Lookup) * (size_t)(2 * SEG + 64) * 3);
            View v; v.ins = s->ins; v.base = s->base; v.span = span; v.global = 0; v.overflow = 0;
            State st = { s->base, s->base, n, n, 0, 0 };
            const int limit = imin(s->base + 2 * SEG, cx.srcEnd);
            while (s->nev < cap && st.pos < s->base + SEG && step(&cx, &v, &st, limit, &s->ev[s->nev], s->lk, &s->nlk, (2 * SEG + 64) * 3)) s->nev++;
        }
        /* ---- (2) in-order stitch */
        Event* F = malloc(sizeof(Event) * (size_t)maxEv); int nF = 0;
        View g; g.ins = calloc((size_t)n + 8, 1); g.global = 1; g.overflow = 0; g.base = 0; g.span = n;
        State st = { 0, 0, n, n, 0, 0 };
        Lookup* lkTmp = malloc(sizeof(Lookup) * ((size_t)n + 16));
        long adopted = 0, serialSteps = 0, cuts = 0;
        while (st.pos < cx.srcEnd) {
            /* try to meet the speculative trajectory of the segment the true state is in (only right after a match) */
            int took = 0;
            const int k = st.pos / SEG;
            if (k >= 1 && k < K && st.recent == 1 && st.skip == 0 && st.anchor == st.pos) {
                const Seg* s = &sg[k];
                int lo = 0, hi = s->nev - 1, j = -1;
                while (lo <= hi) { const int m = (lo + hi) >> 1; const int a = s->ev[m].pos + s->ev[m].best; if (a < st.pos) lo = m + 1; else if (a > st.pos) hi = m - 1; else { j = m; break; } }
                if (j >= 0 && s->ev[j].rep0 == st.rep0 && s->ev[j].rep1 == st.rep1) {
                    /* events j+1 .. are candidates; validate their lookups against the truth below A = st.pos */
                    const int A = st.pos;
                    int e = j + 1;
                    for (; e < s->nev; e++) {
                        int okEv = 1;
                        for (int x = s->ev[e].lk0; x < s->ev[e].lk1 && okEv; x++) {
                            const int cs = s->lk[x].cand;
                            if (cs >= A) continue;                         /* decided inside the trusted range */
                            int c = prev[s->lk[x].q];
                            while (c >= A) c = prev[c];
                            while (c > 0 && !g.ins[c]) c = prev[c];
                            if (c != cs) okEv = 0;
                        }
                        if (!okEv) { cuts++; break; }
                    }
                    if (e > j + 1) {
                        /* take over events j+1 .. e-1: the table state of their byte range is the segment's local one */
                        const int Aend = s->ev[e - 1].pos + s->ev[e - 1].best;
                        for (int p = A; p < Aend && p - s->base < s->span; p++) g.ins[p] = s->ins[p - s->base];
                        for (int x = j + 1; x < e; x++) { F[nF++] = s->ev[x]; adopted++; }
                        st.pos = st.anchor = Aend; st.rep0 = s->ev[e - 1].rep0; st.rep1 = s->ev[e - 1].rep1; st.recent = 1; st.skip = 0;
                        took = 1;
                    }
                }
            }
            if (took) continue;
            int nlk = 0;
            if (!step(&cx, &g, &st, cx.srcEnd, &F[nF], lkTmp, &nlk, n + 16)) break;
            tVisitsSerial += nlk;
            nF++; serialSteps++;
        }
        /* ---- compare with the plain parse, then (3) emit and compare bytes */
        int same = nF == nT;
        for (int i = 0; i < nF && same; i++) same = F[i].pos == T[i].pos && F[i].best == T[i].best && F[i].dist == T[i].dist && F[i].lit0 == T[i].lit0;
        uint8_t* o1 = malloc((size_t)n + n / 32 + 1024); uint8_t* o2 = malloc((size_t)n + n / 32 + 1024);
        const int l1 = emit(&cx, T, nT, o1), l2 = emit(&cx, F, nF, o2);
        const int sameBytes = l1 == l2 && (l1 < 0 || memcmp(o1, o2, (size_t)l1) == 0);
        if (!same || !sameBytes) { bad++; fprintf(stderr, "MISMATCH block at %ld: events %d vs %d same=%d bytes=%d\n", off, nF, nT, same, sameBytes); }
        tEvents += nT; tAdopted += adopted; tSerialSteps += serialSteps; tCuts += cuts;
        for (int k = 1; k < K; k++) { free(sg[k].ev); free(sg[k].lk); free(sg[k].ins); }
        free(sg); free(F); free(T); free(prev); free(g.ins); free(lkTmp); free(o1); free(o2);
    }
    printf("%s block=%d seg=%d %s: %ld blocks, %ld events, taken from speculative parses %ld (%.1f%%), serial %ld, cuts %ld, positions visited serially %ld -> %s\n",
           argv[1], bs, SEG, extra ? "LZX" : "LZ", blocks, tEvents, tAdopted, 100.0 * tAdopted / (tEvents ? tEvents : 1), tSerialSteps, tCuts, tVisitsSerial, bad ? "MISMATCH" : "identical to the serial parse");
    return bad != 0;
}
