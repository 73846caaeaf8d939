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
#include <cstring>
#include <cstdlib>
#include <atomic>
#include <algorithm>
#include "prims.hpp"

namespace knz {

constexpr int LZ_MAXD1 = (1 << 16) - 2;
constexpr int LZ_MAXD2 = (1 << 24) - 2;
constexpr int LZ_MM = 4;                                  // data type hint is always "undefined" on this path
constexpr int LZ_MAXMATCH = 65535 + 254 + 4;
constexpr int LZ_MINBLOCK = 24;

__host__ __device__ inline int lz_max_encoded(int n) { return ((n <= 1024) ? n + 16 : n + n / 64) + 2; }

// unaligned loads of block text, through the GLOBAL address space: a block's pointer comes out of an array of pointers, so the compiler
// would make every access a flat load, which counts as an LDS operation too -- and the walk waits for its LDS windows all the time
#ifdef KNZ_EMU
__device__ __forceinline__ u64 ld64u(const u8* p) { u64 v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ u32 ld32u(const u8* p) { u32 v; __builtin_memcpy(&v, p, 4); return v; }
#else
typedef u64 __attribute__((aligned(1))) lz_u64_u;
typedef u32 __attribute__((aligned(1))) lz_u32_u;
__device__ __forceinline__ u64 ld64u(const u8* p) { return *(const __attribute__((address_space(1))) lz_u64_u*)(uintptr_t)p; }
__device__ __forceinline__ u32 ld32u(const u8* p) { return *(const __attribute__((address_space(1))) lz_u32_u*)(uintptr_t)p; }
#endif
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
    KNZ_WAVE_ORDER();                                     // every lane's OR before any lane's AND (one instruction each on the GPU)
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
__device__ int lz_match(const u8* s, int a, int b, int limit, int lane, int n = 0)
{
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

// ------------------------------------------------------------------------------------------------
// Encoder.  What sits in bucket hash(q) when the reference looks at position q is, with one exception, simply the
// previous position with the same hash: every position below q has been stored by then (visited positions by the
// step that visited them, positions under a match by the fill behind it).  The exception are the positions the
// reference jumps over once a literal run has seen 64 misses (its stride then grows); they are only stored if a
// later match extends back over them.  So a first phase computes for ALL positions of ALL blocks in parallel
//   prev[q]  = previous position with the same hash (stable radix sort of (block, hash) keys, neighbours compared),
//   info[q]  = findMatch(q, prev[q]) up to LZ_LCAP bytes, plus "stopped at a difference" (exact) in bit 15,
// and the serial walk (one wave per block) uses prev[q] as the bucket content whenever prev[q] lies above the highest
// position ever jumped over (`maxGap`), or is 0; only otherwise it reads the real table, which is still kept exactly
// as the reference leaves it (stores only, off the dependent chain).  Checked on the CPU against the reference's
// own table at every lookup (zero mispredictions; 0.3 - 17 % of the lookups take the table path).
// ------------------------------------------------------------------------------------------------
constexpr int LZ_LCAP = 248;          // longest candidate length phase 1 measures (multiple of 8)
constexpr int LZW = 1024;             // positions per LDS window of the walk

struct LzWs {
    int* tables; u8* side; size_t secStride;
    u32* keysA; u32* keysB; u32* valsA; u32* valsB; u32* prev; u16* info;      // (keysA = the hash of every position: the walk reads it too)
    u32 S;                            // positions per block in the flat arrays
};

template <int HL>
__global__ __launch_bounds__(256) void k_lz_keys(XfStage st, u32 S, int nBlocks, u32* __restrict__ keys, u32* __restrict__ vals)
{
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (size_t)nBlocks * S) return;
    const int b = (int)(g / S);
    const u32 q = (u32)(g - (size_t)b * S);
    const int n = (int)st.len[b];
    const bool live = n >= LZ_MINBLOCK && st.cap[b] >= (u32)lz_max_encoded(n) && (int)q < n - 18;
    keys[g] = live ? (((u32)b << HL) | lz_hash<HL>(ld64u(st.src[b] + q))) : ((u32)nBlocks << HL);
    vals[g] = (u32)g;
}

template <int HL>
__global__ __launch_bounds__(256) void k_lz_prev(XfStage st, u32 S, int nBlocks, size_t total, const u32* __restrict__ keysS,
                                                 const u32* __restrict__ valsS, u32* __restrict__ prev, u16* __restrict__ info)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const u32 key = keysS[i];
    const int b = (int)(key >> HL);
    if (b >= nBlocks) return;
    const u32 g = valsS[i];
    const u32 base = (u32)b * S;
    const u32 q = g - base;
    u32 pv = 0;
    if (i > 0 && keysS[i - 1] == key) pv = valsS[i - 1] - base;
    prev[g] = pv;
    u32 inf = 0;
    if (pv) {
        const u8* __restrict__ src = st.src[b];
        const int srcEnd = (int)st.len[b] - 18;
        const int limit = min(srcEnd - (int)q, LZ_MAXMATCH);
        int l = 0;
        while (l + 8 <= limit && l < LZ_LCAP) {
            const u64 x = ld64u(src + q + l) ^ ld64u(src + pv + l);
            if (x) { l += (int)(__ffsll((long long)x) - 1) >> 3; inf = 0x8000u; break; }
            l += 8;
        }
        inf |= (u32)l;
    }
    info[g] = (u16)inf;
}

__device__ __forceinline__ u64 readlane64(u64 v, int l)
{
    return ((u64)(u32)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (u32)__builtin_amdgcn_readlane((int)v, l);
}

// The walk.  Positions of a literal run do not depend on one another except through the table, so the next <= 64
// positions the reference would visit are tested at once (lane i = i-th visit): bucket candidate (from prev / info,
// or the table) and the two repeat distances (one 8-byte gather each: the byte before the candidate and the four
// bytes compared).  Lanes before the first one with a possible match are misses for sure: their buckets are
// stored and the serial state jumps over them.  The first lane with a possible match goes through the reference's
// step with everything it needs already in registers; its look-ahead candidates come from the same window.
// While the reference's stride is 1 the per-position data comes from an LDS window over phase 1's arrays; with a
// larger stride (an incompressible stretch) the batch reads source and table directly, and a lane whose bucket was
// last written by an earlier lane of the batch takes that lane's position (exact search, run when a 64 Kbit LDS
// filter over the hashes reports a possible duplicate).
template <int HASH_LOG, bool EXTRA>
__global__ __launch_bounds__(64) void k_lz_walk(XfStage st, LzWs ws)
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
    const size_t gbase = (size_t)b * ws.S;
    const u32* __restrict__ gPrev = ws.prev + gbase;
    const u32* __restrict__ gKeys = ws.keysA + gbase;
    const u16* __restrict__ gInfo = ws.info + gbase;
    constexpr u32 HMASK = (1u << HASH_LOG) - 1;
    __shared__ u32 seen[2048];
    __shared__ u32 wPrev[LZW];
    __shared__ u32 wHash[LZW];
    __shared__ u16 wInfo[LZW];
    __shared__ u32 wSrc[LZW / 4 + 4];
    for (int i = lane; i < 2048; i += 64) seen[i] = 0;
    __syncthreads();
    if (st.cap[b] >= (u32)lz_max_encoded(n) && n >= LZ_MINBLOCK) {
        const int srcEnd = n - 16 - 2;
        const int maxDist = (srcEnd < 4 * LZ_MAXD1) ? LZ_MAXD1 : LZ_MAXD2;
        int pos = 0, d = 13, anchor = 0, nm = 0, nl = 0, nt = 0;
        int rep0 = n, rep1 = n, recent = 0, skip = 0;
        int maxGap = -1;
        int wbase = -(1 << 30);
        bool reject = false;
        auto refill = [&](int from) {
            wbase = from;
            __syncthreads();
#pragma unroll 4
            for (int k = 0; k < LZW / 64; k++) {
                const int idx = lane + 64 * k;
                const u32 g = (u32)(from + idx);
                const bool in = g < ws.S;
                wPrev[idx] = in ? gPrev[g] : 0u;
                wHash[idx] = in ? (gKeys[g] & HMASK) : 0u;
                wInfo[idx] = in ? gInfo[g] : (u16)0;
            }
            for (int k = lane; k < LZW / 4 + 4; k += 64) {
                const int o = from + 4 * k;
                u32 v = 0;
                if (o + 4 <= n) v = ld32u(src + o);
                else for (int j = 0; j < 4; j++) if (o + j < n) v |= (u32)src[o + j] << (8 * j);
                wSrc[k] = v;
            }
            __syncthreads();
        };
        auto win4 = [&](int o) -> u32 {                       // 4 source bytes at window offset o
            const u32 a = wSrc[o >> 2], c = wSrc[(o >> 2) + 1];
            const int sh = 8 * (o & 3);
            return sh ? ((a >> sh) | (c << (32 - sh))) : a;
        };
        auto hash_at = [&](int p) -> u32 {                    // uniform p
            const int o = p - wbase;
            return (o >= 0 && o < LZW) ? (u32)sgpr((int)wHash[o]) : ((u32)sgpr((int)gKeys[p]) & HMASK);
        };
        while (pos < srcEnd) {
            KNZ_WAVE_ORDER();
            // ---- the next (up to 64) positions of a literal run, all at once
            const int stride = 1 + (skip >> 6);
            const bool fast = stride == 1;
            const int cnt = 64 - (skip & 63);                 // visits until the stride of the reference changes
            const int bp = pos + lane * stride;
            const bool act = (lane < cnt) && (bp < srcEnd);
            u32 bh = 0, w4 = 0, n4 = 0, inf = 0;
            int cd = 0;
            bool fromPrev = false;
            if (fast) {
                if (pos < wbase || pos + 66 > wbase + LZW) refill(pos);
                const int o = bp - wbase;                     // < LZW for every lane
                const u32 pv = wPrev[o];
                bh = wHash[o];
                inf = wInfo[o];
                w4 = win4(o);
                n4 = win4(o + 1);
                fromPrev = (pv == 0) || ((int)pv > maxGap);
                cd = (int)pv;
                if (act && !fromPrev) cd = tab_get(table, bh);
            } else {
                const u64 bw = act ? ld64u(src + bp) : 0ull;
                bh = lz_hash<HASH_LOG>(bw);
                w4 = (u32)bw;
                n4 = (u32)(bw >> 8);
                cd = act ? tab_get(table, bh) : 0;
            }
            const bool dups = lz_maybe_dups(seen, bh, act);
            if (!fast && dups) {
#pragma unroll 9
                for (int j = 0; j < 63; j++) {
                    const u32 hj = (u32)__builtin_amdgcn_readlane((int)bh, j);
                    const int aj = __builtin_amdgcn_readlane(act ? 1 : 0, j);
                    if (aj && lane > j && hj == bh) cd = pos + j * stride;
                }
            }
            bool hit = false;
            u64 x8 = 0, y8 = 0;
            u32 cw = 0;
            if (act) {
                const int blo = (bp - maxDist > 0) ? bp - maxDist : 0;
                const int rx = bp + 1 - rep0, ry = bp + 1 - rep1;
                x8 = ld64u(src + (rx > blo ? rx - 1 : 0));    // byte 0: the byte before the candidate, bytes 1-4: compared
                y8 = ld64u(src + (ry > blo ? ry - 1 : 0));
                bool hh;
                if (fromPrev) hh = (inf & 0x7FFFu) >= 4;
                else { cw = ld32u(src + (cd > blo ? cd : 0)); hh = cw == w4; }
                hit = (cd > blo && hh) || (rx > blo && (u32)(x8 >> 8) == n4) || (ry > blo && (u32)(y8 >> 8) == n4);
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
            if (f > 0) {
                skip += f; recent = 0; pos += f * stride;
                if (stride > 1) maxGap = pos - 1;
            }
            if (f == 64 || !((hitMask >> f) & 1)) continue;

            // ---- the serial step of the reference at `pos`
            const u32 w4f = (u32)__builtin_amdgcn_readlane((int)w4, f);
            const u32 nx4 = (u32)__builtin_amdgcn_readlane((int)n4, f);
            const u32 h0 = (u32)__builtin_amdgcn_readlane((int)bh, f);
            const int cand = __builtin_amdgcn_readlane(cd, f);
            const bool candPrev = __builtin_amdgcn_readlane(fromPrev ? 1 : 0, f) != 0;
            const u32 inf0 = (u32)__builtin_amdgcn_readlane((int)inf, f);
            const u32 c4 = (u32)__builtin_amdgcn_readlane((int)cw, f);
            const u64 x8f = readlane64(x8, f), y8f = readlane64(y8, f);
            if (lane == 0) tab_put(table, h0, pos);
            KNZ_WAVE_ORDER();
            const int nxt = pos + 1;
            const int lo = (pos - maxDist > 0) ? pos - maxDist : 0;
            const int refA = nxt - (recent ? rep1 : rep0);
            const int refB = nxt - (recent ? rep0 : rep1);
            const u64 A8 = recent ? y8f : x8f, B8 = recent ? x8f : y8f;
            int best = 0, ref = refA;
            u32 beforeRef = (u32)A8 & 0xFF;
            if (refA > lo && (u32)(A8 >> 8) == nx4) best = lz_match(src, nxt, refA, min(srcEnd - nxt, LZ_MAXMATCH), lane);
            else {
                ref = refB; beforeRef = (u32)B8 & 0xFF;
                if (refB > lo && (u32)(B8 >> 8) == nx4) best = lz_match(src, nxt, refB, min(srcEnd - nxt, LZ_MAXMATCH), lane);
            }
            if (best < LZ_MM) {
                ref = cand;
                if (cand > lo) {
                    const int lim0 = min(srcEnd - pos, LZ_MAXMATCH);
                    if (candPrev) {
                        const int l0 = (int)(inf0 & 0x7FFFu);
                        if (l0 >= 4) best = (!(inf0 >> 15) && l0 == LZ_LCAP && l0 + 8 <= lim0) ? lz_match(src, pos, cand, lim0, lane, LZ_LCAP) : l0;
                    } else if (c4 == w4f) best = lz_match(src, pos, cand, lim0, lane);
                }
                if (best < LZ_MM) {
                    if (skip >> 6) maxGap = nxt + (skip >> 6) - 1;
                    pos = nxt + (skip >> 6); skip++; recent = 0;
                    continue;
                }
                if (pos - ref != rep0 && pos - ref != rep1) {
                    // new distance: is the match one (LZX: two) position(s) further at least as long?
                    const int origin = pos;
                    for (int k = 1; k <= (EXTRA ? 2 : 1); k++) {
                        const int pk = origin + k;
                        const int ok_ = pk - wbase;
                        const bool inWin = fast && ok_ >= 0 && ok_ < LZW;
                        u32 hk, infk = 0;
                        int ck;
                        bool fromk = false;
                        if (inWin) {
                            const int pvk = sgpr((int)wPrev[ok_]);
                            hk = (u32)sgpr((int)wHash[ok_]);
                            infk = (u32)sgpr((int)wInfo[ok_]);
                            fromk = (pvk == 0) || (pvk > maxGap);
                            ck = fromk ? pvk : sgpr(tab_get(table, hk));
                        } else {
                            hk = lz_hash<HASH_LOG>(sgpr64(ld64u(src + pk)));
                            ck = sgpr(tab_get(table, hk));
                        }
                        if (lane == 0) tab_put(table, hk, pk);
                        KNZ_WAVE_ORDER();
                        if (ck <= lo + k) continue;
                        const int limk = min(srcEnd - pk, LZ_MAXMATCH);
                        const int lk = (int)(infk & 0x7FFFu);
                        const bool capped = !(infk >> 15) && lk == LZ_LCAP && lk + 8 <= limk;
                        int bk = -1;
                        if (fromk && (infk >> 15)) { if (lk >= best + 1) bk = lk; }
                        else if (fromk && lk >= best + 1) bk = capped ? lz_match(src, pk, ck, limk, lane, LZ_LCAP) : lk;
                        else if (sgpr((int)ld32u(src + pk + best - 3)) == sgpr((int)ld32u(src + ck + best - 3)))
                            bk = lz_match(src, pk, ck, limk, lane);
                        if (bk >= best) { ref = ck; best = bk; pos = pk; }
                    }
                }
                // extend backwards, 64 bytes per step
                for (;;) {
                    const bool c = (pos - lane > anchor) && (ref - lane > lo) && (ldg<u8>(src + (pos - 1 - lane)) == ldg<u8>(src + (ref - 1 - lane)));
                    const u64 fail = __ballot(!c);
                    const int k = fail ? __ffsll((long long)fail) - 1 : 64;
                    best += k; ref -= k; pos -= k;
                    if (k < 64) break;
                }
                if (best > LZ_MAXMATCH) { ref += best - LZ_MAXMATCH; pos += best - LZ_MAXMATCH; best = LZ_MAXMATCH; }
            } else {
                // repeat match at pos + 1: take the byte at pos with it when it matches too
                if (best >= LZ_MAXMATCH || (w4f & 0xFF) != beforeRef) {
                    pos++;
                    const u32 hp = hash_at(pos);
                    if (lane == 0) tab_put(table, hp, pos);
                    KNZ_WAVE_ORDER();
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
                const int o = p - wbase;
                u32 hp = 0;
                if (in) hp = (o >= 0 && o < LZW) ? wHash[o] : (gKeys[p] & HMASK);
                tab_put_wave(table, seen, hp, p, in, lane);
                KNZ_WAVE_ORDER();
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

// V5: the block layout of bitstream versions below 6 (LZCodec.cpp:614-760): same four sections, other token -- bits 3-0 match
// length - minMatch (14: + extension; 15: repeat distance, bit 4 picks the older one, the length is all extension), bit 4 otherwise:
// one distance byte more than the flag byte's bit 0 gives; minimum match { 4, 9, 6, 6 }[flag bits 2-1]; repeat distances start at 0.
template <bool V5>
__global__ __launch_bounds__(64) void k_lz_inverse(XfStage st, const u32* __restrict__ leftOnly)
{
    const int b = blockIdx.x;
    const int lane = lane_id();
    const int n = (int)st.len[b];
    if (n == 0) return;
    if (leftOnly != nullptr && leftOnly[4 * (size_t)b + 2] == 0) return;        // decoded by the data-parallel path below
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
        const int mm = V5 ? ((((flags >> 1) & 3) == 0) ? 4 : ((((flags >> 1) & 3) == 1) ? 9 : 6)) : ((flags >> 1) & 7) + 2;
        int s = 13, rep0 = V5 ? 0 : n, rep1 = V5 ? 0 : n;
        int settled = 0;                         // output bytes below this index are known to have left the wave
        LzByteWin wt, wm, wl, ws;
        wt.refill(src, t, n, lane); wm.refill(src, m, n, lane); wl.refill(src, l, n, lane); ws.refill(src, s, n, lane);
        ok = 1;
        for (;;) {
            const int token = (int)wt.at(src, t, n, lane);
            t++;
            int mlen, dist;
            if (V5) {
                mlen = token & 15;
                if (mlen == 15) {
                    mlen = mm + (int)lz_get_len(wl, src, l, n, lane);
                    dist = (token & 0x10) ? rep1 : rep0;
                } else {
                    mlen = (mlen == 14) ? 14 + mm + (int)lz_get_len(wl, src, l, n, lane) : mlen + mm;
                    const int nb = 1 + (flags & 1) + ((token >> 4) & 1);     // most significant first
                    dist = (int)wm.at(src, m, n, lane);
                    if (nb >= 2) dist = (dist << 8) | (int)wm.at(src, m + 1, n, lane);
                    if (nb == 3) dist = (dist << 8) | (int)wm.at(src, m + 2, n, lane);
                    m += nb;
                }
            } else if ((token & 0x18) == 0) {
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

// ------------------------------------------------------------------------------------------------
// Decoder in two parts (round 4).  The serial kernel above spends most of a token on the copy it makes (a load, a store and, when
// the next match reads what this one wrote, a fence: ~2.5 us per token).  Here the chain per block only PARSES: k_lz_i_parse walks the
// four byte sequences exactly as k_lz_inverse does, with the same checks in the same order, and leaves one record per token
// (output position, literal count, literal source, distance).  Everything that moves bytes is data parallel over the output:
//   k_lz_i_first  for every tile of 2048 output bytes the token its first byte belongs to
//   k_lz_i_expand      one entry per output byte: where the byte comes from -- a literal (offset in the block's literal section) or the
//                    output byte `distance` in front of it (dst[p] = dst[p - dist] holds for overlapping matches too: it is what the
//                    reference's byte loop, its 16-byte steps for dist >= 16 and its memset for dist == 1 all compute)
//   k_lz_i_jump        pointer jumping: an entry that points at a byte which itself points further back takes that byte's entry.
//                    Entries only ever move towards their literal, so the rounds run in place without a barrier between readers
//                    and writers; ceil(log2(longest chain)) + 1 rounds, tiles that are done drop out, and a round in which no tile
//                    is left costs a launch of workgroups that read one flag
//   k_lz_i_emit        dst[p] = literal section[entry]
// A distance of 0 (damaged input only) leaves the byte as it is, as the reference's byte loop does: such entries are roots of their own.
constexpr u32 LZI_T = 2048;                       // output bytes per tile
constexpr u32 LZI_ROOT = 0x80000000u;             // entry = LZI_ROOT | offset of a literal in src
constexpr u32 LZI_STALE = 0xC0000000u;            // entry = LZI_STALE | output position whose present content stays
constexpr int LZI_ROUNDS = 32;
constexpr u32 LZI_RECS = 1032;                    // records a tile can meet: every token but the last writes >= 2 bytes

struct LzInvWs {
    uint4* recs; size_t recStride;                // per block: (d, literal count, literal source, distance), then a sentinel (total, 0, 0, 0)
    u32* P; size_t pStride;                       // per block: one entry per output byte
    u32* tileFirst; u8* tileFlag; size_t tStride; // per block and tile
    u32* info;                                    // per block 4 words: records (with the sentinel), output bytes, 1 = left to the serial kernel
    u32* pending;                                 // LZI_ROUNDS + 1 words: tiles left after round r (index r + 1); [0] = tiles to start with
};

template <bool V5>
__global__ __launch_bounds__(64) void k_lz_i_parse(XfStage st, LzInvWs w)
{
    const int b = blockIdx.x;
    const int lane = lane_id();
    const int n = (int)st.len[b];
    u32* info = w.info + 4 * (size_t)b;
    if (n == 0 || info[3] == 0) return;                      // (k_lz_i_pparse has dealt with the block)
#ifdef KNZ_EMU
    if (lane == 0 && getenv("KNZ_EMU_VERBOSE")) fprintf(stderr, "k_lz_i_parse: block %d (n = %d) comes from the parallel parse\n", b, n);
#endif
    const u8* __restrict__ src = st.src[b];
    const int cap = (st.cap[b] > 0x7FFFFFFFu) ? 0x7FFFFFFF : (int)st.cap[b];
    if ((size_t)cap > w.pStride || (size_t)cap / 2 + 4 > w.recStride) { if (lane == 0) info[2] = 1; return; }   // (sized by the caller's bound: does not happen)
    uint4* recs = w.recs + (size_t)b * w.recStride;
    int ok = 0, d = 0;
    u32 nr = 0;
    do {
        if (n < 13) break;
        const int litEnd = sgpr((int)ld32u(src)), nTok = sgpr((int)ld32u(src + 4)), nDist = sgpr((int)ld32u(src + 8));
        if (litEnd < 0 || nTok < 0 || nDist < 0) break;
        if (litEnd < 13 || litEnd > n || nTok > n - litEnd || nDist > n - litEnd - nTok) break;
        int t = litEnd, m = litEnd + nTok, l = m + nDist;
        const int flags = sgpr((int)src[12]);
        const int maxDist = (flags & 1) ? LZ_MAXD2 : LZ_MAXD1;
        const int mm = V5 ? ((((flags >> 1) & 3) == 0) ? 4 : ((((flags >> 1) & 3) == 1) ? 9 : 6)) : ((flags >> 1) & 7) + 2;
        int s = 13, rep0 = V5 ? 0 : n, rep1 = V5 ? 0 : n;
        LzByteWin wt, wm, wl, ws;
        wt.refill(src, t, n, lane); wm.refill(src, m, n, lane); wl.refill(src, l, n, lane); ws.refill(src, s, n, lane);
        ok = 1;
        for (;;) {
            const int token = (int)wt.at(src, t, n, lane);
            t++;
            int mlen, dist;
            if (V5) {
                mlen = token & 15;
                if (mlen == 15) {
                    mlen = mm + (int)lz_get_len(wl, src, l, n, lane);
                    dist = (token & 0x10) ? rep1 : rep0;
                } else {
                    mlen = (mlen == 14) ? 14 + mm + (int)lz_get_len(wl, src, l, n, lane) : mlen + mm;
                    const int nb = 1 + (flags & 1) + ((token >> 4) & 1);
                    dist = (int)wm.at(src, m, n, lane);
                    if (nb >= 2) dist = (dist << 8) | (int)wm.at(src, m + 1, n, lane);
                    if (nb == 3) dist = (dist << 8) | (int)wm.at(src, m + 2, n, lane);
                    m += nb;
                }
            } else if ((token & 0x18) == 0) {
                mlen = token & 3;
                mlen = (mlen == 3) ? 3 + mm + (int)lz_get_len(wl, src, l, n, lane) : mlen + mm;
                dist = (token & 4) ? rep1 : rep0;
            } else {
                mlen = token & 7;
                mlen = (mlen == 7) ? 7 + mm + (int)lz_get_len(wl, src, l, n, lane) : mlen + mm;
                const int nb = (token >> 3) & 3;
                dist = (int)wm.at(src, m, n, lane);
                if (nb >= 2) dist = (dist << 8) | (int)wm.at(src, m + 1, n, lane);
                if (nb == 3) dist = (dist << 8) | (int)wm.at(src, m + 2, n, lane);
                m += nb;
            }
            int lit = 0, litSrc = s;
            const int d0 = d;
            if (token >= 32) {
                const u32 ul = (token >= 0xE0) ? 7u + lz_get_len(ws, src, s, n, lane) : (u32)(token >> 5);
                if (ul > (u32)(cap - d) || ul > (u32)(litEnd - s)) { ok = 0; break; }
                lit = (int)ul; litSrc = s;
                s += lit; d += lit;
                if (s >= litEnd - 13) {
                    if (lane == 0) recs[nr] = make_uint4((u32)d0, (u32)lit, (u32)litSrc, 1u);
                    nr++;
                    break;
                }
            }
            rep1 = rep0; rep0 = dist;
            const int end = d + mlen;
            const int ref = d - dist;
            if (ref < 0 || dist > maxDist || end > cap) { ok = 0; break; }
            if (lane == 0) recs[nr] = make_uint4((u32)d0, (u32)lit, (u32)litSrc, (u32)dist);
            nr++;
            d = end;
        }
        ok = ok && (s == litEnd);
    } while (0);
    if (lane == 0) {
        st.ok[b] = (u8)(ok ? 1 : 0); st.newLen[b] = (u32)d;
        if (ok) { recs[nr] = make_uint4((u32)d, 0u, 0u, 1u); info[0] = nr + 1; info[1] = (u32)d; }
    }
}

// ---- the parse without the chain (current block layout only) ---------------------------------------------------------------
// What makes the parse serial is only WHERE each token's pieces sit: its distance bytes, its length extension, its literals. Three
// of the four positions are prefix sums over token bytes (distance bytes: 0-3 per token; which tokens have a length extension, which
// a literal extension; the literal counts below 7). The length extensions are self-delimiting (1, 3 or 4 bytes by the first byte):
// their starts are the states "0" of a 4-state machine run over the section's bytes, and a machine with 4 states is a scan over
// functions {0..3} -> {0..3}. The distances a token repeats are the previous token's or the one before: (rep0, rep1) after a token is
// a function of (rep0, rep1) before it whose two outputs are a constant or one of the inputs -- closed under composition, so a scan
// again. Only the literal extensions stay a chain (the extension bytes sit between the literals they count), but a short one: one
// step per literal run of 7 bytes or more (1 token in 100 to 400). One workgroup of 1024 threads per block streams over the
// sections with carries; anything out of the ordinary (a header that fails the checks, a stream the serial parse would stop early
// or late on, a bound exceeded) is left to k_lz_i_parse, which then decides with the reference's own sequence of checks.
struct LzPWs {
    u32* extM; size_t extStride;              // per block: value of the e-th match-length extension
    uint2* extL;                              // per block: e-th literal extension: (token, literals below 7 in front of it), later (first literal, count)
    u32* xcum;                                // per block: bytes the first e literal extensions and their runs take
};

template <typename T>
__device__ __forceinline__ T lzp_shfl_up(T v, int off)
{
    static_assert(sizeof(T) % 4 == 0, "");
    u32 wds[sizeof(T) / 4];
    __builtin_memcpy(wds, &v, sizeof(T));
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; i++) wds[i] = (u32)__shfl_up((int)wds[i], (unsigned)off);
    T r;
    __builtin_memcpy(&r, wds, sizeof(T));
    return r;
}

// exclusive scan over the 1024 threads of the workgroup with a (non-commutative) combine(earlier, later); sm: 16 entries
template <typename T, typename Op>
__device__ __forceinline__ T lzp_scan(T v, const T ident, Op op, T* sm, T& total)
{
    const int lane = lane_id(), wv = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T o = lzp_shfl_up(v, off);
        if (lane >= off) v = op(o, v);
    }
    T ex = lzp_shfl_up(v, 1);
    if (lane == 0) ex = ident;
    if (lane == 63) sm[wv] = v;
    __syncthreads();
    T pre = ident, all = ident;
    for (int i = 0; i < 16; i++) { if (i == wv) pre = all; all = op(all, sm[i]); }
    __syncthreads();
    total = all;
    return op(pre, ex);
}

struct LzFsm { u32 fn; u32 c01, c23; };       // fn: 2 bits per incoming state; starts seen per incoming state (16 bits each)
__device__ __forceinline__ u32 lzfsm_cnt(const LzFsm& f, u32 s) { return ((s & 2 ? f.c23 : f.c01) >> (16 * (s & 1))) & 0xFFFFu; }
struct LzFsmOp {
    __device__ __forceinline__ LzFsm operator()(const LzFsm& a, const LzFsm& b) const
    {
        LzFsm r; r.fn = 0;
        u32 c[4];
#pragma unroll
        for (u32 s = 0; s < 4; s++) {
            const u32 mid = (a.fn >> (2 * s)) & 3;
            r.fn |= ((b.fn >> (2 * mid)) & 3) << (2 * s);
            c[s] = lzfsm_cnt(a, s) + lzfsm_cnt(b, mid);
        }
        r.c01 = c[0] | (c[1] << 16); r.c23 = c[2] | (c[3] << 16);
        return r;
    }
};
struct LzRep { u32 a, b; };                   // (rep0, rep1) after, each: kind << 30 | value; kind 0 constant, 1 = rep0 before, 2 = rep1 before
struct LzRepOp {
    __device__ __forceinline__ LzRep operator()(const LzRep& f, const LzRep& g) const
    {
        LzRep r;
        r.a = (g.a >> 30) == 0 ? g.a : ((g.a >> 30) == 1 ? f.a : f.b);
        r.b = (g.b >> 30) == 0 ? g.b : ((g.b >> 30) == 1 ? f.a : f.b);
        return r;
    }
};
struct LzSum64 { __device__ __forceinline__ u64 operator()(u64 a, u64 b) const { return a + b; } };
struct LzU64 { u32 lo, hi; };

__device__ __forceinline__ u32 lzp_byte(const u8* s, int x, int limit) { return (x >= 0 && x < limit) ? (u32)s[x] : 0u; }
// readLength at x (bytes at or behind `limit` read as zero): value, size
__device__ __forceinline__ u32 lzp_len_at(const u8* s, int x, int limit, u32& size)
{
    const u32 b0 = lzp_byte(s, x, limit);
    if (b0 < 254) { size = 1; return b0; }
    const u32 b1 = lzp_byte(s, x + 1, limit), b2 = lzp_byte(s, x + 2, limit);
    if (b0 == 254) { size = 3; return 254 + ((b1 << 8) | b2); }
    size = 4;
    return 255 + ((b1 << 16) | (b2 << 8) | lzp_byte(s, x + 3, limit));
}

// the four counts of a token, 16 bits each: distance bytes | length extension << 16 | literal extension << 32 | literals below 7 << 48
__device__ __forceinline__ u64 lzp_counts(u32 tok)
{
    const bool rep = (tok & 0x18) == 0;
    const u64 nb = rep ? 0 : (tok >> 3) & 3;
    const u64 me = rep ? ((tok & 3) == 3) : ((tok & 7) == 7);
    const u64 le = tok >= 0xE0;
    const u64 ls = (tok >= 32 && tok < 0xE0) ? (tok >> 5) : 0;
    return nb | (me << 16) | (le << 32) | (ls << 48);
}

__global__ __launch_bounds__(1024) void k_lz_i_pparse(XfStage st, LzInvWs w, LzPWs pw, int oldLayout)
{
    const int b = blockIdx.x;
    const int tid = (int)threadIdx.x;
    const int lane = lane_id();
    const int n = (int)st.len[b];
    u32* info = w.info + 4 * (size_t)b;
    __shared__ u64 smSum[16];
    __shared__ LzFsm smFsm[16];
    __shared__ LzRep smRep[16];
    __shared__ u32 sBad, sExtM;
    if (tid == 0) { info[0] = 0; info[1] = 0; info[2] = 0; info[3] = 0; sBad = 0; }
    if (n == 0) return;
    __syncthreads();
    const u8* __restrict__ src = st.src[b];
    const int cap = (st.cap[b] > 0x7FFFFFFFu) ? 0x7FFFFFFF : (int)st.cap[b];
    bool plain = !oldLayout && n >= 13 && n < (1 << 30) && (size_t)cap <= w.pStride && (size_t)cap / 2 + 4 <= w.recStride;
    int litEnd = 0, nTok = 0, nDist = 0, flags = 0;
    if (plain) {
        litEnd = (int)ld32u(src); nTok = (int)ld32u(src + 4); nDist = (int)ld32u(src + 8); flags = (int)src[12];
        if (litEnd < 13 || nTok < 1 || nDist < 0 || litEnd > n || nTok > n - litEnd || nDist > n - litEnd - nTok) plain = false;
        if (plain && (size_t)nTok + 2 > w.recStride) plain = false;      // (more tokens than the output has room for: nothing that decodes)
    }
    if (!plain) { if (tid == 0) info[3] = 1; return; }
    const int t0 = litEnd, m0 = litEnd + nTok, l0 = m0 + nDist;
    const int maxDist = (flags & 1) ? LZ_MAXD2 : LZ_MAXD1;
    const int mm = ((flags >> 1) & 7) + 2;
    u32* extM = pw.extM + (size_t)b * pw.extStride;
    uint2* extL = pw.extL + (size_t)b * pw.extStride;
    u32* xcum = pw.xcum + (size_t)b * pw.extStride;
    uint4* recs = w.recs + (size_t)b * w.recStride;

    // ---- 1. match-length extensions: where they start, what they say
    {
        u32 state = 0, ordinal = 0;
        for (int base = l0; base < n; base += 8192) {
            const int x0 = base + 8 * tid;
            u32 by[8];
#pragma unroll
            for (int j = 0; j < 8; j++) by[j] = lzp_byte(src, x0 + j, n);
            LzFsm f; f.fn = 0;
            u32 c[4];
#pragma unroll
            for (u32 s0 = 0; s0 < 4; s0++) {
                u32 s = s0, cnt = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (x0 + j < n) {
                        if (s == 0) { cnt++; s = by[j] < 254 ? 0 : (by[j] == 254 ? 2 : 3); }
                        else s--;
                    }
                }
                f.fn |= s << (2 * s0); c[s0] = cnt;
            }
            f.c01 = c[0] | (c[1] << 16); f.c23 = c[2] | (c[3] << 16);
            LzFsm ident; ident.fn = 0xE4; ident.c01 = 0; ident.c23 = 0;
            LzFsm tot;
            const LzFsm pre = lzp_scan(f, ident, LzFsmOp(), smFsm, tot);
            u32 s = (pre.fn >> (2 * state)) & 3;
            u32 e = ordinal + lzfsm_cnt(pre, state);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (x0 + j < n) {
                    if (s == 0) {
                        u32 size;
                        const u32 v = lzp_len_at(src, x0 + j, n, size);
                        if (e < (u32)nTok) extM[e] = v;
                        e++; s = size - 1;
                    } else s--;
                }
            }
            ordinal += lzfsm_cnt(tot, state);
            state = (tot.fn >> (2 * state)) & 3;
        }
        if (tid == 0) sExtM = ordinal - (state != 0 ? 1u : 0u);         // extensions that lie inside the block with all their bytes
    }
    __syncthreads();
    const u32 extMComplete = sExtM;

    // ---- 2. tokens, first pass: the literal extensions in order -- (token, literals of the short runs in front of it)
    u32 cNb = 0, cMe = 0, cLe = 0, cLs = 0;
    for (int base = 0; base < nTok; base += 4096) {
        const int k0 = base + 4 * tid;
        u64 c[4], mine = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { c[j] = (k0 + j < nTok) ? lzp_counts(src[t0 + k0 + j]) : 0; mine += c[j]; }
        u64 tot;
        u64 pre = lzp_scan(mine, (u64)0, LzSum64(), smSum, tot);        // (16-bit fields: a tile's sums stay below 65536)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if ((c[j] >> 32) & 1) extL[cLe + (u32)((pre >> 32) & 0xFFFFu)] = make_uint2((u32)(k0 + j), cLs + (u32)(pre >> 48));
            pre += c[j];
        }
        cNb += (u32)(tot & 0xFFFFu); cMe += (u32)((tot >> 16) & 0xFFFFu); cLe += (u32)((tot >> 32) & 0xFFFFu); cLs += (u32)(tot >> 48);
    }
    const u32 leTot = cLe;
    if (cNb > (u32)(n - m0) || cMe > extMComplete) { if (tid == 0) info[3] = 1; return; }   // pieces beyond the block's end (uniform)
    // the literal section into this XCD's L2 before one wave walks through it with dependent loads
    {
        u32 sink = 0;
        for (int off = 64 * tid; off < litEnd; off += 65536) sink += src[off];
        if (sink == 0xFFFFFFFFu) sBad = 2;                              // (never: keeps the loads)
    }
    __syncthreads();

    // ---- 3. the chain that is left: one step per literal run of 7 or more
    if (tid < 64) {
        u32 X = 0;
        bool bad = false;
        for (u32 i0 = 0; i0 < leTot && !bad; i0 += 64) {
            const uint2 mine = (i0 + (u32)lane < leTot) ? extL[i0 + (u32)lane] : make_uint2(0u, 0u);
            const u32 cnt = (leTot - i0 < 64u) ? leTot - i0 : 64u;
            uint2 res = make_uint2(0u, 0u);
            u32 resX = 0;
            for (u32 j = 0; j < cnt; j++) {
                const u32 S = (u32)__builtin_amdgcn_readlane((int)mine.y, (int)j);
                const u32 sp = 13u + S + X;
                if (sp >= (u32)litEnd) { bad = true; break; }
                u32 size;
                const u32 lit = 7u + lzp_len_at(src, sgpr((int)sp), n, size);
                const u32 sa = sp + size;
                if (sa > (u32)litEnd || lit > (u32)litEnd - sa) { bad = true; break; }
                if ((u32)lane == j) { res = make_uint2(sa, lit); resX = X; }
                X += size + lit;
            }
            if (i0 + (u32)lane < leTot) { extL[i0 + (u32)lane] = res; xcum[i0 + (u32)lane] = resX; }
        }
        if (lane == 0) { xcum[leTot] = X; if (bad) sBad = 1; }
    }
    __syncthreads();
    if (sBad) { if (tid == 0) info[3] = 1; return; }

    // ---- 4. tokens, second pass: lengths, output positions, distances; the checks of the serial parse
    cNb = 0; cMe = 0; cLe = 0; cLs = 0;
    u64 cD = 0;
    u32 cRep0 = (u32)n, cRep1 = (u32)n;
    bool bad = false;
    for (int base = 0; base < nTok; base += 4096) {
        const int k0 = base + 4 * tid;
        u32 tk[4];
        u64 c[4], mine = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { tk[j] = (k0 + j < nTok) ? (u32)src[t0 + k0 + j] : 0x100u; c[j] = (tk[j] < 0x100u) ? lzp_counts(tk[j]) : 0; mine += c[j]; }
        u64 tot;
        u64 pre = lzp_scan(mine, (u64)0, LzSum64(), smSum, tot);
        u32 lit[4], lsrc[4], mlen[4], dsp[4];                           // dsp: the distance as kind << 30 | value
        u64 lenSum = 0;
        LzRep F; F.a = 1u << 30; F.b = 2u << 30;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            lit[j] = 0; lsrc[j] = 0; mlen[j] = 0; dsp[j] = 1u << 30;
            if (tk[j] < 0x100u) {
                const u32 M = cNb + (u32)(pre & 0xFFFFu), E = cMe + (u32)((pre >> 16) & 0xFFFFu), L = cLe + (u32)((pre >> 32) & 0xFFFFu), S = cLs + (u32)(pre >> 48);
                const u32 t = tk[j];
                if (t >= 0xE0) { const uint2 e = extL[L]; lsrc[j] = e.x; lit[j] = e.y; }
                else { lsrc[j] = 13u + S + xcum[L]; lit[j] = (t >= 32) ? (t >> 5) : 0u; }
                const bool rep = (t & 0x18) == 0;
                const u32 mb = rep ? (t & 3) : (t & 7);
                const bool me = rep ? (mb == 3) : (mb == 7);
                mlen[j] = mb + (u32)mm + (me ? extM[E] : 0u);
                if (rep) dsp[j] = (t & 4) ? (2u << 30) : (1u << 30);
                else {
                    const u32 nb = (t >> 3) & 3;
                    u32 dv = src[m0 + M];
                    if (nb >= 2) dv = (dv << 8) | src[m0 + M + 1];
                    if (nb == 3) dv = (dv << 8) | src[m0 + M + 2];
                    dsp[j] = dv;
                }
                if (k0 + j == nTok - 1) mlen[j] = 0;                    // the last token ends with its literals
                lenSum += (u64)lit[j] + mlen[j];
                LzRep g; g.a = dsp[j]; g.b = 1u << 30;                   // rep0 = this distance, rep1 = the rep0 before
                F = LzRepOp()(F, g);
                pre += c[j];
            }
        }
        u64 dTot;
        u64 d = cD + lzp_scan(lenSum, (u64)0, LzSum64(), smSum, dTot);
        LzRep ident; ident.a = 1u << 30; ident.b = 2u << 30;
        LzRep repTot;
        const LzRep rp = lzp_scan(F, ident, LzRepOp(), smRep, repTot);
        LzRep cst; cst.a = cRep0; cst.b = cRep1;                        // (n < 2^30: constants)
        const LzRep in = LzRepOp()(cst, rp);
        u32 rep0 = in.a, rep1 = in.b;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (tk[j] < 0x100u) {
                const u32 t = tk[j];
                const bool last = (k0 + j == nTok - 1);
                const u32 dist = (dsp[j] >> 30) == 0 ? dsp[j] : ((dsp[j] >> 30) == 1 ? rep0 : rep1);
                const u64 dl = d + lit[j];
                if (t >= 32) {
                    if (dl > (u64)cap) bad = true;
                    const u32 sAfter = lsrc[j] + lit[j];
                    if (last ? (sAfter != (u32)litEnd) : (sAfter >= (u32)litEnd - 13u)) bad = true;
                } else if (last) bad = true;
                if (!last) {
                    if (dl < dist || dist > (u32)maxDist || dl + mlen[j] > (u64)cap) bad = true;
                    rep1 = rep0; rep0 = dist;
                }
                recs[k0 + j] = make_uint4((u32)d, lit[j], lsrc[j], last ? 1u : dist);
                d = dl + mlen[j];
            }
        }
        cNb += (u32)(tot & 0xFFFFu); cMe += (u32)((tot >> 16) & 0xFFFFu); cLe += (u32)((tot >> 32) & 0xFFFFu); cLs += (u32)(tot >> 48);
        cD += dTot;
        const LzRep after = LzRepOp()(cst, repTot);
        cRep0 = after.a; cRep1 = after.b;
    }
    if (bad) sBad = 1;
    __syncthreads();
    if (tid == 0) {
        if (sBad || cD > (u64)cap) info[3] = 1;
        else { recs[nTok] = make_uint4((u32)cD, 0u, 0u, 1u); info[0] = (u32)nTok + 1u; info[1] = (u32)cD; st.ok[b] = 1; st.newLen[b] = (u32)cD; }
    }
}

__global__ __launch_bounds__(256) void k_lz_i_first(LzInvWs w, int wgPerBlock)
{
    const int b = blockIdx.x / wgPerBlock;
    const u32 part = blockIdx.x - (u32)b * (u32)wgPerBlock;
    const u32* info = w.info + 4 * (size_t)b;
    const u32 nr = info[0];
    if (nr < 2) return;
    const uint4* recs = w.recs + (size_t)b * w.recStride;
    u32* tf = w.tileFirst + (size_t)b * w.tStride;
    for (u32 k = part * 256 + threadIdx.x; k + 1 < nr; k += (u32)wgPerBlock * 256) {
        const u32 d0 = recs[k].x, d1 = recs[k + 1].x;
        for (u32 t = (d0 + LZI_T - 1) / LZI_T; (u64)t * LZI_T < d1; t++) tf[t] = k;
    }
}

__global__ __launch_bounds__(256) void k_lz_i_expand(LzInvWs w, int tilesPerBlock)
{
    const int b = blockIdx.x / tilesPerBlock;
    const u32 tile = blockIdx.x - (u32)b * (u32)tilesPerBlock;
    const u32* info = w.info + 4 * (size_t)b;
    const u32 nr = info[0], total = info[1];
    const u32 p0 = tile * LZI_T;
    if (nr < 2 || p0 >= total) return;
    const uint4* recs = w.recs + (size_t)b * w.recStride;
    __shared__ uint4 R[LZI_RECS];
    __shared__ u32 sDone[LZI_RECS / 256 + 3];                // one flag per batch of records (no flag is written again after it was read), one for the result
    const u32 k0 = w.tileFirst[(size_t)b * w.tStride + tile];
    if (threadIdx.x < LZI_RECS / 256 + 3) sDone[threadIdx.x] = 0;
    __syncthreads();
    const u32 pEnd = (p0 + LZI_T < total) ? p0 + LZI_T : total;
    // records k0 .. up to the first one that starts at or behind the tile's end (the sentinel starts at `total`)
    u32 nLoaded = 0;
    for (u32 base = 0; base < LZI_RECS; base += 256) {
        const u32 idx = base + threadIdx.x;
        if (idx < LZI_RECS) {
            uint4 r = make_uint4(0xFFFFFFFFu, 0u, 0u, 1u);
            if (k0 + idx < nr) r = recs[k0 + idx];
            R[idx] = r;
            if (r.x >= pEnd) sDone[base / 256] = 1;
        }
        nLoaded = (base + 256 < LZI_RECS) ? base + 256 : LZI_RECS;
        __syncthreads();
        if (sDone[base / 256]) break;
    }
    // thread: 8 consecutive bytes
    const u32 p = p0 + 8 * threadIdx.x;
    u32 out[8];
    bool pend = false;
    if (p < pEnd) {
        u32 lo = 0, hi = nLoaded - 1;                           // largest i with R[i].x <= p (R[0].x <= p0)
        while (lo < hi) { const u32 mid = (lo + hi + 1) >> 1; if (R[mid].x <= p) lo = mid; else hi = mid - 1; }
        u32 i = lo;
        uint4 r = R[i];
        u32 nextD = (i + 1 < nLoaded) ? R[i + 1].x : 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u32 q = p + (u32)j;
            while (q >= nextD) { i++; r = R[i]; nextD = (i + 1 < nLoaded) ? R[i + 1].x : 0xFFFFFFFFu; }
            u32 e;
            if (q < r.x + r.y) e = LZI_ROOT | (r.z + (q - r.x));
            else if (r.w == 0) e = LZI_STALE | q;
            else { e = q - r.w; pend = true; }
            if (q >= pEnd) { e = LZI_STALE | q; }
            out[j] = e;
        }
        u32* P = w.P + (size_t)b * w.pStride + p;
        *reinterpret_cast<uint4*>(P) = make_uint4(out[0], out[1], out[2], out[3]);
        *reinterpret_cast<uint4*>(P + 4) = make_uint4(out[4], out[5], out[6], out[7]);
    }
    if (pend) sDone[LZI_RECS / 256 + 2] = 1;
    __syncthreads();
    const bool any = sDone[LZI_RECS / 256 + 2] != 0;
    if (threadIdx.x == 0) {
        w.tileFlag[(size_t)b * w.tStride + tile] = any ? 1 : 0;
        if (any) atomicAdd(&w.pending[0], 1u);
    }
}

__global__ __launch_bounds__(256) void k_lz_i_jump(LzInvWs w, int tilesPerBlock, int round)
{
    if (w.pending[round] == 0) return;                       // nothing was left after the round before
    const int b = blockIdx.x / tilesPerBlock;
    const u32 tile = blockIdx.x - (u32)b * (u32)tilesPerBlock;
    u8* flag = w.tileFlag + (size_t)b * w.tStride + tile;
    const u32 total = w.info[4 * (size_t)b + 1];
    if ((u64)tile * LZI_T >= total || *flag == 0) return;
    u32* Pb = w.P + (size_t)b * w.pStride;
    u32* P = Pb + (size_t)tile * LZI_T + 8 * threadIdx.x;
    bool pend = false;
    if (tile * LZI_T + 8 * threadIdx.x < total) {              // (entries behind the last group of 8 were never written)
        uint4 a = *reinterpret_cast<const uint4*>(P), c = *reinterpret_cast<const uint4*>(P + 4);
        u32 e[8] = { a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w };
        bool changed = false;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (!(e[j] & LZI_ROOT)) {
                u32 v = Pb[e[j]];                                   // up to three hops a round (the loads of a thread's eight
                if (!(v & LZI_ROOT)) v = Pb[v];                     // entries overlap): the rounds go down from log2 to log4
                if (!(v & LZI_ROOT)) v = Pb[v];                     // of the longest chain and more
                e[j] = v; changed = true;
                if (!(v & LZI_ROOT)) pend = true;
            }
        }
        if (changed) {
            *reinterpret_cast<uint4*>(P) = make_uint4(e[0], e[1], e[2], e[3]);
            *reinterpret_cast<uint4*>(P + 4) = make_uint4(e[4], e[5], e[6], e[7]);
        }
    }
    __shared__ u32 sPend;
    if (threadIdx.x == 0) sPend = 0;
    __syncthreads();
    if (pend) sPend = 1;
    __syncthreads();
    const bool any = sPend != 0;
    if (threadIdx.x == 0) {
        *flag = any ? 1 : 0;
        if (any) atomicAdd(&w.pending[round + 1], 1u);
    }
}

__global__ __launch_bounds__(256) void k_lz_i_emit(XfStage st, LzInvWs w, int tilesPerBlock)
{
    const int b = blockIdx.x / tilesPerBlock;
    const u32 tile = blockIdx.x - (u32)b * (u32)tilesPerBlock;
    const u32 total = w.info[4 * (size_t)b + 1];
    const u32 p = tile * LZI_T + 8 * threadIdx.x;
    if (p >= total) return;
    const u8* __restrict__ src = st.src[b];
    u8* dst = st.dst[b];
    const u32* P = w.P + (size_t)b * w.pStride + p;
    const uint4 a = *reinterpret_cast<const uint4*>(P), c = *reinterpret_cast<const uint4*>(P + 4);
    const u32 e[8] = { a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w };
    u64 v = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const u32 x = e[j];
        u8 by;
        if ((x & LZI_STALE) == LZI_ROOT) by = src[x & 0x3FFFFFFFu];
        else if ((x & LZI_STALE) == LZI_STALE) by = dst[x & 0x3FFFFFFFu];
        else by = dst[p + j];                                 // (not reached: every chain ends within LZI_ROUNDS rounds)
        v |= (u64)by << (8 * j);
    }
    if (p + 8 <= total && (reinterpret_cast<uintptr_t>(dst) & 7) == 0) *reinterpret_cast<u64*>(dst + p) = v;
    else for (int j = 0; j < 8 && p + j < total; j++) dst[p + j] = (u8)(v >> (8 * j));
}

static int lz_group_blocks(int ttype, int nBlocks)
{
    const int hl = (ttype == KNZ_T_LZX) ? 19 : 16;
    const int most = (1 << (32 - hl)) - 2;            // (block id + 1) << hl has to fit the 32-bit sort key
    return nBlocks < most ? nBlocks : most;
}

static size_t lz_align(size_t v) { return (v + 255) & ~(size_t)255; }

size_t lz_forward_scratch_bytes(int ttype, int nBlocks, u32 maxLen)
{
    const int hl = (ttype == KNZ_T_LZX) ? 19 : 16;
    const int G = lz_group_blocks(ttype, nBlocks);
    const size_t total = (size_t)G * maxLen;
    const size_t sec = ((size_t)maxLen + 64 + 15) & ~(size_t)15;
    return lz_align(((size_t)G << hl) * 4) + lz_align((size_t)G * 3 * sec) + 7 * lz_align(total * 4) + lz_align(total * 2) + lz_align(prims::rs_ws_bytes(total, 1)) + 512;
}

// Returns 0 or a negative value on a HIP error.
int launch_lz_forward(hipStream_t s, const XfStage& stAll, int ttype, void* scratch, size_t scratchBytes)
{
    const int hl = (ttype == KNZ_T_LZX) ? 19 : 16;
    const int G = lz_group_blocks(ttype, stAll.nBlocks);
    const u32 S = stAll.maxLen;
    const size_t totalMax = (size_t)G * S;
    const size_t sec = ((size_t)S + 64 + 15) & ~(size_t)15;
    u8* p = reinterpret_cast<u8*>(scratch);
    LzWs ws;
    ws.tables = reinterpret_cast<int*>(p); p += lz_align(((size_t)G << hl) * 4);
    ws.side = p; p += lz_align((size_t)G * 3 * sec);
    ws.secStride = sec;
    ws.keysA = reinterpret_cast<u32*>(p); p += lz_align(totalMax * 4);
    ws.keysB = reinterpret_cast<u32*>(p); p += lz_align(totalMax * 4);
    ws.valsA = reinterpret_cast<u32*>(p); p += lz_align(totalMax * 4);
    ws.valsB = reinterpret_cast<u32*>(p); p += lz_align(totalMax * 4);
    ws.prev = reinterpret_cast<u32*>(p); p += lz_align(totalMax * 4);
    ws.info = reinterpret_cast<u16*>(p); p += lz_align(totalMax * 2);
    u32* keysC = reinterpret_cast<u32*>(p); p += lz_align(totalMax * 4);
    u32* valsC = reinterpret_cast<u32*>(p); p += lz_align(totalMax * 4);
    u32* seg2 = reinterpret_cast<u32*>(p); p += 256;
    if ((size_t)(p - reinterpret_cast<u8*>(scratch)) + prims::rs_ws_bytes(totalMax, 1) > scratchBytes) return -2;
    const prims::RsWs rs = prims::rs_carve(p, totalMax, 1, seg2, 1);
    ws.S = S;
    for (int g0 = 0; g0 < stAll.nBlocks; g0 += G) {
        XfStage st = stAll;
        st.src += g0; st.dst += g0; st.len += g0; st.cap += g0; st.ok += g0; st.newLen += g0;
        st.nBlocks = (stAll.nBlocks - g0 < G) ? stAll.nBlocks - g0 : G;
        const size_t total = (size_t)st.nBlocks * S;
        int bbits = 1;
        while ((1 << bbits) < st.nBlocks + 1) bbits++;
        const dim3 grid((unsigned)((total + 255) / 256));
        if (hipMemsetAsync(ws.tables, 0, ((size_t)st.nBlocks << hl) * sizeof(int), s) != hipSuccess) return -1;
        // (hash, position) pairs of the group's blocks in hash order, equal hashes in position order: one stable sort
        auto sortPairs = [&](int bits) -> int {
            KScope ks_("k_lz_sort");
            hipLaunchKernelGGL(prims::k_rs_one_segment, dim3(1), dim3(64), 0, s, seg2, (u32)total);
            prims::rs_launch_layout(s, rs);
            return prims::rs_sort_keep<u32, true>(s, rs, ws.keysA, ws.valsA, ws.keysB, keysC, ws.valsB, valsC, total, 0, bits);
        };
        if (ttype == KNZ_T_LZX) {
            { KScope ks_("k_lz_keys"); hipLaunchKernelGGL((k_lz_keys<19>), grid, dim3(256), 0, s, st, S, st.nBlocks, ws.keysA, ws.valsA); }
            const int r = sortPairs(19 + bbits);
            { KScope ks_("k_lz_prev"); hipLaunchKernelGGL((k_lz_prev<19>), grid, dim3(256), 0, s, st, S, st.nBlocks, total, r ? keysC : ws.keysB, r ? valsC : ws.valsB, ws.prev, ws.info); }
            { KScope ks_("k_lz_walk"); hipLaunchKernelGGL((k_lz_walk<19, true>), dim3(st.nBlocks), dim3(64), 0, s, st, ws); }
        } else {
            { KScope ks_("k_lz_keys"); hipLaunchKernelGGL((k_lz_keys<16>), grid, dim3(256), 0, s, st, S, st.nBlocks, ws.keysA, ws.valsA); }
            const int r = sortPairs(16 + bbits);
            { KScope ks_("k_lz_prev"); hipLaunchKernelGGL((k_lz_prev<16>), grid, dim3(256), 0, s, st, S, st.nBlocks, total, r ? keysC : ws.keysB, r ? valsC : ws.valsB, ws.prev, ws.info); }
            { KScope ks_("k_lz_walk"); hipLaunchKernelGGL((k_lz_walk<16, false>), dim3(st.nBlocks), dim3(64), 0, s, st, ws); }
        }
    }
    return 0;
}

int lz_serial_decode(int set)
{
    static std::atomic<int> v{ getenv("KNZ_LZ_SERIAL_DECODE") ? atoi(getenv("KNZ_LZ_SERIAL_DECODE")) : 0 };
    if (set >= 0) v.store(set);
    return v.load();
}

static size_t lzi_tiles(u32 maxCap) { return ((size_t)maxCap + LZI_T - 1) / LZI_T + 1; }

size_t lz_inverse_scratch_bytes(int nBlocks, u32 maxCap)
{
    const size_t tiles = lzi_tiles(maxCap);
    return (size_t)nBlocks * (2 * lz_align(((size_t)maxCap / 2 + 8) * sizeof(uint4)) + lz_align(tiles * LZI_T * 4) + lz_align(tiles * 4) + lz_align(tiles)) +
           lz_align((size_t)nBlocks * 16) + lz_align((LZI_ROUNDS + 2) * 4) + 512;
}

// scratch == nullptr (or maxCap == 0): the one-wave-per-block decoder alone
void launch_lz_inverse(hipStream_t s, const XfStage& st, void* scratch, size_t scratchBytes, u32 maxCap)
{
    if (scratch == nullptr || maxCap == 0 || maxCap > (1u << 30) || scratchBytes < lz_inverse_scratch_bytes(st.nBlocks, maxCap)) {
        KScope ks_("k_lz_inverse");
        if (st.bsVersion < 6) hipLaunchKernelGGL(k_lz_inverse<true>, dim3(st.nBlocks), dim3(64), 0, s, st, (const u32*)nullptr);
        else hipLaunchKernelGGL(k_lz_inverse<false>, dim3(st.nBlocks), dim3(64), 0, s, st, (const u32*)nullptr);
        return;
    }
    const size_t tiles = lzi_tiles(maxCap);
    u8* p = reinterpret_cast<u8*>(scratch);
    LzInvWs w;
    w.recStride = (size_t)maxCap / 2 + 8;
    w.recs = reinterpret_cast<uint4*>(p); p += (size_t)st.nBlocks * lz_align(w.recStride * sizeof(uint4));
    w.recStride = lz_align(w.recStride * sizeof(uint4)) / sizeof(uint4);
    LzPWs pw;                                                   // 16 bytes per possible token again: extension values, literal extensions, their sums
    pw.extStride = w.recStride;
    pw.extL = reinterpret_cast<uint2*>(p); p += (size_t)st.nBlocks * pw.extStride * 8;
    pw.extM = reinterpret_cast<u32*>(p); p += (size_t)st.nBlocks * pw.extStride * 4;
    pw.xcum = reinterpret_cast<u32*>(p); p += (size_t)st.nBlocks * pw.extStride * 4;
    w.pStride = lz_align(tiles * LZI_T * 4) / 4;
    w.P = reinterpret_cast<u32*>(p); p += (size_t)st.nBlocks * w.pStride * 4;
    w.tStride = lz_align(tiles * 4) / 4;
    w.tileFirst = reinterpret_cast<u32*>(p); p += (size_t)st.nBlocks * w.tStride * 4;
    w.tileFlag = p; p += lz_align((size_t)st.nBlocks * w.tStride);
    w.info = reinterpret_cast<u32*>(p); p += lz_align((size_t)st.nBlocks * 16);
    w.pending = reinterpret_cast<u32*>(p);
    const int tilesPerBlock = (int)(tiles - 1);
    hipMemsetAsync(w.pending, 0, (LZI_ROUNDS + 2) * 4, s);
    { KScope ks_("k_lz_i_pparse"); hipLaunchKernelGGL(k_lz_i_pparse, dim3(st.nBlocks), dim3(1024), 0, s, st, w, pw, st.bsVersion < 6 ? 1 : 0); }
    { KScope ks_("k_lz_i_parse");                               // blocks the parallel parse has passed on (old layout, anything irregular)
      if (st.bsVersion < 6) hipLaunchKernelGGL(k_lz_i_parse<true>, dim3(st.nBlocks), dim3(64), 0, s, st, w);
      else hipLaunchKernelGGL(k_lz_i_parse<false>, dim3(st.nBlocks), dim3(64), 0, s, st, w); }
    const int wgPerBlock = 64;
    { KScope ks_("k_lz_i_first"); hipLaunchKernelGGL(k_lz_i_first, dim3(st.nBlocks * wgPerBlock), dim3(256), 0, s, w, wgPerBlock); }
    const dim3 grid((unsigned)((size_t)st.nBlocks * tilesPerBlock));
    { KScope ks_("k_lz_i_expand"); hipLaunchKernelGGL(k_lz_i_expand, grid, dim3(256), 0, s, w, tilesPerBlock); }
    { KScope ks_("k_lz_i_jump");
      for (int r = 0; r < LZI_ROUNDS; r++) hipLaunchKernelGGL(k_lz_i_jump, grid, dim3(256), 0, s, w, tilesPerBlock, r); }
    { KScope ks_("k_lz_i_emit"); hipLaunchKernelGGL(k_lz_i_emit, grid, dim3(256), 0, s, st, w, tilesPerBlock); }
    { KScope ks_("k_lz_inverse");                               // blocks the parse left alone (none with a workspace sized by maxCap)
      if (st.bsVersion < 6) hipLaunchKernelGGL(k_lz_inverse<true>, dim3(st.nBlocks), dim3(64), 0, s, st, (const u32*)w.info);
      else hipLaunchKernelGGL(k_lz_inverse<false>, dim3(st.nBlocks), dim3(64), 0, s, st, (const u32*)w.info); }
}

}  // namespace knz
