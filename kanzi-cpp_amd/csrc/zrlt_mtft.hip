// ZRLT byte transform on gfx950, batched over all blocks of a call (the MTFT kernels that used to share this file live in mtft.hip).
//
// Reference being replaced (bit-identical results, same success/failure decisions):
//   ZRLT  transform/ZRLT.cpp:27-117 (forward), :119-215 (inverse)
//
// The CPU loops are sequential; here they are rebuilt from scans:
//   ZRLT forward   zero-run starts need the run end = next non-zero position: per-tile first-nonzero +
//                  suffix-min over tiles; token sizes -> per-tile sums -> exclusive scan -> scatter.
//   ZRLT inverse   tokens are classified by two prefix-max scans (parity inside 0xFF streaks decides
//                  escape markers; "last non run-bit" gives each bit group); output offsets from a
//                  scan; the output is pre-zeroed so zero runs cost no stores.
#include "common.hpp"
#include "stages.hpp"

namespace knz {

constexpr u32 ZT = 4096;            // ZRLT tile (bytes) per 256-thread workgroup
constexpr u32 ZPT = 16;             // bytes per thread
constexpr u32 NOPOS = 0xFFFFFFFFu;

// exclusive sum over the 256 threads of a workgroup; returns the thread's offset, total in *tot
__device__ __forceinline__ u32 block_excl_scan256(u32 v, u32* lds4, u32* tot)
{
    const u32 incl = wave_incl_scan(v);
    const int wv = threadIdx.x >> 6;
    if (lane_id() == 63) lds4[wv] = incl;
    __syncthreads();
    u32 off = 0;
    for (int k = 0; k < wv; k++) off += lds4[k];
    *tot = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    __syncthreads();
    return off + incl - v;
}

// exclusive prefix maximum over the 256 threads of a workgroup (0 in front of thread 0): inside a wave by shuffles, the waves in front
// through LDS -- one barrier pair where a log-step scan through LDS takes sixteen
__device__ __forceinline__ u32 block_excl_max256(u32 v, u32* lds4)
{
    const int lane = lane_id(), wv = (int)(threadIdx.x >> 6);
    u32 val = v;
    for (int d = 1; d < 64; d <<= 1) {
        const u32 o = (u32)__shfl_up((int)val, (unsigned)d, 64);
        if (lane >= d) val = o > val ? o : val;
    }
    u32 ex = (u32)__shfl_up((int)val, 1u, 64);
    if (lane == 0) ex = 0;
    if (lane == 63) lds4[wv] = val;
    __syncthreads();
    for (int k = 0; k < wv; k++) { const u32 x = lds4[k]; ex = x > ex ? x : ex; }
    __syncthreads();
    return ex;
}

// ------------------------------------------------------------------------------------------------
// generic per-block scans over tile arrays (one workgroup per block)
// ------------------------------------------------------------------------------------------------
// exclusive sum of vals[b*per .. +cnt_b) in place; total -> tot[b]
__global__ __launch_bounds__(256) void k_tile_excl_sum(u32* vals, int per, const u32* __restrict__ lens, u32 tile, u32* tot)
{
    const int b = blockIdx.x;
    const u32 cnt = (lens[b] + tile - 1) / tile;
    u32* v = vals + (size_t)b * per;
    __shared__ u32 l4[4];
    __shared__ u32 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < cnt; base += 256) {
        const u32 i = base + threadIdx.x;
        const u32 x = i < cnt ? v[i] : 0;
        u32 t;
        const u32 off = block_excl_scan256(x, l4, &t);
        if (i < cnt) v[i] = carry + off;
        __syncthreads();
        if (threadIdx.x == 0) carry += t;
        __syncthreads();
    }
    if (threadIdx.x == 0) tot[b] = carry;
}

__device__ __forceinline__ u64 block_excl_scan256_64(u64 v, u64* lds4, u64* tot)
{
    const u64 incl = wave_incl_scan64(v);
    const int wv = threadIdx.x >> 6;
    if (lane_id() == 63) lds4[wv] = incl;
    __syncthreads();
    u64 off = 0;
    for (int k = 0; k < wv; k++) off += lds4[k];
    *tot = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    __syncthreads();
    return off + incl - v;
}

// 64-bit variant (ZRLT inverse: one token can expand to 2^30 zeros). Totals saturate at 2^32-1.
__global__ __launch_bounds__(256) void k_tile_excl_sum64(u64* vals, int per, const u32* __restrict__ lens, u32 tile, u32* tot)
{
    const int b = blockIdx.x;
    const u32 cnt = (lens[b] + tile - 1) / tile;
    u64* v = vals + (size_t)b * per;
    __shared__ u64 l4[4];
    __shared__ u64 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < cnt; base += 256) {
        const u32 i = base + threadIdx.x;
        const u64 x = i < cnt ? v[i] : 0;
        u64 t;
        const u64 off = block_excl_scan256_64(x, l4, &t);
        if (i < cnt) v[i] = carry + off;
        __syncthreads();
        if (threadIdx.x == 0) carry += t;
        __syncthreads();
    }
    if (threadIdx.x == 0) tot[b] = carry > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)carry;
}

// ------------------------------------------------------------------------------------------------
// ZRLT forward
// ------------------------------------------------------------------------------------------------
struct XfView {                     // per-block source/destination of one transform stage
    const u8* const* src;           // device array of per-block pointers
    u8* const* dst;
    const u32* len;                 // current lengths
    const u32* cap;                 // destination capacities (SliceArray::_length - _index)
};

__global__ __launch_bounds__(256) void k_zrlt_f_tileinfo(XfView v, int per, u32* __restrict__ firstNZ)
{
    const int b = blockIdx.y;
    const u32 t = blockIdx.x;
    const u32 n = v.len[b];
    const u32 base = t * ZT;
    if (base >= n) return;
    const u8* s = v.src[b];
    const u32 i0 = base + threadIdx.x * ZPT;
    u32 mine = NOPOS;
    if (ZPT == 16 && i0 + ZPT <= n && ((reinterpret_cast<uintptr_t>(s) + i0) & 15) == 0) {
        // the thread's 16 bytes as one load; the first non-zero byte from the words (little endian: the lowest set bit)
        const uint4 x = *reinterpret_cast<const uint4*>(s + i0);
        if (x.x) mine = i0 + ((u32)__ffs((int)x.x) - 1) / 8;
        else if (x.y) mine = i0 + 4 + ((u32)__ffs((int)x.y) - 1) / 8;
        else if (x.z) mine = i0 + 8 + ((u32)__ffs((int)x.z) - 1) / 8;
        else if (x.w) mine = i0 + 12 + ((u32)__ffs((int)x.w) - 1) / 8;
    } else
    for (u32 k = 0; k < ZPT; k++) {
        const u32 i = i0 + k;
        if (i < n && s[i] != 0) { mine = i; break; }
    }
    // block min
    __shared__ u32 m;
    if (threadIdx.x == 0) m = NOPOS;
    __syncthreads();
    if (mine != NOPOS) atomicMin(&m, mine);
    __syncthreads();
    if (threadIdx.x == 0) firstNZ[(size_t)b * per + t] = m;
}

// per block: nextNZ[t] = first non-zero position in tiles >= t (n if none). One workgroup per block: every thread owns a stretch of
// tiles, the stretches' summaries are combined through LDS (round 2 walked the whole block with one thread: 0.44 ms per 212 MB).
__global__ __launch_bounds__(256) void k_zrlt_f_suffix(const u32* __restrict__ firstNZ, u32* __restrict__ nextNZ, int per, const u32* __restrict__ lens, int nBlocks)
{
    const int b = blockIdx.x;
    if (b >= nBlocks) return;
    __shared__ u32 sum[256];
    const u32 n = lens[b];
    const u32 cnt = (n + ZT - 1) / ZT;
    const u32 chunk = (cnt + 255) / 256;
    const u32 lo = threadIdx.x * chunk, hi = (lo + chunk < cnt) ? lo + chunk : cnt;
    u32 cur = NOPOS;
    for (u32 t = hi; t > lo; t--) { const u32 f = firstNZ[(size_t)b * per + t - 1]; if (f != NOPOS) cur = f; }
    sum[threadIdx.x] = cur;                                  // first non-zero position of my stretch
    __syncthreads();
    u32 carry = NOPOS;                                       // ... of the stretches behind mine
    for (u32 j = threadIdx.x + 1; j < 256; j++) { const u32 x = sum[j]; if (x != NOPOS) { carry = x; break; } }
    cur = (carry != NOPOS) ? carry : n;
    for (u32 t = hi; t > lo; t--) {
        const u32 f = firstNZ[(size_t)b * per + t - 1];
        if (f != NOPOS) cur = f;
        nextNZ[(size_t)b * per + t - 1] = cur;
    }
}

// token size of position i (0 for zeros inside a run). runEnd = position of the next non-zero (or n).
__device__ __forceinline__ u32 zrlt_tok(u8 c, bool runStart, u32 i, u32 runEnd)
{
    if (c != 0) return c >= 0xFE ? 2u : 1u;
    if (!runStart) return 0;
    return (u32)ilog2_u32(runEnd - i + 1);
}

template <bool EMIT>
__global__ __launch_bounds__(256) void k_zrlt_f_pass(XfView v, int per, const u32* __restrict__ nextNZ, u32* __restrict__ tileSize,
                                                     const u32* __restrict__ tileOff, const u8* __restrict__ okFlags)
{
    const int b = blockIdx.y;
    const u32 t = blockIdx.x;
    const u32 n = v.len[b];
    const u32 base = t * ZT;
    if (base >= n) return;
    if (EMIT && !okFlags[b]) return;
    const u8* s = v.src[b];
    __shared__ u32 tnz[4];
    __shared__ u32 l4[4];
    const u32 i0 = base + threadIdx.x * ZPT;
    u8 c[ZPT];
    u32 myFirst = NOPOS;
    if (ZPT == 16 && i0 + ZPT <= n && ((reinterpret_cast<uintptr_t>(s) + i0) & 15) == 0) {
        const uint4 x = *reinterpret_cast<const uint4*>(s + i0);       // the thread's 16 bytes as one load
        const u32 w[4] = { x.x, x.y, x.z, x.w };
#pragma unroll
        for (u32 k = 0; k < ZPT; k++) {
            c[k] = (u8)(w[k >> 2] >> (8 * (k & 3)));
            if (myFirst == NOPOS && c[k] != 0) myFirst = i0 + k;
        }
    } else {
#pragma unroll
        for (u32 k = 0; k < ZPT; k++) {
            const u32 i = i0 + k;
            c[k] = (i < n) ? s[i] : (u8)1;            // padding behaves as "non-zero" but is never emitted
            if (myFirst == NOPOS && i < n && c[k] != 0) myFirst = i;
        }
    }
    // next non-zero after this thread's bytes: the threads behind it in its wave (shuffles), the waves behind (their minima through LDS),
    // then the next tiles (round 6: the log-step scan over the 256 threads through LDS cost sixteen barriers per tile and pass)
    u32 after = NOPOS;
    {
        const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
        u32 val = myFirst;
        for (int d = 1; d < 64; d <<= 1) {
            const u32 o = (u32)__shfl_down((int)val, (unsigned)d, 64);
            if (lane + d < 64) val = o < val ? o : val;
        }
        const u32 nxt = (u32)__shfl_down((int)val, 1u, 64);
        if (lane == 0) tnz[wave] = val;
        __syncthreads();
        after = (lane < 63) ? nxt : NOPOS;
        for (int w = wave + 1; w < 4; w++) { const u32 x = tnz[w]; after = x < after ? x : after; }
        if (after == NOPOS) {
            const u32 cnt = (n + ZT - 1) / ZT;
            after = (t + 1 < cnt) ? nextNZ[(size_t)b * per + t + 1] : n;
        }
    }
    const u8 prevByte = (i0 == 0 || i0 >= n + 1) ? (u8)1 : ((i0 - 1 < n) ? s[i0 - 1] : (u8)1);
    // the thread's non-zero bytes (inside the block) as a bit mask: where a run that starts at byte k ends is the first set bit above k
    // (round 6: a loop over the bytes behind k, unrolled sixteen times in two places, was two thirds of the kernel's instructions)
    u32 nzm = 0;
#pragma unroll
    for (u32 k = 0; k < ZPT; k++) nzm |= ((i0 + k < n && c[k] != 0) ? 1u : 0u) << k;
    auto run_end = [&](u32 k) -> u32 {
        const u32 m = nzm >> (k + 1);
        u32 e = m ? i0 + k + 1 + ((u32)__ffs((int)m) - 1u) : after;
        return e > n ? n : e;
    };
    u32 sizes[ZPT];
    u32 sum = 0;
#pragma unroll
    for (u32 k = 0; k < ZPT; k++) {
        const u32 i = i0 + k;
        u32 sz = 0;
        if (i < n) {
            const u8 pv = (k == 0) ? prevByte : c[k - 1];
            const bool runStart = (c[k] == 0) && (i == 0 || pv != 0);
            sz = zrlt_tok(c[k], runStart, i, runStart ? run_end(k) : after);
        }
        sizes[k] = sz;
        sum += sz;
    }
    u32 tot;
    const u32 off = block_excl_scan256(sum, l4, &tot);
    if (!EMIT) {
        if (threadIdx.x == 0) tileSize[(size_t)b * per + t] = tot;
        return;
    }
    // (tried in round 5: the tile's tokens put together in LDS and written as consecutive bytes -- 0.48 -> 0.58 ms; the byte stores of
    // neighbouring threads fall into the same lines and the write path merges them)
    u8* d = v.dst[b] + tileOff[(size_t)b * per + t] + off;
#pragma unroll
    for (u32 k = 0; k < ZPT; k++) {
        const u32 i = i0 + k;
        if (i >= n || sizes[k] == 0) continue;
        if (c[k] != 0) {
            if (c[k] >= 0xFE) { d[0] = 0xFF; d[1] = (u8)(c[k] - 0xFE); d += 2; }
            else { *d++ = (u8)(c[k] + 1); }
        } else {
            // run length L: emit the bits of L+1 below the MSB, one byte each (ZRLT.cpp:60-83)
            const u32 rl = run_end(k) - i + 1;
            for (int lg = (int)sizes[k] - 1; lg >= 0; lg--) *d++ = (u8)((rl >> lg) & 1);
        }
    }
}

// ok[b] = capacity >= n (getMaxEncodedLength) and every token fits; newLen[b] = total
__global__ void k_zrlt_f_finish(const u32* __restrict__ lens, const u32* __restrict__ caps, const u32* __restrict__ totals,
                                int nBlocks, u8* __restrict__ ok, u32* __restrict__ newLen)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    const u32 n = lens[b];
    const bool good = (n == 0) || (caps[b] >= n && totals[b] <= caps[b]);
    ok[b] = good ? 1 : 0;
    newLen[b] = totals[b];
}

// ------------------------------------------------------------------------------------------------
// ZRLT inverse
// ------------------------------------------------------------------------------------------------
// classes: 0 literal (emits v-1), 1 escape marker, 2 escape payload (emits 0xFE+v), 3 run bit
__global__ __launch_bounds__(256) void k_zrlt_i_lastnonff(XfView v, int per, u32* __restrict__ tileLast)
{
    const int b = blockIdx.y;
    const u32 t = blockIdx.x;
    const u32 n = v.len[b];
    const u32 base = t * ZT;
    if (base >= n) return;
    const u8* s = v.src[b];
    __shared__ u32 m;                // position+1 of the last non-0xFF byte in the tile, 0 = none
    if (threadIdx.x == 0) m = 0;
    __syncthreads();
    const u32 i0 = base + threadIdx.x * ZPT;
    u32 mine = 0;
    if (ZPT == 16 && i0 + ZPT <= n && ((reinterpret_cast<uintptr_t>(s) + i0) & 15) == 0) {
        // the thread's 16 bytes as one load; the last byte that is not 0xFF = the highest zero bit of the inverted words
        const uint4 x = *reinterpret_cast<const uint4*>(s + i0);
        const u32 a = ~x.x, bq = ~x.y, cq = ~x.z, dq = ~x.w;
        if (dq) mine = i0 + 12 + (31u - (u32)__clz((int)dq)) / 8 + 1;
        else if (cq) mine = i0 + 8 + (31u - (u32)__clz((int)cq)) / 8 + 1;
        else if (bq) mine = i0 + 4 + (31u - (u32)__clz((int)bq)) / 8 + 1;
        else if (a) mine = i0 + (31u - (u32)__clz((int)a)) / 8 + 1;
    } else
    for (u32 k = 0; k < ZPT; k++) { const u32 i = i0 + k; if (i < n && s[i] != 0xFF) mine = i + 1; }
    if (mine) atomicMax(&m, mine);
    __syncthreads();
    if (threadIdx.x == 0) tileLast[(size_t)b * per + t] = m;
}

// exclusive prefix max over tiles, one workgroup per block (stretches of tiles per thread, summaries through LDS)
__global__ __launch_bounds__(256) void k_tile_prefix_max(const u32* __restrict__ in, u32* __restrict__ out, int per, const u32* __restrict__ lens, int nBlocks)
{
    const int b = blockIdx.x;
    if (b >= nBlocks) return;
    __shared__ u32 sum[256];
    const u32 cnt = (lens[b] + ZT - 1) / ZT;
    const u32 chunk = (cnt + 255) / 256;
    const u32 lo = threadIdx.x * chunk, hi = (lo + chunk < cnt) ? lo + chunk : cnt;
    u32 mx = 0;
    for (u32 t = lo; t < hi; t++) { const u32 x = in[(size_t)b * per + t]; mx = x > mx ? x : mx; }
    sum[threadIdx.x] = mx;
    __syncthreads();
    u32 cur = 0;
    for (u32 j = 0; j < threadIdx.x; j++) { const u32 x = sum[j]; cur = x > cur ? x : cur; }
    for (u32 t = lo; t < hi; t++) {
        const u32 x = in[(size_t)b * per + t];
        out[(size_t)b * per + t] = cur;
        cur = x > cur ? x : cur;
    }
}

// class of byte i given lastNonFF(i-1)+1 (= position+1 of the last non-0xFF byte strictly before i, 0 = none)
__device__ __forceinline__ u32 zrlt_class(u8 c, u32 i, u32 lastNonFFBefore)
{
    // streak of 0xFF bytes immediately before i has length i - lastNonFFBefore; odd length => the byte at i
    // is the payload of the marker at i-1
    const u32 streak = i - lastNonFFBefore;
    const bool payload = (streak & 1) != 0;
    if (payload) return 2;
    if (c == 0xFF) return 1;
    if (c <= 1) return 3;
    return 0;
}

// mode 0: tileLastNonR ; mode 1: tile output sizes ; mode 2: emit
template <int MODE>
__global__ __launch_bounds__(256) void k_zrlt_i_pass(XfView v, int per, const u32* __restrict__ prevNonFF, u32* __restrict__ tileLastNonR,
                                                     const u32* __restrict__ prevNonR, u64* __restrict__ tileSize,
                                                     const u64* __restrict__ tileOff, u32* __restrict__ errFlags, const u8* __restrict__ okFlags)
{
    const int b = blockIdx.y;
    const u32 t = blockIdx.x;
    const u32 n = v.len[b];
    const u32 base = t * ZT;
    if (base >= n) return;
    if (MODE == 2 && !okFlags[b]) return;
    const u8* s = v.src[b];
    __shared__ u32 m4[4];
    __shared__ u64 l4[4];
    __shared__ u32 mx;
    const u32 i0 = base + threadIdx.x * ZPT;
    u8 c[ZPT];
    u32 lastNonFF = 0;               // position+1 of last non-FF within my bytes
    if (ZPT == 16 && i0 + ZPT <= n && ((reinterpret_cast<uintptr_t>(s) + i0) & 15) == 0) {
        const uint4 x = *reinterpret_cast<const uint4*>(s + i0);       // the thread's 16 bytes as one load
        const u32 w[4] = { x.x, x.y, x.z, x.w };
#pragma unroll
        for (u32 k = 0; k < ZPT; k++) {
            c[k] = (u8)(w[k >> 2] >> (8 * (k & 3)));
            if (c[k] != 0xFF) lastNonFF = i0 + k + 1;
        }
    } else {
#pragma unroll
        for (u32 k = 0; k < ZPT; k++) {
            const u32 i = i0 + k;
            c[k] = (i < n) ? s[i] : (u8)2;
            if (i < n && c[k] != 0xFF) lastNonFF = i + 1;
        }
    }
    // exclusive prefix max over threads of lastNonFF
    u32 before = block_excl_max256(lastNonFF, m4);
    const u32 tp = prevNonFF[(size_t)b * per + t];
    before = before > tp ? before : tp;

    u32 cls[ZPT];
    u32 lastNonR = 0;                // position+1 of last non-run-bit byte within my bytes
    {
        u32 lb = before;
#pragma unroll
        for (u32 k = 0; k < ZPT; k++) {
            const u32 i = i0 + k;
            cls[k] = (i < n) ? zrlt_class(c[k], i, lb) : 0u;
            if (i < n && c[k] != 0xFF) lb = i + 1;
            if (i < n && cls[k] != 3) lastNonR = i + 1;
        }
    }
    if (MODE == 0) {
        if (threadIdx.x == 0) mx = 0;
        __syncthreads();
        if (lastNonR) atomicMax(&mx, lastNonR);
        __syncthreads();
        if (threadIdx.x == 0) tileLastNonR[(size_t)b * per + t] = mx;
        return;
    }
    // exclusive prefix max over threads of lastNonR
    u32 nrBefore = block_excl_max256(lastNonR, m4);
    const u32 tr = prevNonR[(size_t)b * per + t];
    nrBefore = nrBefore > tr ? nrBefore : tr;

    u32 sizes[ZPT];
    u64 sum = 0;
    u32 err = 0;
    {
        u32 nr = nrBefore;           // group start = nr (position of first run bit)
#pragma unroll
        for (u32 k = 0; k < ZPT; k++) {
            const u32 i = i0 + k;
            u32 sz = 0;
            if (i < n) {
                if (cls[k] == 0 || cls[k] == 2) sz = 1;
                else if (cls[k] == 1) { if (i + 1 >= n) err = 1; }       // truncated escape (ZRLT.cpp:177-180)
                else {
                    // run bit: the group's last byte carries the whole run
                    const bool last = (i + 1 >= n) || (((k + 1 < ZPT) ? c[k + 1] : s[i + 1]) > 1);
                    if (last) {
                        const u32 start = nr;
                        const u32 klen = i - start + 1;
                        if (klen > 30) err = 1;              // runs are bounded by the 1 GiB block size
                        else {
                            u32 rl = 1;
                            for (u32 q = start; q <= i; q++) rl = (rl << 1) | (u32)s[q];
                            sz = rl - 1;
                        }
                    }
                }
                if (cls[k] != 3) nr = i + 1;
            }
            sizes[k] = sz;
            sum += sz;
        }
    }
    u64 tot;
    const u64 off = block_excl_scan256_64(sum, l4, &tot);
    if (MODE == 1) {
        if (err) atomicOr(&errFlags[b], 1u);
        if (threadIdx.x == 0) tileSize[(size_t)b * per + t] = tot;
        return;
    }
    u8* d = v.dst[b] + tileOff[(size_t)b * per + t] + off;
#pragma unroll
    for (u32 k = 0; k < ZPT; k++) {
        const u32 i = i0 + k;
        if (i >= n) continue;
        if (cls[k] == 0) { *d = (u8)(c[k] - 1); d += 1; }
        else if (cls[k] == 2) { *d = (u8)(0xFE + c[k]); d += 1; }
        else d += sizes[k];          // zero run: destination was pre-zeroed
    }
}

__global__ void k_zrlt_i_finish(const u32* __restrict__ lens, const u32* __restrict__ caps, const u32* __restrict__ totals,
                                const u32* __restrict__ errFlags, int nBlocks, u8* __restrict__ ok, u32* __restrict__ newLen)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    const bool good = (lens[b] == 0) || (!errFlags[b] && totals[b] <= caps[b]);
    ok[b] = good ? 1 : 0;
    newLen[b] = totals[b];
}

// zero the destination prefix each block may write (capacity-bounded)
__global__ __launch_bounds__(256) void k_zero_dst(XfView v, const u32* __restrict__ totals)
{
    const int b = blockIdx.y;
    u32 n = totals[b];
    if (n > v.cap[b]) n = v.cap[b];
    u8* d = v.dst[b];
    const size_t stride = (size_t)gridDim.x * 256 * 16;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16; i < n; i += stride) {
        if (i + 16 <= n && ((reinterpret_cast<uintptr_t>(d + i) & 15) == 0)) *reinterpret_cast<uint4*>(d + i) = make_uint4(0, 0, 0, 0);
        else for (size_t k = i; k < n && k < i + 16; k++) d[k] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static XfView mk(const XfStage& st) { XfView v; v.src = st.src; v.dst = st.dst; v.len = st.len; v.cap = st.cap; return v; }

void launch_zrlt_forward(hipStream_t s, const XfStage& st)
{
    const XfView v = mk(st);
    const int per = (int)((st.maxLen + ZT - 1) / ZT);
    u32* firstNZ = st.scratchU32;
    u32* nextNZ = firstNZ + (size_t)st.nBlocks * per;
    u32* tileSize = nextNZ + (size_t)st.nBlocks * per;
    u32* totals = tileSize + (size_t)st.nBlocks * per;
    const dim3 grid(per, st.nBlocks);
    { KScope ks_("k_zrlt_f_tileinfo"); hipLaunchKernelGGL(k_zrlt_f_tileinfo, grid, dim3(256), 0, s, v, per, firstNZ); }
    { KScope ks_("k_zrlt_f_suffix"); hipLaunchKernelGGL(k_zrlt_f_suffix, dim3(st.nBlocks), dim3(256), 0, s, firstNZ, nextNZ, per, st.len, st.nBlocks); }
    { KScope ks_("k_zrlt_f_count"); hipLaunchKernelGGL(k_zrlt_f_pass<false>, grid, dim3(256), 0, s, v, per, nextNZ, tileSize, nullptr, nullptr); }
    { KScope ks_("k_tile_excl_sum"); hipLaunchKernelGGL(k_tile_excl_sum, dim3(st.nBlocks), dim3(256), 0, s, tileSize, per, st.len, ZT, totals); }
    { KScope ks_("k_zrlt_f_finish"); hipLaunchKernelGGL(k_zrlt_f_finish, dim3((st.nBlocks + 255) / 256), dim3(256), 0, s, st.len, st.cap, totals, st.nBlocks, st.ok, st.newLen); }
    { KScope ks_("k_zrlt_f_emit"); hipLaunchKernelGGL(k_zrlt_f_pass<true>, grid, dim3(256), 0, s, v, per, nextNZ, nullptr, tileSize, st.ok); }
}

void launch_zrlt_inverse(hipStream_t s, const XfStage& st)
{
    const XfView v = mk(st);
    const int per = (int)(((st.maxLen + ZT - 1) / ZT + 1) & ~1u);
    u32* a = st.scratchU32;                                  // tileLastNonFF -> prevNonFF (separate arrays)
    u32* prevNonFF = a + (size_t)st.nBlocks * per;
    u32* bArr = prevNonFF + (size_t)st.nBlocks * per;       // tileLastNonR
    u32* prevNonR = bArr + (size_t)st.nBlocks * per;
    u64* tileSize = reinterpret_cast<u64*>(prevNonR + (size_t)st.nBlocks * per);   // scratch is 8-byte aligned, per is padded even
    u32* totals = reinterpret_cast<u32*>(tileSize + (size_t)st.nBlocks * per);
    u32* errFlags = totals + st.nBlocks;
    const dim3 grid(per, st.nBlocks);
    hipMemsetAsync(errFlags, 0, sizeof(u32) * st.nBlocks, s);
    { KScope ks_("k_zrlt_i_lastnonff"); hipLaunchKernelGGL(k_zrlt_i_lastnonff, grid, dim3(256), 0, s, v, per, a); }
    { KScope ks_("k_tile_prefix_max"); hipLaunchKernelGGL(k_tile_prefix_max, dim3(st.nBlocks), dim3(256), 0, s, a, prevNonFF, per, st.len, st.nBlocks); }
    { KScope ks_("k_zrlt_i_class"); hipLaunchKernelGGL(k_zrlt_i_pass<0>, grid, dim3(256), 0, s, v, per, prevNonFF, bArr, nullptr, nullptr, nullptr, nullptr, nullptr); }
    { KScope ks_("k_tile_prefix_max"); hipLaunchKernelGGL(k_tile_prefix_max, dim3(st.nBlocks), dim3(256), 0, s, bArr, prevNonR, per, st.len, st.nBlocks); }
    { KScope ks_("k_zrlt_i_count"); hipLaunchKernelGGL(k_zrlt_i_pass<1>, grid, dim3(256), 0, s, v, per, prevNonFF, nullptr, prevNonR, tileSize, nullptr, errFlags, nullptr); }
    { KScope ks_("k_tile_excl_sum64"); hipLaunchKernelGGL(k_tile_excl_sum64, dim3(st.nBlocks), dim3(256), 0, s, tileSize, per, st.len, ZT, totals); }
    { KScope ks_("k_zrlt_i_finish"); hipLaunchKernelGGL(k_zrlt_i_finish, dim3((st.nBlocks + 255) / 256), dim3(256), 0, s, st.len, st.cap, totals, errFlags, st.nBlocks, st.ok, st.newLen); }
    { KScope ks_("k_zero_dst"); hipLaunchKernelGGL(k_zero_dst, dim3(256, st.nBlocks), dim3(256), 0, s, v, totals); }
    { KScope ks_("k_zrlt_i_emit"); hipLaunchKernelGGL(k_zrlt_i_pass<2>, grid, dim3(256), 0, s, v, per, prevNonFF, nullptr, prevNonR, nullptr, tileSize, nullptr, st.ok); }
}

size_t zrlt_scratch_u32(int nBlocks, u32 maxLen) { return (size_t)nBlocks * ((maxLen + ZT - 1) / ZT + 2) * 6 + 2 * (size_t)nBlocks + 64; }

}  // namespace knz
