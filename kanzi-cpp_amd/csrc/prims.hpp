// Device-wide primitives written for this library (gfx950, wave64): prefix scans over 32-bit words and a stable LSD
// radix sort (8-bit digits) over segmented arrays. They replace the rocPRIM calls of rounds 1-2 in the BWT stages and
// the LZ match finder. Every size a kernel needs can be read from device memory (`nDev`, `base[]`), so that a sequence of
// launches does not need the host to read a counter back in between.
//
// Sort pass = four launches: tile histograms (one row of 256 counters per tile of 4096 keys, coalesced), a column scan
// of the rows in two levels (inside groups of RS_GROUP tiles, then over the groups and the digits), and the scatter:
// the tile's keys are ranked inside the workgroup (ballot matching inside a wave, rows in order; per-wave digit counters
// scanned in (digit, wave) order), staged through LDS in sorted order and written out as runs of consecutive addresses.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include "common.hpp"

namespace knz {
namespace prims {

// ------------------------------------------------------------------------------------------------
// scans over u32
// ------------------------------------------------------------------------------------------------
enum ScanOp { SCAN_SUM_EXCL = 0, SCAN_MAX_INCL = 1, SCAN_MIN_INCL = 2 };

constexpr u32 SC_TILE = 4096;            // 256 threads x 16 words

template <int OP> __device__ __forceinline__ u32 sc_ident() { return OP == SCAN_MIN_INCL ? 0xFFFFFFFFu : 0u; }
template <int OP> __device__ __forceinline__ u32 sc_op(u32 a, u32 b) { return OP == SCAN_SUM_EXCL ? a + b : (OP == SCAN_MAX_INCL ? (a > b ? a : b) : (a < b ? a : b)); }

template <int OP>
__device__ __forceinline__ u32 sc_wave_incl(u32 v)
{
    if (OP == SCAN_SUM_EXCL) return wave_incl_scan(v);
    for (int o = 1; o < 64; o <<= 1) { const u32 t = (u32)__shfl_up((int)v, (unsigned)o, 64); if (lane_id() >= o) v = sc_op<OP>(t, v); }
    return v;
}

// inclusive scan of one value per thread over a workgroup of NW waves; *total = reduction over the workgroup
template <int OP, int NW = 4>
__device__ __forceinline__ u32 sc_block_incl(u32 v, u32* wsum /* [NW] in LDS */, u32* total)
{
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    const u32 incl = sc_wave_incl<OP>(v);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 carry = sc_ident<OP>();
    for (int w = 0; w < wave; w++) carry = sc_op<OP>(carry, wsum[w]);
    u32 tot = sc_ident<OP>();
    for (int w = 0; w < NW; w++) tot = sc_op<OP>(tot, wsum[w]);
    *total = tot;
    __syncthreads();
    return sc_op<OP>(carry, incl);
}

// partial[t] = reduction of tile t
template <int OP>
__global__ __launch_bounds__(256) void k_scan_reduce(const u32* __restrict__ in, u32 nMax, const u32* __restrict__ nDev, u32* __restrict__ partial)
{
    __shared__ u32 wsum[4];
    const u32 n = nDev ? (*nDev < nMax ? *nDev : nMax) : nMax;
    const u32 nT = (n + SC_TILE - 1) / SC_TILE;
    for (u32 t = blockIdx.x; t < nT; t += gridDim.x) {
        u32 acc = sc_ident<OP>();
        const u32 i0 = t * SC_TILE + threadIdx.x * 16u;
        if (i0 + 16 <= n) {
            const uint4* p = reinterpret_cast<const uint4*>(in + i0);
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint4 x = p[k]; acc = sc_op<OP>(sc_op<OP>(sc_op<OP>(sc_op<OP>(acc, x.x), x.y), x.z), x.w); }
        } else {
            for (u32 k = 0; k < 16; k++) if (i0 + k < n) acc = sc_op<OP>(acc, in[i0 + k]);
        }
        u32 tot;
        sc_block_incl<OP>(acc, wsum, &tot);
        if (threadIdx.x == 0) partial[t] = tot;
    }
}

// one workgroup of 256 threads scans a short array in place (exclusive for the sum, which is what the tiles need as their
// carry; for max / min the carry of tile t is the inclusive result of tile t-1, i.e. also "exclusive")
template <int OP>
__global__ __launch_bounds__(256) void k_scan_carries(u32* __restrict__ partial, u32 nMax, const u32* __restrict__ nDev)
{
    __shared__ u32 wsum[4];
    __shared__ u32 inclAll[256];
    const u32 n = nDev ? (*nDev < nMax ? *nDev : nMax) : nMax;
    const u32 nT = (n + SC_TILE - 1) / SC_TILE;
    u32 carry = sc_ident<OP>();
    for (u32 c0 = 0; c0 < nT; c0 += 256 * 16) {
        u32 x[16];
        const u32 i0 = c0 + threadIdx.x * 16u;
        u32 acc = sc_ident<OP>();
#pragma unroll
        for (u32 k = 0; k < 16; k++) { x[k] = (i0 + k < nT) ? partial[i0 + k] : sc_ident<OP>(); acc = sc_op<OP>(acc, x[k]); }
        u32 tot;
        const u32 incl = sc_block_incl<OP>(acc, wsum, &tot);
        inclAll[threadIdx.x] = incl;
        __syncthreads();
        u32 run = sc_op<OP>(carry, threadIdx.x ? inclAll[threadIdx.x - 1] : sc_ident<OP>());     // everything before this thread's 16 values
        __syncthreads();
#pragma unroll
        for (u32 k = 0; k < 16; k++) { if (i0 + k < nT) partial[i0 + k] = run; run = sc_op<OP>(run, x[k]); }
        carry = sc_op<OP>(carry, tot);
    }
}

template <int OP>
__global__ __launch_bounds__(256) void k_scan_apply(const u32* __restrict__ in, u32* __restrict__ out, u32 nMax, const u32* __restrict__ nDev,
                                                    const u32* __restrict__ carries, u32* __restrict__ totalOut)
{
    __shared__ u32 wsum[4];
    __shared__ u32 inclAll[256];
    const u32 n = nDev ? (*nDev < nMax ? *nDev : nMax) : nMax;
    const u32 nT = (n + SC_TILE - 1) / SC_TILE;
    for (u32 t = blockIdx.x; t < nT; t += gridDim.x) {
        u32 x[16];
        const u32 i0 = t * SC_TILE + threadIdx.x * 16u;
        u32 acc = sc_ident<OP>();
        if (i0 + 16 <= n) {
            const uint4* p = reinterpret_cast<const uint4*>(in + i0);
#pragma unroll
            for (int k = 0; k < 4; k++) { const uint4 v = p[k]; x[4 * k] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w; }
        } else {
#pragma unroll
            for (u32 k = 0; k < 16; k++) x[k] = (i0 + k < n) ? in[i0 + k] : sc_ident<OP>();
        }
#pragma unroll
        for (u32 k = 0; k < 16; k++) acc = sc_op<OP>(acc, x[k]);
        u32 tot;
        const u32 incl = sc_block_incl<OP>(acc, wsum, &tot);
        inclAll[threadIdx.x] = incl;
        __syncthreads();
        u32 run = threadIdx.x ? inclAll[threadIdx.x - 1] : sc_ident<OP>();
        __syncthreads();
        run = sc_op<OP>(carries[t], run);
        u32 y[16];
#pragma unroll
        for (u32 k = 0; k < 16; k++) {
            if (OP == SCAN_SUM_EXCL) { y[k] = run; run += x[k]; }
            else { run = sc_op<OP>(run, x[k]); y[k] = run; }
        }
        if (i0 + 16 <= n) {
            uint4* q = reinterpret_cast<uint4*>(out + i0);
#pragma unroll
            for (int k = 0; k < 4; k++) q[k] = make_uint4(y[4 * k], y[4 * k + 1], y[4 * k + 2], y[4 * k + 3]);
        } else {
#pragma unroll
            for (u32 k = 0; k < 16; k++) if (i0 + k < n) out[i0 + k] = y[k];
        }
        if (totalOut != nullptr && t + 1 == nT && threadIdx.x == 255) *totalOut = run;       // sum of everything (exclusive scan: n-th value)
    }
}

static inline size_t scan_tmp_bytes(size_t nMax) { return (((nMax + SC_TILE - 1) / SC_TILE) * 4 + 255 + 256) & ~(size_t)255; }

// out[i] = scan of in[0..i) (sum) or in[0..i] (max / min); in == out allowed; both 16-byte aligned. n = min(nMax, *nDev) when nDev
// is given. totalOut (optional, sum only): the sum of all n values.
template <int OP>
static inline void launch_scan(hipStream_t s, const u32* in, u32* out, size_t nMax, const u32* nDev, void* tmp, u32* totalOut = nullptr)
{
    if (nMax == 0) return;
    u32* partial = reinterpret_cast<u32*>(tmp);
    const u32 nT = (u32)((nMax + SC_TILE - 1) / SC_TILE);
    const u32 grid = nT < 4096 ? nT : 4096;
    hipLaunchKernelGGL((k_scan_reduce<OP>), dim3(grid), dim3(256), 0, s, in, (u32)nMax, nDev, partial);
    hipLaunchKernelGGL((k_scan_carries<OP>), dim3(1), dim3(256), 0, s, partial, (u32)nMax, nDev);
    hipLaunchKernelGGL((k_scan_apply<OP>), dim3(grid), dim3(256), 0, s, in, out, (u32)nMax, nDev, partial, totalOut);
}

// ------------------------------------------------------------------------------------------------
// radix sort
// ------------------------------------------------------------------------------------------------
// Eight waves (tiles of 8,192 keys) since the end of round 4: half as many tiles to rank, publish and look back over; round 0 of the
// suffix sort 5.06 -> 4.57 ms, the run round's member sort 2.06 -> 1.86 ms on the headline workload. (The kernels are templates with
// one name per instantiation whatever the tile size, so two translation units built with different values would share one
// symbol and fault on the device -- which happened once, through a -D on one object file. The value is therefore a constant of this
// header, not a build option; tests/test_host_abi.py checks that every object file of the library carries the same tile size.)
constexpr int RS_WAVES = 8;               // waves per workgroup: each owns 16 rows of 64 consecutive keys
// k_rs_onesweep<u64, true>: sK 64 KiB + sV 32 KiB + counters 8 KiB + flags: gfx950 has 160 KiB of LDS per CU; fail the build elsewhere
static_assert(sizeof(unsigned long long) * 1024u * RS_WAVES + 4u * 1024u * RS_WAVES + 4u * 256u * RS_WAVES + 4096u <= 160u * 1024u,
              "radix tile does not fit the LDS of gfx950");
// every translation unit that includes this header emits the symbol rs_tile_tag<keys per tile>(): one name in all of them or the build is mixed
template <unsigned KEYS> __attribute__((used, visibility("default"))) void rs_tile_tag() {}
template void rs_tile_tag<1024u * RS_WAVES>();
constexpr int RS_THREADS = 64 * RS_WAVES;
constexpr u32 RS_TILE = 1024u * RS_WAVES; // keys per tile
constexpr u32 RS_GROUP = 64;             // tiles per group of the column scan
constexpr int RS_MAXPASS = 8;            // passes the one-read variant counts ahead

// One row of 64 keys of a wave: pos = the number of keys of the wave's earlier rows and of the row's lower lanes that show the same digit
// (cntw: the wave's 256 running counters). A row that shows ONE digit (runs, the constant high bytes of keys) costs two ballots. Else the
// lanes that share a digit come from eight ballots, accumulated as "lanes that DIFFER from me in some bit": per bit the lane's bit spread
// over a word (one arithmetic shift), XORed with the two halves of the ballot, two bits ORed in per three-input OR -- about 45 vector
// instructions per row where the select form (peers &= one ? bal : ~bal on 64-bit values) compiled to about 100, which made a radix pass
// compute-bound (0.97 ms for 3.4 GB that stream in 0.73). Round 6 also measured a bit set per digit in LDS (every lane ORs its bit in, reads
// the pair of words, clears its bit): five dependent LDS round trips per row, slower than the ballots (r0 4.08 -> 4.46 ms).
#define RS_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
__device__ __forceinline__ unsigned long long digit_peers_fast(bool valid, u32 dg)
{
    const unsigned long long va = __ballot(valid);
    u32 dlo = 0, dhi = 0;                                   // lanes whose digit differs from mine in some bit
#pragma unroll
    for (int bit = 0; bit < 8; bit += 2) {
        const int x0 = (int)(dg << (31 - bit)), x1 = (int)(dg << (30 - bit));
        const unsigned long long b0 = __ballot(x0 < 0), b1 = __ballot(x1 < 0);
        const u32 m0 = (u32)(x0 >> 31), m1 = (u32)(x1 >> 31);
        dlo |= ((u32)b0 ^ m0) | ((u32)b1 ^ m1);
        dhi |= ((u32)(b0 >> 32) ^ m0) | ((u32)(b1 >> 32) ^ m1);
    }
    const unsigned long long same = ~(((unsigned long long)dhi << 32) | dlo);
    return valid ? (same & va) : 0ull;
}
__device__ __forceinline__ u32 rs_row_rank(bool valid, u32 dg, u32* cntw, int lane, unsigned long long ltMask)
{
    (void)lane;
    unsigned long long peers;
    const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)dg);              // (lane 0 is valid when any lane of the row is)
    if (__ballot(valid && dg != d0) == 0) peers = __ballot(valid);
    else peers = digit_peers_fast(valid, dg);
    const u32 rnk = (u32)__popcll(peers & ltMask);
    const u32 old = valid ? cntw[dg] : 0u;
    RS_WAVE_FENCE();
    if (valid && rnk == 0) cntw[dg] = old + (u32)__popcll(peers);              // (the lowest lane of every digit)
    RS_WAVE_FENCE();
    return old + rnk;
}

// Segments: segment g is [base[g], base[g+1]) of the key arrays (dense, base[0] = 0); a tile never straddles segments.
// tileOff[g] = number of tiles in the segments before g (tileOff[nSeg] = all tiles), grpOff likewise for tile groups.
struct RsLayout {
    const u32* base;
    u32* tileOff;        // [nSeg + 1]
    u32* grpOff;         // [nSeg + 1]
    u32* tileHist;       // [tiles][256]: counts, then exclusive inside the tile's group
    u32* grpSum;         // [groups][256]: group totals, then exclusive over the groups of the segment
    u32* digitBase;      // [nSeg][256]: output position of the first key with that digit (includes base[g])
    u32* histAll;        // [RS_MAXPASS][nSeg][256]: one-read variant: digit counts of every pass, then the digits' first output positions
    int nSeg;
};

static __global__ void k_rs_layout(RsLayout L)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u32 t = 0, g = 0;
    for (int sgm = 0; sgm < L.nSeg; sgm++) {
        L.tileOff[sgm] = t; L.grpOff[sgm] = g;
        const u32 len = L.base[sgm + 1] - L.base[sgm];
        const u32 nT = (len + RS_TILE - 1) / RS_TILE;
        t += nT; g += (nT + RS_GROUP - 1) / RS_GROUP;
    }
    L.tileOff[L.nSeg] = t; L.grpOff[L.nSeg] = g;
}

// Source of a pass: element i of segment sgm (which starts at b0 and has len elements) as a key, the digit of a key, and the
// digit of element i alone (what the counting kernel needs). The functor may read anything: a key array, or the text for the
// first pass of the suffix sort.
// STAGED: the source wants the tile's input staged in LDS first (stage() by the whole workgroup, then load_staged() per element).
template <class KEY> struct DigitOfKey {
    static constexpr bool STAGED = false;
    const KEY* keys; int shift; u32 mask;
    __device__ __forceinline__ bool stage(int, u32, u32, u32*, int) const { return false; }
    __device__ __forceinline__ KEY load_staged(int, u32, u32, u32, const u32*, u32) const { return (KEY)0; }
    __device__ __forceinline__ KEY load(int, u32 b0, u32, u32 i) const { return keys[b0 + i]; }
    __device__ __forceinline__ u32 digit(KEY k) const { return (u32)(k >> shift) & mask; }
    __device__ __forceinline__ u32 digit_at(int, u32 b0, u32, u32 i) const { return (u32)(keys[b0 + i] >> shift) & mask; }
};

// Tile histograms. A row of 64 keys that shows one digit (runs, zero stretches, the high bytes of small keys) costs one counter
// update; other rows go through LDS atomics on the wave's private counters (equal digits inside a row serialise, which bounds the cost
// at the multiplicity of the row's most frequent digit).
template <class SRC>
__global__ __launch_bounds__(RS_THREADS) void k_rs_count(SRC src, RsLayout L)
{
    __shared__ u32 cnt[RS_WAVES][256];
    const int sgm = blockIdx.y;
    const u32 b0 = L.base[sgm], len = L.base[sgm + 1] - b0;
    const u32 nT = (len + RS_TILE - 1) / RS_TILE;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (u32 t = blockIdx.x; t < nT; t += gridDim.x) {
        for (int q = tid; q < RS_WAVES * 256; q += RS_THREADS) (&cnt[0][0])[q] = 0;
        __syncthreads();
        u32 dg[16];
        const u32 i0 = t * RS_TILE + (u32)wave * 1024u + (u32)lane;
#pragma unroll
        for (int r = 0; r < 16; r++) { const u32 i = i0 + (u32)r * 64u; dg[r] = (i < len) ? src.digit_at(sgm, b0, len, i) : 0xFFFFFFFFu; }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const bool valid = dg[r] != 0xFFFFFFFFu;
            const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)dg[r]);        // (lane 0 is valid when any lane of the row is)
            const unsigned long long va = __ballot(valid);
            if (__ballot(valid && dg[r] != d0) == 0) { if (lane == 0 && va) cnt[wave][d0] += (u32)__popcll(va); }
            else if (valid) atomicAdd(&cnt[wave][dg[r]], 1u);
            KNZ_WAVE_ORDER();
        }
        __syncthreads();
        if (tid < 256) { u32 sum = 0; for (int w = 0; w < RS_WAVES; w++) sum += cnt[w][tid]; L.tileHist[((size_t)L.tileOff[sgm] + t) * 256 + tid] = sum; }
        __syncthreads();
    }
}

// exclusive scan of the tile rows inside every group of RS_GROUP tiles, one thread per digit; grpSum = the group's totals
static __global__ __launch_bounds__(256) void k_rs_colscan1(RsLayout L)
{
    const int sgm = blockIdx.y;
    const u32 len = L.base[sgm + 1] - L.base[sgm];
    const u32 nT = (len + RS_TILE - 1) / RS_TILE, nG = (nT + RS_GROUP - 1) / RS_GROUP;
    const int tid = (int)threadIdx.x;
    for (u32 g = blockIdx.x; g < nG; g += gridDim.x) {
        const u32 t0 = g * RS_GROUP, t1 = (t0 + RS_GROUP < nT) ? t0 + RS_GROUP : nT;
        u32* p = L.tileHist + (size_t)L.tileOff[sgm] * 256 + tid;
        u32 run = 0;
        for (u32 t = t0; t < t1; t++) { const u32 x = p[(size_t)t * 256]; p[(size_t)t * 256] = run; run += x; }
        L.grpSum[((size_t)L.grpOff[sgm] + g) * 256 + tid] = run;
    }
}

// per segment: exclusive scan over the groups (per digit), then over the digits
static __global__ __launch_bounds__(256) void k_rs_colscan2(RsLayout L)
{
    __shared__ u32 wsum[4];
    __shared__ u32 inclAll[256];
    const int sgm = blockIdx.x;
    const u32 len = L.base[sgm + 1] - L.base[sgm];
    const u32 nT = (len + RS_TILE - 1) / RS_TILE, nG = (nT + RS_GROUP - 1) / RS_GROUP;
    const int tid = (int)threadIdx.x;
    u32* p = L.grpSum + (size_t)L.grpOff[sgm] * 256 + tid;
    u32 run = 0;
    for (u32 g = 0; g < nG; g++) { const u32 x = p[(size_t)g * 256]; p[(size_t)g * 256] = run; run += x; }
    u32 tot;
    const u32 incl = sc_block_incl<SCAN_SUM_EXCL>(run, wsum, &tot);
    inclAll[tid] = incl;
    __syncthreads();
    L.digitBase[sgm * 256 + tid] = L.base[sgm] + (tid ? inclAll[tid - 1] : 0u);
}

// Ranks the tile's keys (held 16 per thread, element (wave, row, lane) = tile index wave*1024 + row*64 + lane) by digit, stably,
// and leaves in pos[] the index every element takes in the tile's sorted order; dStart[d] (LDS) = first sorted index of digit d.
__device__ __forceinline__ void rs_rank_tile(const u32 (&dg)[16], const bool (&valid)[16], u32 (&pos)[16], u32 (*cnt)[256], u32* dStart, u32* wsum, u32* inclAll)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int q = tid; q < RS_WAVES * 256; q += RS_THREADS) (&cnt[0][0])[q] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) pos[r] = rs_row_rank(valid[r], dg[r], cnt[wave], lane, ltMask);
    __syncthreads();
    // (digit, wave) order: the first 256 threads own a digit each
    u32 c[RS_WAVES];
    u32 sum = 0;
    if (tid < 256) { for (int w = 0; w < RS_WAVES; w++) { c[w] = cnt[w][tid]; sum += c[w]; } }
    u32 tot;
    const u32 incl = sc_block_incl<SCAN_SUM_EXCL, RS_WAVES>(sum, wsum, &tot);
    if (tid < 256) inclAll[tid] = incl;
    __syncthreads();
    if (tid < 256) {
        u32 run = tid ? inclAll[tid - 1] : 0u;
        dStart[tid] = run;
        for (int w = 0; w < RS_WAVES; w++) { cnt[w][tid] = run; run += c[w]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) if (valid[r]) pos[r] += cnt[wave][dg[r]];
}

template <class KEY, bool HAS_VAL, class SRC>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(SRC src, const u32* __restrict__ vin, KEY* __restrict__ kout, u32* __restrict__ vout, RsLayout L)
{
    __shared__ u32 cnt[RS_WAVES][256];
    __shared__ u32 dStart[256];
    __shared__ u32 gBase[256];
    __shared__ u32 wsum[RS_WAVES];
    __shared__ u32 inclAll[256];
    __shared__ KEY sK[RS_TILE];
    __shared__ u32 sV[HAS_VAL ? RS_TILE : 1];
    const int sgm = blockIdx.y;
    const u32 b0 = L.base[sgm], len = L.base[sgm + 1] - b0;
    const u32 nT = (len + RS_TILE - 1) / RS_TILE;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (u32 t = blockIdx.x; t < nT; t += gridDim.x) {
        KEY key[16]; u32 val[16]; u32 dg[16], pos[16]; bool valid[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const u32 i = t * RS_TILE + (u32)wave * 1024u + (u32)r * 64u + (u32)lane;
            valid[r] = i < len;
            key[r] = valid[r] ? src.load(sgm, b0, len, i) : (KEY)0;
            if (HAS_VAL) val[r] = valid[r] ? vin[b0 + i] : 0u;
            dg[r] = valid[r] ? src.digit(key[r]) : 0u;
        }
        rs_rank_tile(dg, valid, pos, cnt, dStart, wsum, inclAll);
        if (tid < 256) gBase[tid] = L.digitBase[sgm * 256 + tid] + L.grpSum[((size_t)L.grpOff[sgm] + t / RS_GROUP) * 256 + tid]
                                    + L.tileHist[((size_t)L.tileOff[sgm] + t) * 256 + tid];
#pragma unroll
        for (int r = 0; r < 16; r++) if (valid[r]) { sK[pos[r]] = key[r]; if (HAS_VAL) sV[pos[r]] = val[r]; }
        __syncthreads();
        const u32 cntTile = (len - t * RS_TILE < RS_TILE) ? len - t * RS_TILE : RS_TILE;
#pragma unroll 4
        for (u32 j = (u32)tid; j < cntTile; j += RS_THREADS) {
            const KEY k = sK[j];
            const u32 d = src.digit(k);
            const u32 at = gBase[d] + (j - dStart[d]);
            kout[at] = k;
            if (HAS_VAL) vout[at] = sV[j];
        }
        __syncthreads();
    }
}

// ---- one read per pass ("onesweep") ----------------------------------------------------------------------------------------
// The pass above reads its keys twice (tile histograms, then the scatter) because a tile has to know how many keys with each digit
// sit in the tiles in front of it. Digit totals do not depend on the order of the keys, so ONE kernel counts the digits of all passes
// of a sort from its input (k_rs_hist_all), and a tile learns the counts of its predecessors while it runs: it publishes its own
// counts (per digit one word: 2 flag bits + 30 bits) as soon as it has ranked its keys, then walks back over the tiles before it,
// adding their counts until it meets one that already knows its prefix (decoupled look-back: 256 threads, one digit each). A tile only
// ever waits for tiles with a lower index, and tiles are handed out by a ticket counter as workgroups start, so whoever is waited for
// is running or done. Flag and count travel in one word, read and written with device-scope atomics (the
// tiles in front of this one run on other XCDs).
constexpr u32 RS_FLAG_AGG = 1u << 30, RS_FLAG_PRE = 2u << 30, RS_VAL_MASK = (1u << 30) - 1u;
#ifdef KNZ_EMU
__device__ __forceinline__ u32 rs_ld_dev(const u32* p) { return *p; }
__device__ __forceinline__ void rs_st_dev(u32* p, u32 v) { *p = v; }
#define RS_SPIN() do { fprintf(stderr, "k_rs_onesweep: a tile in front of this one has not run\n"); abort(); } while (0)
#else
#define RS_SPIN() __builtin_amdgcn_s_sleep(1)
__device__ __forceinline__ u32 rs_ld_dev(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rs_st_dev(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

// digit counts of passes 0 .. nPass-1 (digit of pass p = (key >> (shift0 + 8 p)) & 255, the last pass masked with lastMask)
template <class KEY, class SRC>
__global__ __launch_bounds__(RS_THREADS) void k_rs_hist_all(SRC src, RsLayout L, int shift0, int nPass, u32 lastMask)
{
    __shared__ u32 cnt[RS_MAXPASS][256];
    const int sgm = blockIdx.y;
    const u32 b0 = L.base[sgm], len = L.base[sgm + 1] - b0;
    const u32 nT = (len + RS_TILE - 1) / RS_TILE;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = tid; q < RS_MAXPASS * 256; q += RS_THREADS) (&cnt[0][0])[q] = 0;
    __syncthreads();
    for (u32 t = blockIdx.x; t < nT; t += gridDim.x) {
#pragma unroll 4
        for (int r = 0; r < 16; r++) {
            const u32 i = t * RS_TILE + (u32)wave * 1024u + (u32)r * 64u + (u32)lane;
            const bool valid = i < len;
            const KEY k = valid ? src.load(sgm, b0, len, i) : (KEY)0;
            for (int ps = 0; ps < nPass; ps++) {
                const u32 dg = (u32)(k >> (shift0 + 8 * ps)) & ((ps + 1 == nPass) ? lastMask : 255u);
                const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)dg);
                const unsigned long long va = __ballot(valid);
                if (__ballot(valid && dg != d0) == 0) { if (lane == 0 && va) atomicAdd(&cnt[ps][d0], (u32)__popcll(va)); }
                else if (valid) atomicAdd(&cnt[ps][dg], 1u);
                KNZ_WAVE_ORDER();
            }
        }
    }
    __syncthreads();
    for (int q = tid; q < nPass * 256; q += RS_THREADS) {
        const u32 c = (&cnt[0][0])[q];
        if (c) atomicAdd(&L.histAll[((size_t)(q >> 8) * L.nSeg + sgm) * 256 + (q & 255)], c);
    }
}

// counts -> first output position of every digit (per pass and segment)
static __global__ __launch_bounds__(256) void k_rs_digit_bases(RsLayout L)
{
    __shared__ u32 wsum[4];
    __shared__ u32 inclAll[256];
    const int sgm = blockIdx.x, ps = blockIdx.y;
    const int tid = (int)threadIdx.x;
    u32* h = L.histAll + ((size_t)ps * L.nSeg + sgm) * 256;
    const u32 c = h[tid];
    u32 tot;
    const u32 incl = sc_block_incl<SCAN_SUM_EXCL>(c, wsum, &tot);
    inclAll[tid] = incl;
    __syncthreads();
    h[tid] = L.base[sgm] + (tid ? inclAll[tid - 1] : 0u);
}

template <class KEY, bool HAS_VAL, class SRC>
__global__ __launch_bounds__(RS_THREADS) void k_rs_onesweep(SRC src, const u32* __restrict__ vin, KEY* __restrict__ kout, u32* __restrict__ vout, RsLayout L, int pass)
{
    __shared__ u32 cnt[RS_WAVES][256];
    __shared__ u32 dStart[256];
    __shared__ u32 gBase[256];
    __shared__ u32 wsum[RS_WAVES];
    __shared__ u32 inclAll[256];
    __shared__ KEY sK[RS_TILE];
    __shared__ u32 sV[HAS_VAL ? RS_TILE : 1];
    __shared__ u32 sTicket;
    const int sgm = blockIdx.y;
    const u32 b0 = L.base[sgm], len = L.base[sgm + 1] - b0;
    const u32 nT = (len + RS_TILE - 1) / RS_TILE;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // The tile is not blockIdx.x but a ticket drawn when the workgroup starts: a tile then only ever waits for tiles whose workgroups
    // started before its own -- they are running or done -- whatever order the dispatcher, several queues or several XCDs start
    // workgroups in.
    if (tid == 0) sTicket = atomicAdd(&L.grpSum[sgm], 1u);
    __syncthreads();
    const u32 t = sTicket;
    if (t >= nT) return;
    KEY key[16]; u32 val[16]; u32 dg[16], pos[16]; bool valid[16];
    // (a staged source borrows sK, which takes the keys only behind the next barrier but one)
    bool staged = false;
    if (SRC::STAGED) { staged = src.stage(sgm, len, t * RS_TILE, reinterpret_cast<u32*>(sK), tid); __syncthreads(); }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32 i = t * RS_TILE + (u32)wave * 1024u + (u32)r * 64u + (u32)lane;
        valid[r] = i < len;
        if (SRC::STAGED && staged) key[r] = valid[r] ? src.load_staged(sgm, b0, len, i, reinterpret_cast<const u32*>(sK), t * RS_TILE) : (KEY)0;
        else key[r] = valid[r] ? src.load(sgm, b0, len, i) : (KEY)0;
        if (HAS_VAL) val[r] = valid[r] ? vin[b0 + i] : 0u;
        dg[r] = valid[r] ? src.digit(key[r]) : 0u;
    }
    // rs_rank_tile in two halves: the tile's digit counts are known after the first one and are published right away, so that the
    // tiles behind this one find them when they look back; this tile looks back only after it has staged its keys
    const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int q = tid; q < RS_WAVES * 256; q += RS_THREADS) (&cnt[0][0])[q] = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) pos[r] = rs_row_rank(valid[r], dg[r], cnt[wave], lane, ltMask);
    __syncthreads();
    u32 c[RS_WAVES];
    u32 mine = 0;
    if (tid < 256) { for (int w = 0; w < RS_WAVES; w++) { c[w] = cnt[w][tid]; mine += c[w]; } }
    u32* stat = L.tileHist + ((size_t)L.tileOff[sgm] + t) * 256 + (tid & 255);     // this tile's status word for digit tid
    if (tid < 256) rs_st_dev(stat, (t == 0 ? RS_FLAG_PRE : RS_FLAG_AGG) | mine);
    u32 tot;
    const u32 incl = sc_block_incl<SCAN_SUM_EXCL, RS_WAVES>(mine, wsum, &tot);
    if (tid < 256) inclAll[tid] = incl;
    __syncthreads();
    if (tid < 256) {
        u32 run = tid ? inclAll[tid - 1] : 0u;
        dStart[tid] = run;
        for (int w = 0; w < RS_WAVES; w++) { cnt[w][tid] = run; run += c[w]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) if (valid[r]) { pos[r] += cnt[wave][dg[r]]; sK[pos[r]] = key[r]; if (HAS_VAL) sV[pos[r]] = val[r]; }
    const u32 cntTile = (len - t * RS_TILE < RS_TILE) ? len - t * RS_TILE : RS_TILE;
    if (tid < 256) {
        u32 excl = 0;
        if (t != 0) {
            for (u32 q = 1; q <= t; q++) {
                u32 v;
                while (((v = rs_ld_dev(stat - (size_t)q * 256)) >> 30) == 0) RS_SPIN();
                excl += v & RS_VAL_MASK;
                if (v & RS_FLAG_PRE) break;
            }
            rs_st_dev(stat, RS_FLAG_PRE | (excl + mine));
        }
        gBase[tid] = L.histAll[((size_t)pass * L.nSeg + sgm) * 256 + tid] + excl;
    }
    __syncthreads();
#pragma unroll 4
    for (u32 j = (u32)tid; j < cntTile; j += RS_THREADS) {
        const KEY k = sK[j];
        const u32 d = src.digit(k);
        const u32 at = gBase[d] + (j - dStart[d]);
        kout[at] = k;
        if (HAS_VAL) vout[at] = sV[j];
    }
}

static __global__ void k_rs_one_segment(u32* seg2, u32 n) { if (threadIdx.x == 0 && blockIdx.x == 0) { seg2[0] = 0; seg2[1] = n; } }

struct RsWs { RsLayout L; size_t maxTiles; };

static inline size_t rs_align(size_t x) { return (x + 255) & ~(size_t)255; }

// workspace for sorts of up to maxN keys in up to maxSeg segments
static inline size_t rs_ws_bytes(size_t maxN, int maxSeg)
{
    const size_t tiles = maxN / RS_TILE + (size_t)maxSeg + 1, groups = tiles / RS_GROUP + (size_t)maxSeg + 1;
    return rs_align(tiles * 1024) + rs_align(groups * 1024) + rs_align((size_t)maxSeg * 1024) + 2 * rs_align(4ull * (maxSeg + 1)) + rs_align((size_t)RS_MAXPASS * maxSeg * 1024) + 256;
}

static inline RsWs rs_carve(void* p, size_t maxN, int maxSeg, const u32* base, int nSeg)
{
    const size_t tiles = maxN / RS_TILE + (size_t)maxSeg + 1, groups = tiles / RS_GROUP + (size_t)maxSeg + 1;
    u8* q = reinterpret_cast<u8*>(p);
    RsWs w;
    w.L.base = base; w.L.nSeg = nSeg;
    w.L.tileHist = reinterpret_cast<u32*>(q); q += rs_align(tiles * 1024);
    w.L.grpSum = reinterpret_cast<u32*>(q); q += rs_align(groups * 1024);
    w.L.digitBase = reinterpret_cast<u32*>(q); q += rs_align((size_t)maxSeg * 1024);
    w.L.tileOff = reinterpret_cast<u32*>(q); q += rs_align(4ull * (maxSeg + 1));
    w.L.grpOff = reinterpret_cast<u32*>(q); q += rs_align(4ull * (maxSeg + 1));
    w.L.histAll = reinterpret_cast<u32*>(q); q += rs_align((size_t)RS_MAXPASS * maxSeg * 1024);
    w.maxTiles = tiles;
    return w;
}

static inline void rs_launch_layout(hipStream_t s, const RsWs& w) { hipLaunchKernelGGL(k_rs_layout, dim3(1), dim3(64), 0, s, w.L); }

// grid.x for a pass over segments of at most maxSegLen keys
static inline unsigned rs_grid_x(size_t maxSegLen, int nSeg)
{
    size_t t = (maxSegLen + RS_TILE - 1) / RS_TILE;
    const size_t cap = nSeg > 1 ? 2048 : 8192;
    if (t > cap) t = cap;
    return (unsigned)(t ? t : 1);
}

// one stable pass on the digit SRC extracts; the layout kernel must have run for these segments
template <class KEY, bool HAS_VAL, class SRC>
static inline void rs_launch_pass(hipStream_t s, const RsWs& w, SRC src, const u32* vin, KEY* kout, u32* vout, size_t maxSegLen)
{
    const dim3 grid(rs_grid_x(maxSegLen, w.L.nSeg), (unsigned)w.L.nSeg);
    const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>(((maxSegLen + RS_TILE - 1) / RS_TILE + RS_GROUP - 1) / RS_GROUP, 1024));
    hipLaunchKernelGGL((k_rs_count<SRC>), grid, dim3(RS_THREADS), 0, s, src, w.L);
    hipLaunchKernelGGL(k_rs_colscan1, dim3(gx, (unsigned)w.L.nSeg), dim3(256), 0, s, w.L);
    hipLaunchKernelGGL(k_rs_colscan2, dim3((unsigned)w.L.nSeg), dim3(256), 0, s, w.L);
    hipLaunchKernelGGL((k_rs_scatter<KEY, HAS_VAL, SRC>), grid, dim3(RS_THREADS), 0, s, src, vin, kout, vout, w.L);
}

// knob "rs_onesweep" (KNZ_RS_ONESWEEP): 1 = the one-read passes where a sort's passes are known ahead (default), 0 = count + scatter
static inline std::atomic<int>& rs_onesweep_knob()
{
    static std::atomic<int> v{ getenv("KNZ_RS_ONESWEEP") ? atoi(getenv("KNZ_RS_ONESWEEP")) : 1 };
    return v;
}

// digit counts of all passes of a sort on bits [loBit, loBit + 8 nPass) (the last digit masked) from its input, and the digits' bases
template <class KEY, class SRC>
static inline void rs_launch_hist_all(hipStream_t s, const RsWs& w, SRC src, size_t maxSegLen, int loBit, int nPass, u32 lastMask)
{
    hipMemsetAsync(w.L.histAll, 0, (size_t)nPass * w.L.nSeg * 1024, s);
    size_t gx = ((maxSegLen + RS_TILE - 1) / RS_TILE + 7) / 8;                      // eight tiles per workgroup: few global atomics
    if (gx < 1) gx = 1;
    if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL((k_rs_hist_all<KEY, SRC>), dim3((unsigned)gx, (unsigned)w.L.nSeg), dim3(RS_THREADS), 0, s, src, w.L, loBit, nPass, lastMask);
    hipLaunchKernelGGL(k_rs_digit_bases, dim3((unsigned)w.L.nSeg, (unsigned)nPass), dim3(256), 0, s, w.L);
}

// pass number `pass` of a sort whose digits rs_launch_hist_all has counted
template <class KEY, bool HAS_VAL, class SRC>
static inline void rs_launch_pass_os(hipStream_t s, const RsWs& w, SRC src, const u32* vin, KEY* kout, u32* vout, size_t maxSegLen, int pass)
{
    const size_t tiles = (maxSegLen + RS_TILE - 1) / RS_TILE;
    const size_t used = std::min<size_t>(w.maxTiles, (tiles + 1) * (size_t)w.L.nSeg);     // status words of the tiles this sort has
    hipMemsetAsync(w.L.tileHist, 0, used * 1024, s);
    hipMemsetAsync(w.L.grpSum, 0, 4ull * (size_t)w.L.nSeg, s);                           // the segments' ticket counters
    hipLaunchKernelGGL((k_rs_onesweep<KEY, HAS_VAL, SRC>), dim3((unsigned)(tiles ? tiles : 1), (unsigned)w.L.nSeg), dim3(RS_THREADS), 0, s, src, vin, kout, vout, w.L, pass);
}

// Stable LSD sort on key bits [loBit, hiBit). Returns 0 when the result is in (ka, va), 1 when in (kb, vb).
template <class KEY, bool HAS_VAL>
static inline int rs_sort(hipStream_t s, const RsWs& w, KEY* ka, KEY* kb, u32* va, u32* vb, size_t maxSegLen, int loBit, int hiBit, bool oneRead = true)
{
    int cur = 0;
    const int nPass = (hiBit - loBit + 7) / 8;
    // the look-back words of the one-read passes hold 2 flag bits + a 30-bit count of keys in front of a tile (RS_VAL_MASK): a
    // segment of 2^30 keys or more (one LZ block of 1 GiB; the run sorts of a 2 GiB batch) takes the count + scatter passes
    const bool os = oneRead && rs_onesweep_knob().load() != 0 && nPass >= 1 && nPass <= RS_MAXPASS && maxSegLen < (size_t)RS_VAL_MASK;
    if (os) {
        DigitOfKey<KEY> src; src.keys = ka; src.shift = loBit; src.mask = 255u;
        const int rem = hiBit - loBit - 8 * (nPass - 1);
        rs_launch_hist_all<KEY>(s, w, src, maxSegLen, loBit, nPass, rem >= 8 ? 255u : ((1u << rem) - 1u));
    }
    int pass = 0;
    for (int sh = loBit; sh < hiBit; sh += 8, pass++) {
        DigitOfKey<KEY> src; src.keys = cur ? kb : ka; src.shift = sh; src.mask = (hiBit - sh >= 8) ? 255u : ((1u << (hiBit - sh)) - 1u);
        if (os) rs_launch_pass_os<KEY, HAS_VAL>(s, w, src, cur ? vb : va, cur ? ka : kb, cur ? va : vb, maxSegLen, pass);
        else rs_launch_pass<KEY, HAS_VAL>(s, w, src, cur ? vb : va, cur ? ka : kb, cur ? va : vb, maxSegLen);
        cur ^= 1;
    }
    return cur;
}

// The same with the input left untouched: the first pass reads (kin, vin), the others move between (kb, vb) and (kc, vc).
// Returns 0 when the result is in (kb, vb), 1 when in (kc, vc).
template <class KEY, bool HAS_VAL>
static inline int rs_sort_keep(hipStream_t s, const RsWs& w, const KEY* kin, const u32* vin, KEY* kb, KEY* kc, u32* vb, u32* vc, size_t maxSegLen, int loBit, int hiBit)
{
    int cur = -1;                                            // -1: input, 0: b, 1: c
    const int nPass = (hiBit - loBit + 7) / 8;
    const bool os = rs_onesweep_knob().load() != 0 && nPass >= 1 && nPass <= RS_MAXPASS && maxSegLen < (size_t)RS_VAL_MASK;
    if (os) {
        DigitOfKey<KEY> src; src.keys = kin; src.shift = loBit; src.mask = 255u;
        const int rem = hiBit - loBit - 8 * (nPass - 1);
        rs_launch_hist_all<KEY>(s, w, src, maxSegLen, loBit, nPass, rem >= 8 ? 255u : ((1u << rem) - 1u));
    }
    int pass = 0;
    for (int sh = loBit; sh < hiBit; sh += 8, pass++) {
        DigitOfKey<KEY> src; src.keys = cur < 0 ? kin : (cur ? kc : kb); src.shift = sh; src.mask = (hiBit - sh >= 8) ? 255u : ((1u << (hiBit - sh)) - 1u);
        const int nxt = cur < 0 ? 0 : (cur ^ 1);
        if (os) rs_launch_pass_os<KEY, HAS_VAL>(s, w, src, cur < 0 ? vin : (cur ? vc : vb), nxt ? kc : kb, nxt ? vc : vb, maxSegLen, pass);
        else rs_launch_pass<KEY, HAS_VAL>(s, w, src, cur < 0 ? vin : (cur ? vc : vb), nxt ? kc : kb, nxt ? vc : vb, maxSegLen);
        cur = nxt;
    }
    return cur < 0 ? 0 : cur;
}

}  // namespace prims
}  // namespace knz
