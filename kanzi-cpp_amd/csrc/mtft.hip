// MTFT (move-to-front) on gfx950, batched over all blocks of a call.
//
// Reference being replaced: transform/SBRT.cpp:46-97 (forward), :99-145 (inverse) with MODE_MTF
// (:28-31), i.e. the classic move-to-front over the identity initial list.
//
// The CPU loop is one dependency chain per block. Here a block is cut into 4 KiB tiles and
//   forward   the recency list at a tile start is recovered without running the chain: symbols
//             ordered by last occurrence before the tile (per-tile last-occurrence table, exclusive
//             prefix-max over tiles), never-seen symbols in ascending order (rank-by-counting sort);
//   inverse   a tile's effect on the list is a permutation of list POSITIONS that does not depend on
//             the list's content, so every tile is decoded symbolically (from the identity list),
//             the permutations are prefix-composed per block, and a parallel gather resolves ids.
// Inside a tile ONE WAVE HOLDS THE WHOLE 256-ENTRY LIST IN REGISTERS, 4 entries per lane packed in a
// dword: finding a symbol is a SWAR zero-byte test + ballot, move-to-front is one cross-lane shift.
// Cost per byte is therefore independent of the rank (the CPU loop, and a lane-serial GPU loop, pay
// O(rank)); many waves per SIMD hide the short dependent chain.
#include "common.hpp"
#include "stages.hpp"
#include <cstdlib>
#include <atomic>

namespace knz {

// bytes per tile (one wave): 4096, or 1024 when the batch has fewer 4 KiB tiles than the device has wave slots -- the tile kernels
// are one dependent chain per tile, so a small batch finishes in the time of ONE chain and shorter chains are what shortens it
// (2 blocks of 8 MiB: k_mtf_f_rank 0.82 -> see DESIGN.md)
// knob "mtf_chain" (KNZ_MTF_CHAIN=1): forward ranks / inverse ids by the byte-serial chain kernels of rounds 2-5 instead of the data-parallel ones
static std::atomic<int> g_mtfChainKnob([]() { const char* e = getenv("KNZ_MTF_CHAIN"); return e ? atoi(e) : 0; }());
int mtft_tune_chain(int on) { g_mtfChainKnob.store(on ? 1 : 0); return 0; }
static std::atomic<int> g_mtfTileKnob([]() { const char* e = getenv("KNZ_MTF_TILE"); return e ? atoi(e) : 0; }());      // 0 = by batch size; 1024 / 4096 force
int mtft_tune(int tileBytes) { g_mtfTileKnob.store((tileBytes == 1024 || tileBytes == 4096) ? tileBytes : 0); return 0; }
static inline u32 mtf_tile_bytes(int nBlocks, u32 maxLen)
{
    if (const int k = g_mtfTileKnob.load()) return (u32)k;
    return ((u64)nBlocks * ((maxLen + 4095) / 4096) < 12288ull) ? 1024u : 4096u;
}

struct XfView {
    const u8* const* src;
    u8* const* dst;
    const u32* len;
    const u32* cap;
};

// count of trailing zeros of a value that is never 0 (no select for the empty case: the chain is bound by the scalar unit)
#define MTF_CTZ64(x) __builtin_ctzll((unsigned long long)(x))
#define MTF_CTZ32(x) __builtin_ctz((unsigned)(x))

// (a & m) | (b & ~m) with a uniform m: one v_bfi instead of a scalar complement and two vector operations
__device__ __forceinline__ u32 mtf_bfi(u32 m, u32 a, u32 b)
{
#ifdef KNZ_EMU
    return (a & m) | (b & ~m);
#else
    u32 r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(m), "v"(a), "v"(b));
    return r;
#endif
}

// rotate list positions [0, rank] right by one and put `front` at position 0.
// w: my 4 entries (position 4*lane in byte 0). rank = 4*lane0 + byteIdx (uniform).
// v << 24 on the vector unit although v is uniform (the scalar unit is the busy one in these chains)
__device__ __forceinline__ u32 mtf_top(u32 v)
{
#ifdef KNZ_EMU
    return v << 24;
#else
    u32 r;
    asm("v_lshlrev_b32 %0, 24, %1" : "=v"(r) : "s"(v));
    return r;
#endif
}

// frontTop: any word with the new front symbol in its top byte
__device__ __forceinline__ u32 mtf_rotate(u32 w, int lane, int lane0, u32 mask, u32 frontTop)
{
    // lane i takes lane i-1's entries, lane 0 the new front symbol (the `old` operand of the DPP move); then one funnel shift
    const u32 prev = (u32)__builtin_amdgcn_update_dpp((int)frontTop, (int)w, 0x138, 0xF, 0xF, false);   // wave_shr:1
    const u32 shifted = __builtin_amdgcn_alignbit(w, prev, 24);                  // (w << 8) | (prev >> 24)
    // mask (uniform): the bytes of lane0's word up to and including the symbol's old place
    const u32 merged = mtf_bfi(mask, shifted, w);
    const u32 upTo = (lane == lane0) ? merged : w;
    return (lane < lane0) ? shifted : upTo;
}

// per tile: last occurrence (position+1) of each symbol -> tileLast[b][t][256]
template <u32 MT>
__global__ __launch_bounds__(64) void k_mtf_f_last(XfView v, int perTiles, u32* __restrict__ tileLast)
{
    const int b = blockIdx.y;
    const u32 n = (v.len[b] <= v.cap[b]) ? v.len[b] : 0;
    const u32 tbase = blockIdx.x * MT;
    if (tbase >= n) return;
    const u8* s = v.src[b];
    __shared__ u32 last[256];
    const int lane = lane_id();
    for (int i = lane; i < 256; i += 64) last[i] = 0;
    __syncthreads();
    const u32 end = (tbase + MT < n) ? tbase + MT : n;
    if ((reinterpret_cast<uintptr_t>(s) & 3) == 0) {
        // four bytes per lane; a byte that is followed by its like is not the last of its symbol (behind a BWT most bytes are), the
        // fourth is always entered (its successor sits in another lane)
        const u32* s4 = reinterpret_cast<const u32*>(s + tbase);
        const u32 full = (end - tbase) / 4;
        for (u32 q = (u32)lane; q < full; q += 64) {
            const u32 x = s4[q];
            const u32 i = tbase + 4 * q;
            const u32 b0 = x & 0xFF, b1 = (x >> 8) & 0xFF, b2 = (x >> 16) & 0xFF, b3 = x >> 24;
            if (b0 != b1) atomicMax(&last[b0], i + 1);
            if (b1 != b2) atomicMax(&last[b1], i + 2);
            if (b2 != b3) atomicMax(&last[b2], i + 3);
            atomicMax(&last[b3], i + 4);
        }
        for (u32 i = tbase + 4 * full + (u32)lane; i < end; i += 64) atomicMax(&last[s[i]], i + 1);
    } else {
        for (u32 i = tbase + lane; i < end; i += 64) atomicMax(&last[s[i]], i + 1);
    }
    __syncthreads();
    u32* o = tileLast + ((size_t)b * perTiles + blockIdx.x) * 256;
    for (int i = lane; i < 256; i += 64) o[i] = last[i];
}

// Exclusive prefix max over the tiles of a block, per symbol, in two levels so that no thread walks more than ~sqrt(tiles) of
// them: tiles are grouped in segments of segT; level 1 scans inside a segment (from 0) and leaves the segment's maximum in segMax,
// level 2 scans the segment maxima; the consumer (k_mtf_f_rank) takes the maximum of both.
__host__ __device__ inline u32 mtf_seg_tiles(u32 perTiles) { u32 g = 1; while (g * g < perTiles) g <<= 1; return g; }

template <u32 MT>
__global__ __launch_bounds__(256) void k_mtf_f_scan(u32* __restrict__ tileLast, int perTiles, u32 segT, u32 nSeg, const u32* __restrict__ lens,
                                                    u32* __restrict__ segMax)
{
    const int b = blockIdx.y;
    const u32 seg = blockIdx.x;
    const u32 cnt = (lens[b] + MT - 1) / MT;
    const u32 t0 = seg * segT;
    const u32 t1 = (t0 + segT < cnt) ? t0 + segT : cnt;
    u32* p = tileLast + (size_t)b * perTiles * 256 + threadIdx.x;
    u32 cur = 0;
    for (u32 t = t0; t < t1; t++) {
        const u32 x = p[(size_t)t * 256];
        p[(size_t)t * 256] = cur;
        cur = x > cur ? x : cur;
    }
    segMax[((size_t)b * nSeg + seg) * 256 + threadIdx.x] = cur;
}

__global__ __launch_bounds__(256) void k_mtf_f_scan2(u32* __restrict__ segMax, u32 nSeg)
{
    u32* p = segMax + (size_t)blockIdx.x * nSeg * 256 + threadIdx.x;
    u32 cur = 0;
    for (u32 g = 0; g < nSeg; g++) {
        const u32 x = p[(size_t)g * 256];
        p[(size_t)g * 256] = cur;
        cur = x > cur ? x : cur;
    }
}

template <u32 MT>
__global__ __launch_bounds__(64) void k_mtf_f_rank(XfView v, int perTiles, const u32* __restrict__ tileState, u32 segT, u32 nSeg,
                                                   const u32* __restrict__ segMax)
{
    const int b = blockIdx.y;
    const u32 n = (v.len[b] <= v.cap[b]) ? v.len[b] : 0;
    const u32 tbase = blockIdx.x * MT;
    if (tbase >= n) return;
    const u8* s = v.src[b];
    u8* d = v.dst[b];
    __shared__ u32 keys[256];
    __shared__ u32 listw[64];
    const int lane = lane_id();
    // start list: symbols by last occurrence desc, never-seen symbols ascending
    const u32* st = tileState + ((size_t)b * perTiles + blockIdx.x) * 256;
    const u32* sg = segMax + ((size_t)b * nSeg + blockIdx.x / segT) * 256;       // last occurrences in the segments before mine
    for (int i = lane; i < 256; i += 64) { const u32 x = st[i], y = sg[i]; keys[i] = x > y ? x : y; }
    __syncthreads();
    u8* listb = reinterpret_cast<u8*>(listw);
    for (int c = lane; c < 256; c += 64) {
        const u32 kc = keys[c];
        u32 r = 0;
        for (int q = 0; q < 256; q++) {
            const u32 kq = keys[q];
            r += (kq > kc || (kq == kc && q < c)) ? 1u : 0u;
        }
        listb[r] = (u8)c;
    }
    __syncthreads();
    u32 w = listw[lane];
    u32 front = (u32)__builtin_amdgcn_readfirstlane((int)w) & 0xFF;   // symbol at list position 0 (uniform)
    const u32 cnt = (n - tbase < MT) ? (n - tbase) : MT;
    const u8* src = s + tbase;
    u8* dst = d + tbase;
    const bool al = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3) == 0;
    // four input bytes (uniform) -> four ranks
    u32 front4 = front * 0x01010101u;           // (kept beside `front`: the chain is bound by the scalar unit, every instruction counts)
    auto rank4 = [&](u32 in4) -> u32 {
        u32 out4 = 0;
        if (in4 != front4) {                    // runs (the common case after a BWT) keep the list unchanged: rank 0
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32 c = (in4 >> (8 * q)) & 0xFF;
                if (c == front) continue;
                front = c;
                front4 = c * 0x01010101u;
                const u32 x = w ^ front4;
                const u32 hz = (x - 0x01010101u) & ~x & 0x80808080u;
                const u64 m = __ballot(hz != 0);                        // (never 0: the list holds every symbol)
                const int lane0 = MTF_CTZ64(m);
                const u32 hz0 = (u32)__builtin_amdgcn_readlane((int)hz, lane0);
                const int byteIdx = MTF_CTZ32(hz0) >> 3;
                out4 |= (u32)(4 * lane0 + byteIdx) << (8 * q);
                w = mtf_rotate(w, lane, lane0, ((hz0 & (0u - hz0)) << 1) - 1u, front4);
            }
        }
        return out4;
    };
    u32 k = 0;
    if (al && cnt == MT) {
        // full tile: the 4 KiB are loaded up front (16 dwords per lane, coalesced) and handed to the chain with
        // readlane, results are collected the same way and stored coalesced -- no memory latency inside the chain
        u32 inr[MT / 256], outr[MT / 256];
#pragma unroll
        for (int t = 0; t < (int)(MT / 256); t++) inr[t] = reinterpret_cast<const u32*>(src)[64 * t + lane];
#pragma unroll
        for (int t = 0; t < (int)(MT / 256); t++) {
            u32 acc = 0;
            for (int l = 0; l < 64; l++) {
                const u32 o = rank4((u32)__builtin_amdgcn_readlane((int)inr[t], l));
                acc = (lane == l) ? o : acc;
            }
            outr[t] = acc;
        }
#pragma unroll
        for (int t = 0; t < (int)(MT / 256); t++) reinterpret_cast<u32*>(dst)[64 * t + lane] = outr[t];
        k = MT;
    }
    for (; k + 4 <= cnt; k += 4) {
        u32 in4;
        if (al) in4 = *reinterpret_cast<const u32*>(src + k);
        else in4 = (u32)src[k] | ((u32)src[k + 1] << 8) | ((u32)src[k + 2] << 16) | ((u32)src[k + 3] << 24);
        in4 = (u32)__builtin_amdgcn_readfirstlane((int)in4);
        const u32 out4 = rank4(in4);
        if (lane == 0) {
            if (al) *reinterpret_cast<u32*>(dst + k) = out4;
            else { dst[k] = (u8)out4; dst[k + 1] = (u8)(out4 >> 8); dst[k + 2] = (u8)(out4 >> 16); dst[k + 3] = (u8)(out4 >> 24); }
        }
    }
    for (; k < cnt; k++) {
        const u32 c = (u32)__builtin_amdgcn_readfirstlane((int)src[k]);
        if (c == front) { if (lane == 0) dst[k] = 0; continue; }
        front = c;
        const u32 x = w ^ (c * 0x01010101u);
        const u32 hz = (x - 0x01010101u) & ~x & 0x80808080u;
        const u64 m = __ballot(hz != 0);
        const int lane0 = MTF_CTZ64(m);
        const u32 hz0 = (u32)__builtin_amdgcn_readlane((int)hz, lane0);
        const int byteIdx = MTF_CTZ32(hz0) >> 3;
        if (lane == 0) dst[k] = (u8)(4 * lane0 + byteIdx);
        w = mtf_rotate(w, lane, lane0, ((hz0 & (0u - hz0)) << 1) - 1u, c << 24);
    }
}

// ---- forward ranks, data parallel inside the tile (round 5) -------------------------------------------------------------------
// k_mtf_f_rank walks a tile byte by byte: one dependent chain of ~14 instructions per byte that changes the front of the list. The
// rank of a byte does not need the list, only counts: with j the previous occurrence of s[i],
//     rank(i) = number of distinct symbols in s(j, i)                      = #{k in (j, i) : prev(k) <= j}
// (k is the first occurrence of its symbol inside the window iff the occurrence before it lies at or in front of j). For a CHUNK of
// 64 consecutive bytes, one per lane, with v = (previous occurrence inside the chunk) + 1, 0 when there is none, this is
//     rank(i) = #{k < i : v(k) <= v(i)} - v(i)
// (every k <= j has v(k) <= k <= j, which is where the subtracted term comes from; a byte that repeats its predecessor gets 0 by the
// same formula) -- a count over earlier lanes with a smaller or equal 6-bit value: six ballots. A symbol that is new to the chunk
// starts from its place L in the list at the chunk's start and has been overtaken by the symbols that were behind it and came
// earlier in the chunk: rank(i) = L(s[i]) + #{first occurrences k < i : L(s[k]) > L(s[i])}, eight ballots over the lanes that are
// first occurrences. The list itself is only ever needed as "place of symbol c" (a 256-byte table per wave): after the chunk its
// distinct symbols stand in front in order of their last occurrence, and every other symbol has moved back by the number of chunk
// symbols that were behind it. About 300 instructions per 64 bytes, none of them on a chain longer than the chunk; the result is the
// reference's byte for byte (transform/SBRT.cpp:46-97), checked against the chain kernel and the oracle in the emulator.
__device__ __forceinline__ u32 mtf_popc64(u64 m) { return (u32)__popcll((unsigned long long)m); }

template <u32 MT>
__global__ __launch_bounds__(64) void k_mtf_f_rank_par(XfView v, int perTiles, const u32* __restrict__ tileState, u32 segT, u32 nSeg,
                                                       const u32* __restrict__ segMax)
{
    const int b = blockIdx.y;
    const u32 n = (v.len[b] <= v.cap[b]) ? v.len[b] : 0;
    const u32 tbase = blockIdx.x * MT;
    if (tbase >= n) return;
    const u8* s = v.src[b];
    u8* d = v.dst[b];
    __shared__ u32 keys[256];
    __shared__ u32 posW[64];                 // byte c = place of symbol c in the list
    __shared__ u32 Mw[8], Sx[8];             // places the chunk's symbols held (bit map), set bits in the words above
    __shared__ u32 tileW[MT / 4];            // the tile: symbols in, ranks out
    __shared__ u32 peerW[512];               // per symbol: the lanes of the chunk that show it (two words)
    const int lane = lane_id();
    const u32* st = tileState + ((size_t)b * perTiles + blockIdx.x) * 256;
    const u32* sg = segMax + ((size_t)b * nSeg + blockIdx.x / segT) * 256;
    for (int i = lane; i < 256; i += 64) { const u32 x = st[i], y = sg[i]; keys[i] = x > y ? x : y; }
    const u32 cnt = (n - tbase < MT) ? (n - tbase) : MT;
    const u8* src = s + tbase;
    u8* dst = d + tbase;
    const bool al = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3) == 0;
    u8* tileB = reinterpret_cast<u8*>(tileW);
    if (al) {
        for (u32 q = (u32)lane; q < cnt / 4; q += 64) tileW[q] = reinterpret_cast<const u32*>(src)[q];
        for (u32 q = (cnt & ~3u) + (u32)lane; q < cnt; q += 64) tileB[q] = src[q];            // (nothing is read behind the block's last byte)
    } else { for (u32 q = (u32)lane; q < cnt; q += 64) tileB[q] = src[q]; }
    __syncthreads();
    u8* posB = reinterpret_cast<u8*>(posW);
    // start list: symbols by last occurrence, latest first; never-seen symbols (key 0) ascending. Keys made unique first (a block has
    // at most 2^30 positions), so that a place is one compare per other symbol.
    for (int i = lane; i < 256; i += 64) { const u32 x = keys[i]; keys[i] = x ? x + 256u : 255u - (u32)i; }
    __syncthreads();
    {
        const u32 k0 = keys[lane], k1 = keys[lane + 64], k2 = keys[lane + 128], k3 = keys[lane + 192];
        u32 r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        for (int q = 0; q < 256; q++) {
            const u32 kq = keys[q];
            r0 += (kq > k0) ? 1u : 0u; r1 += (kq > k1) ? 1u : 0u; r2 += (kq > k2) ? 1u : 0u; r3 += (kq > k3) ? 1u : 0u;
        }
        posB[lane] = (u8)r0; posB[lane + 64] = (u8)r1; posB[lane + 128] = (u8)r2; posB[lane + 192] = (u8)r3;
    }
    __syncthreads();
    const u64 below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const u64 above = (lane == 63) ? 0ull : (~0ull << (lane + 1));
    // Run heads only (round 6). A byte that repeats its predecessor has rank 0 and leaves the list alone, so the ranks of a tile are the ranks
    // of its RUN HEADS (bytes that differ from the byte in front of them; the tile's first byte counts as one) with zeros in between -- and
    // behind a BWT seven bytes in ten are no heads. Lane c keeps the head flags of chunk c (a tile has at most 64 chunks); the heads are packed
    // to the front of the tile in place (a head never moves up), ranked by the chunk loop below as a tile of nH bytes, and spread out again
    // from the back. A tile with few repeats (under one byte in eight) is ranked as it stands.
    u64 headMask = 0;
    u32 headBase = 0;
    u32 cntR = cnt;
    {
        u32 carry = 0x100u;
        for (u32 k = 0, c = 0; k < cnt; k += 64, c++) {
            const bool valid = k + (u32)lane < cnt;
            const u32 bv = valid ? (u32)tileB[k + (u32)lane] : 0x200u;
            u32 pb = (u32)__shfl_up((int)bv, 1u, 64);
            if (lane == 0) pb = carry;
            const u64 m = __ballot(valid && bv != pb);
            if (lane == (int)c) headMask = m;
            carry = (u32)__builtin_amdgcn_readlane((int)bv, 63);
        }
        const u32 mine = mtf_popc64(headMask);
        const u32 incl = wave_incl_scan(mine);
        headBase = incl - mine;
        const u32 nH = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        if (nH * 8u <= cnt * 7u) {
            for (u32 k = 0, c = 0; k < cnt; k += 64, c++) {
                const u64 m = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(headMask >> 32), (int)c) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)(u32)headMask, (int)c);
                const u32 base = (u32)__builtin_amdgcn_readlane((int)headBase, (int)c);
                const bool head = (m >> lane) & 1ull;
                const u32 bv = head ? (u32)tileB[k + (u32)lane] : 0u;
                KNZ_WAVE_ORDER();                        // (every lane has read its byte before one is overwritten)
                if (head) tileB[base + mtf_popc64(m & below)] = (u8)bv;
            }
            cntR = nH;
            __syncthreads();
        }
    }
    const bool packed = cntR != cnt;
    for (u32 k = 0; k < cntR; k += 64) {
        const bool valid = k + (u32)lane < cntR;
        const u32 c = valid ? (u32)tileB[k + (u32)lane] : 0u;
        // the lanes that show my symbol: every lane sets its bit in the symbol's word pair in LDS (round 6; eight ballots with a 64-bit
        // select each were 120 of the chunk's 400 instructions)
        if (valid) { peerW[2 * c] = 0; peerW[2 * c + 1] = 0; }
        __syncthreads();
        if (valid) atomicOr(&peerW[2 * c + ((u32)lane >> 5)], 1u << (lane & 31));
        __syncthreads();
        const u64 peers = valid ? ((u64)peerW[2 * c] | ((u64)peerW[2 * c + 1] << 32)) : 0ull;
        const u64 pm = peers & below;
        const u32 vv = pm ? 64u - (u32)__clzll((long long)pm) : 0u;            // previous occurrence in the chunk + 1
        u32 rank;
        {
            u64 P = below;
            u32 C = 0;
#pragma unroll
            for (int bit = 5; bit >= 0; bit--) {
                const bool one = (vv >> bit) & 1u;
                const u64 B = __ballot(one);
                if (one) { C += mtf_popc64(P & ~B); P &= B; } else P &= ~B;
            }
            rank = C + mtf_popc64(P) - vv;
        }
        const bool first = valid && vv == 0;
        const u32 Lc = posB[c];
        const u64 F = __ballot(first);
        const u32 nF = mtf_popc64(F);
        if (nF <= 16) {
            // few new symbols (what the output of a BWT looks like): one step per new symbol, lanes behind it compare places
            u32 G = 0;
            u64 rem = F;
            while (rem) {
                const int kf = __builtin_ctzll((unsigned long long)rem);
                rem &= rem - 1;
                const u32 Lk = (u32)__builtin_amdgcn_readlane((int)Lc, kf);
                G += (Lk > Lc && lane > kf) ? 1u : 0u;
            }
            if (first) rank = Lc + G;
        } else {
            u64 Q = F & below;
            u32 G = 0;
#pragma unroll
            for (int bit = 7; bit >= 0; bit--) {
                const bool one = (Lc >> bit) & 1u;
                const u64 B = __ballot(first && one);
                if (one) Q &= B; else { G += mtf_popc64(Q & B); Q &= ~B; }
            }
            if (first) rank = Lc + G;
        }
        if (valid) tileB[k + (u32)lane] = (u8)rank;
        // The list behind the chunk. When the chunk's symbols held the front places anyway (the usual case behind a BWT: the same few
        // symbols come again and again) nobody else moves; else every symbol moves back by the chunk symbols that were behind it.
        if (__ballot(first && Lc >= nF) != 0) {
            if (lane < 8) Mw[lane] = 0;
            __syncthreads();
            if (first) atomicOr(&Mw[Lc >> 5], 1u << (Lc & 31));
            __syncthreads();
            if (lane < 8) { u32 a = 0; for (int w = lane + 1; w < 8; w++) a += (u32)__popc(Mw[w]); Sx[lane] = a; }
            __syncthreads();
            const u32 pw = posW[lane];                // places of the symbols 4 * lane .. 4 * lane + 3
            u32 npw = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const u32 p = (pw >> (8 * t)) & 0xFFu;
                const u32 w = p >> 5;
                const u32 add = Sx[w] + (u32)__popc(Mw[w] & ~((2u << (p & 31)) - 1u));
                npw |= ((p + add) & 0xFFu) << (8 * t);
            }
            posW[lane] = npw;
            __syncthreads();
        }
        const bool last = valid && (peers & above) == 0;
        const u64 Lm = __ballot(last);
        if (last) posB[c] = (u8)mtf_popc64(Lm & above);
        __syncthreads();
    }
    __syncthreads();
    if (packed) {
        // the ranks of the heads back to their places, last chunk first: chunk c reads tile bytes below 64 c + 64 and writes [64 c, 64 c + 64),
        // the chunks in front of it read below 64 c
        for (int c = (int)((cnt + 63u) / 64u) - 1; c >= 0; c--) {
            const u64 m = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(headMask >> 32), c) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)(u32)headMask, c);
            const u32 base = (u32)__builtin_amdgcn_readlane((int)headBase, c);
            const bool head = (m >> lane) & 1ull;
            u32 r = 0;
            if (head) r = (u32)tileB[base + mtf_popc64(m & below)];
            KNZ_WAVE_ORDER();
            if (64u * (u32)c + (u32)lane < cnt) tileB[64u * (u32)c + (u32)lane] = (u8)r;
            KNZ_WAVE_ORDER();
        }
        __syncthreads();
    }
    if (al && (cnt & 3) == 0) { for (u32 q = (u32)lane; q < cnt / 4; q += 64) reinterpret_cast<u32*>(dst)[q] = tileW[q]; }
    else { for (u32 q = (u32)lane; q < cnt; q += 64) dst[q] = tileB[q]; }
}

// inverse, pass 1: symbolic decode from the identity list; ids -> dst, final list (ids) -> tilePerm
template <u32 MT>
__global__ __launch_bounds__(64) void k_mtf_i_symbolic(XfView v, int perTiles, u8* __restrict__ tilePerm)
{
    const int b = blockIdx.y;
    const u32 n = (v.len[b] <= v.cap[b]) ? v.len[b] : 0;
    const u32 tbase = blockIdx.x * MT;
    if (tbase >= n) return;
    const u8* src = v.src[b] + tbase;
    u8* dst = v.dst[b] + tbase;
    const int lane = lane_id();
    u32 w = (u32)(4 * lane) | ((u32)(4 * lane + 1) << 8) | ((u32)(4 * lane + 2) << 16) | ((u32)(4 * lane + 3) << 24);
    u32 front = 0;                                   // id at list position 0 (uniform)
    const u32 cnt = (n - tbase < MT) ? (n - tbase) : MT;
    const bool al = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3) == 0;
    // four ranks (uniform) -> four list ids
    auto ids4 = [&](u32 in4) -> u32 {
        u32 out4 = 0;
        if (in4 == 0) out4 = front * 0x01010101u;       // rank 0 four times: the front symbol repeats
        else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32 r = (in4 >> (8 * q)) & 0xFF;
                if (r == 0) { out4 |= front << (8 * q); continue; }
                const int lane0 = (int)(r >> 2);
                const int byteIdx = (int)(r & 3);
                // (the byte is cut out on the vector unit, in every lane, before lane0's copy is read: two scalar instructions less)
                const u32 c = (u32)__builtin_amdgcn_readlane((int)((w >> (8 * byteIdx)) & 0xFFu), lane0);
                out4 |= c << (8 * q);
                front = c;
                w = mtf_rotate(w, lane, lane0, (u32)((1ull << (8 * (byteIdx + 1))) - 1ull), mtf_top(c));
            }
        }
        return out4;
    };
    u32 k = 0;
    if (al && cnt == MT) {
        u32 inr[MT / 256], outr[MT / 256];
#pragma unroll
        for (int t = 0; t < (int)(MT / 256); t++) inr[t] = reinterpret_cast<const u32*>(src)[64 * t + lane];
        // Only the words that hold a rank other than 0 walk the chain (round 6: behind a BWT two words in three are four zeros, and each cost
        // the chain a readlane, a compare and a select); a word of zeros repeats the front as it stood, which is the last id of the nearest
        // word in front of it that went through the chain, or the front the row began with.
        const u64 belowL = lane ? (~0ull >> (64 - lane)) : 0ull;
#pragma unroll
        for (int t = 0; t < (int)(MT / 256); t++) {
            const u32 x = inr[t];
            const u64 nz = __ballot(x != 0);
            const u32 frontIn = front;
            u32 acc = 0;
            for (u64 m = nz; m != 0; m &= m - 1) {
                const int l = MTF_CTZ64(m);
                const u32 o = ids4((u32)__builtin_amdgcn_readlane((int)x, l));
                acc = (lane == l) ? o : acc;
            }
            const u64 bel = nz & belowL;
            const int from = bel ? 63 - __clzll((long long)bel) : 0;
            const u32 prevTop = (u32)__shfl((int)acc, from, 64) >> 24;
            outr[t] = (x != 0) ? acc : (bel ? prevTop : frontIn) * 0x01010101u;
        }
#pragma unroll
        for (int t = 0; t < (int)(MT / 256); t++) reinterpret_cast<u32*>(dst)[64 * t + lane] = outr[t];
        k = MT;
    }
    for (; k + 4 <= cnt; k += 4) {
        u32 in4;
        if (al) in4 = *reinterpret_cast<const u32*>(src + k);
        else in4 = (u32)src[k] | ((u32)src[k + 1] << 8) | ((u32)src[k + 2] << 16) | ((u32)src[k + 3] << 24);
        in4 = (u32)__builtin_amdgcn_readfirstlane((int)in4);
        const u32 out4 = ids4(in4);
        if (lane == 0) {
            if (al) *reinterpret_cast<u32*>(dst + k) = out4;
            else { dst[k] = (u8)out4; dst[k + 1] = (u8)(out4 >> 8); dst[k + 2] = (u8)(out4 >> 16); dst[k + 3] = (u8)(out4 >> 24); }
        }
    }
    for (; k < cnt; k++) {
        const u32 r = (u32)__builtin_amdgcn_readfirstlane((int)src[k]);
        if (r == 0) { if (lane == 0) dst[k] = (u8)front; continue; }
        const int lane0 = (int)(r >> 2);
        const int byteIdx = (int)(r & 3);
        const u32 wl = (u32)__builtin_amdgcn_readlane((int)w, lane0);
        const u32 c = (wl >> (8 * byteIdx)) & 0xFF;
        if (lane == 0) dst[k] = (u8)c;
        front = c;
        w = mtf_rotate(w, lane, lane0, (u32)((1ull << (8 * (byteIdx + 1))) - 1ull), c << 24);
    }
    reinterpret_cast<u32*>(tilePerm + ((size_t)b * perTiles + blockIdx.x) * 256)[lane] = w;
}

// ---- inverse, pass 1, data parallel inside the tile (round 6) ---------------------------------------------------------------------
// k_mtf_i_symbolic walks the tile rank by rank: about 21 instructions (half of them scalar) per rank that is not 0, one dependent chain.
// A rank 0 repeats the id in front of it and leaves the list alone, so the tile is decoded from its EVENTS (ranks other than 0, three
// bytes in ten behind a BWT), 64 at a time, one per lane, and no lane waits for another:
//   * lane i wants the id at place q = rank_i of the list as it stands before event i. Going back over the events j = i-1 .. 0 of the
//     chunk: event j moved place rank_j to the front and places 0 .. rank_j - 1 one back, so a place q of the list behind event j was
//     place q - 1 in front of it when 1 <= q <= rank_j, place q when q > rank_j, and q = 0 is the id event j itself delivered. A lane
//     ends with "the id of event j" or with a place of the list at the chunk's start; every lane runs the same <= 63 steps of one
//     readlane, two compares, a select and a conditional decrement;
//   * ids copied from an earlier event are resolved by pointer jumping over the lanes (six shuffles at most);
//   * the list behind the chunk: its distinct ids by last occurrence, latest first, then the places no event touched in their old order
//     -- when the touched places were the front ones anyway (the usual case: the same few symbols come again and again), only those
//     are rewritten.
// The ids of the events are packed at the front of the tile in LDS and spread out again from the back, every byte in between
// repeating the event in front of it (the forward kernel's run heads, mirrored). Checked against the chain kernel and the oracle in
// the emulator (tests/emu/mtft_emu.cpp, KNZ_MTF_CHAIN=0/1) and on the device by every MTFT case of the GPU suite.
template <u32 MT>
__global__ __launch_bounds__(64) void k_mtf_i_symbolic_par(XfView v, int perTiles, u8* __restrict__ tilePerm)
{
    const int b = blockIdx.y;
    const u32 n = (v.len[b] <= v.cap[b]) ? v.len[b] : 0;
    const u32 tbase = blockIdx.x * MT;
    if (tbase >= n) return;
    const u8* src = v.src[b] + tbase;
    u8* dst = v.dst[b] + tbase;
    const int lane = lane_id();
    __shared__ u32 tileW[MT / 4];            // ranks in, ids out
    __shared__ u32 listW[2][64];             // place -> id, two copies in turn
    __shared__ u32 Mw[8], Sx[8];             // places the chunk's events took their ids from (bit map), set bits in the words above
    __shared__ u32 lastW[256];               // per id: 1 + the last lane of the chunk that delivered it
    u8* tileB = reinterpret_cast<u8*>(tileW);
    const u32 cnt = (n - tbase < MT) ? (n - tbase) : MT;
    const bool al = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3) == 0;
    if (al) {
        for (u32 q = (u32)lane; q < cnt / 4; q += 64) tileW[q] = reinterpret_cast<const u32*>(src)[q];
        for (u32 q = (cnt & ~3u) + (u32)lane; q < cnt; q += 64) tileB[q] = src[q];
    } else { for (u32 q = (u32)lane; q < cnt; q += 64) tileB[q] = src[q]; }
    listW[0][lane] = (u32)(4 * lane) | ((u32)(4 * lane + 1) << 8) | ((u32)(4 * lane + 2) << 16) | ((u32)(4 * lane + 3) << 24);
    __syncthreads();
    const u64 below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const u64 above = (lane == 63) ? 0ull : (~0ull << (lane + 1));
    // ---- events of every chunk (lane c keeps the flags of chunk c), packed to the front of the tile
    u64 evMask = 0;
    for (u32 k = 0, c = 0; k < cnt; k += 64, c++) {
        const bool valid = k + (u32)lane < cnt;
        const u32 r = valid ? (u32)tileB[k + (u32)lane] : 0u;
        const u64 m = __ballot(r != 0);
        if (lane == (int)c) evMask = m;
    }
    const u32 evMine = mtf_popc64(evMask);
    const u32 evIncl = wave_incl_scan(evMine);
    const u32 evBase = evIncl - evMine;
    const u32 nEv = (u32)__builtin_amdgcn_readlane((int)evIncl, 63);
    for (u32 k = 0, c = 0; k < cnt; k += 64, c++) {
        const u64 m = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(evMask >> 32), (int)c) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)(u32)evMask, (int)c);
        const u32 base = (u32)__builtin_amdgcn_readlane((int)evBase, (int)c);
        const bool ev = (m >> lane) & 1ull;
        const u32 r = ev ? (u32)tileB[k + (u32)lane] : 0u;
        KNZ_WAVE_ORDER();                            // (every lane has read its byte before one is overwritten)
        if (ev) tileB[base + mtf_popc64(m & below)] = (u8)r;
    }
    __syncthreads();
    int cur = 0;
    for (u32 k = 0; k < nEv; k += 64) {
        u8* listB = reinterpret_cast<u8*>(listW[cur]);
        const u32 nv = (nEv - k < 64u) ? nEv - k : 64u;           // events of this chunk (uniform)
        const bool valid = (u32)lane < nv;
        const u32 rho = valid ? (u32)tileB[k + (u32)lane] : 0u;     // >= 1 for an event
        // ---- back over the earlier events of the chunk
        // (step t looks at event lane - t: the ranks move up one lane per step, 0 comes in below the chunk's first event. The place is a
        // SIGNED number: a lane whose place has come to 0 has found its event and counts the steps since below 0 -- 0 and everything
        // under it is "<= rank" for every rank --, so a step is a DPP move, a compare and a conditional decrement and the step of the
        // find is read off at the end. Past its first event -- rank 0 from below -- a lane does nothing, except that a place 0 "finds"
        // an event in front of the chunk, which the step number tells. The trip count is uniform and rounded up: extra steps see rank 0)
        int q = (int)rho, rs = (int)rho;
        const u32 steps = ((u32)__builtin_amdgcn_readfirstlane((int)nv) + 2u) & ~3u;      // >= nv - 1, a multiple of 4
        for (u32 t = 0; t < steps; t += 4) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                rs = __builtin_amdgcn_update_dpp(0, rs, 0x138, 0xF, 0xF, true);           // wave_shr:1, 0 into lane 0
                q -= (q <= rs) ? 1 : 0;
            }
        }
        const u32 hitT = (q < 0) ? (u32)((int)steps + 1 + q) : 0u;                        // the step that began with place 0
        int from = (hitT != 0 && hitT <= (u32)lane) ? lane - (int)hitT : -1;            // the event whose id this one repeats
        q = (q < 0) ? 0 : q;                                                              // (found in front of the chunk: place 0 of the list)
        const bool fromList = valid && from < 0;
        u32 id = fromList ? (u32)listB[(u32)q] : 0u;
        // ---- ids that repeat an earlier event of the chunk
        for (int it = 0; it < 6; it++) {
            if (__ballot(valid && from >= 0) == 0) break;
            const int at = from < 0 ? lane : from;
            const u32 tid2 = (u32)__shfl((int)id, at, 64);
            const int tf = __shfl(from, at, 64);
            if (from >= 0) { if (tf < 0) { id = tid2; from = -1; } else from = tf; }
        }
        KNZ_WAVE_ORDER();
        if (valid) tileB[k + (u32)lane] = (u8)id;
        // ---- the list behind the chunk
        // (the last event of every id: the highest lane that shows it, through one LDS maximum per lane)
        lastW[lane] = 0; lastW[lane + 64] = 0; lastW[lane + 128] = 0; lastW[lane + 192] = 0;
        __syncthreads();
        if (valid) atomicMax(&lastW[id], (u32)lane + 1u);
        __syncthreads();
        const bool last = valid && lastW[id] == (u32)lane + 1u;
        const u64 Lm = __ballot(last);
        const u32 D = mtf_popc64(Lm);                              // distinct ids of the chunk = places that gave an id
        const u32 newPlace = mtf_popc64(Lm & above);
        if (__ballot(fromList && (u32)q >= D) == 0) {
            // the ids came from the front D places: nobody else moves
            if (last) listB[newPlace] = (u8)id;
        } else {
            u8* nextB = reinterpret_cast<u8*>(listW[cur ^ 1]);
            if (lane < 8) Mw[lane] = 0;
            __syncthreads();
            if (fromList) atomicOr(&Mw[(u32)q >> 5], 1u << ((u32)q & 31));
            __syncthreads();
            if (lane < 8) { u32 a = 0; for (int w = lane + 1; w < 8; w++) a += (u32)__popc(Mw[w]); Sx[lane] = a; }
            __syncthreads();
            const u32 lw = listW[cur][lane];                       // ids at the places 4 * lane .. 4 * lane + 3
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const u32 p = 4u * (u32)lane + (u32)t;
                const u32 w = p >> 5;
                const u32 mw = Mw[w];
                if ((mw >> (p & 31)) & 1u) continue;               // this place gave its id to an event
                const u32 add = Sx[w] + (u32)__popc(mw & ~((2u << (p & 31)) - 1u));
                nextB[p + add] = (u8)(lw >> (8 * t));
            }
            if (last) nextB[newPlace] = (u8)id;
            cur ^= 1;
        }
        __syncthreads();
    }
    // ---- every byte of the tile: the id of the last event at or in front of it (none: the list's first id, 0)
    for (int c = (int)((cnt + 63u) / 64u) - 1; c >= 0; c--) {
        const u64 m = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(evMask >> 32), c) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)(u32)evMask, c);
        const u32 base = (u32)__builtin_amdgcn_readlane((int)evBase, c);
        const u32 upto = base + mtf_popc64(m & (below | (1ull << lane)));       // events up to and including this byte
        const u32 id = upto ? (u32)tileB[upto - 1] : 0u;
        KNZ_WAVE_ORDER();
        if (64u * (u32)c + (u32)lane < cnt) tileB[64u * (u32)c + (u32)lane] = (u8)id;
        KNZ_WAVE_ORDER();
    }
    __syncthreads();
    if (al && (cnt & 3) == 0) { for (u32 q = (u32)lane; q < cnt / 4; q += 64) reinterpret_cast<u32*>(dst)[q] = tileW[q]; }
    else { for (u32 q = (u32)lane; q < cnt; q += 64) dst[q] = tileB[q]; }
    reinterpret_cast<u32*>(tilePerm + ((size_t)b * perTiles + blockIdx.x) * 256)[lane] = listW[cur][lane];
}

// inverse, pass 2: state before tile t: S_0 = identity, S_{t+1}[j] = S_t[perm_t[j]]. Composition is associative, so it runs in two
// levels like the forward scan: inside a segment from the identity (L_t, stored in place, and the segment's total in segPerm), then
// over the segment totals (A_g = state before segment g, in place); the state a tile needs is S_t[j] = A_g[L_t[j]] (pass 3).
template <u32 MT>
__global__ __launch_bounds__(256) void k_mtf_i_compose(u8* __restrict__ tilePerm, int perTiles, u32 segT, u32 nSeg, const u32* __restrict__ lens,
                                                       u8* __restrict__ segPerm)
{
    const int b = blockIdx.y;
    const u32 seg = blockIdx.x;
    const u32 cnt = (lens[b] + MT - 1) / MT;
    const u32 t0 = seg * segT;
    const u32 t1 = (t0 + segT < cnt) ? t0 + segT : cnt;
    __shared__ u8 S[2][256];
    S[0][threadIdx.x] = (u8)threadIdx.x;
    __syncthreads();
    u8* p = tilePerm + (size_t)b * perTiles * 256;
    int cur = 0;
    u8 pj = (t0 < t1) ? p[(size_t)t0 * 256 + threadIdx.x] : (u8)0;
    for (u32 t = t0; t < t1; t++) {
        const u8 pjNext = (t + 1 < t1) ? p[(size_t)(t + 1) * 256 + threadIdx.x] : (u8)0;   // prefetch: independent of the chain
        const u8 nv = S[cur][pj];
        p[(size_t)t * 256 + threadIdx.x] = S[cur][threadIdx.x];   // state before tile t, relative to the segment start
        S[cur ^ 1][threadIdx.x] = nv;
        __syncthreads();
        cur ^= 1;
        pj = pjNext;
    }
    segPerm[((size_t)b * nSeg + seg) * 256 + threadIdx.x] = S[cur][threadIdx.x];
}

__global__ __launch_bounds__(256) void k_mtf_i_compose2(u8* __restrict__ segPerm, u32 nSeg)
{
    __shared__ u8 S[2][256];
    S[0][threadIdx.x] = (u8)threadIdx.x;
    __syncthreads();
    u8* p = segPerm + (size_t)blockIdx.x * nSeg * 256;
    int cur = 0;
    for (u32 g = 0; g < nSeg; g++) {
        const u8 pj = p[(size_t)g * 256 + threadIdx.x];
        const u8 nv = S[cur][pj];
        p[(size_t)g * 256 + threadIdx.x] = S[cur][threadIdx.x];   // state before segment g
        S[cur ^ 1][threadIdx.x] = nv;
        __syncthreads();
        cur ^= 1;
    }
}

// inverse, pass 3: resolve ids in place: out[i] = state_tile[id]
template <u32 MT>
__global__ __launch_bounds__(256) void k_mtf_i_resolve(XfView v, int perTiles, const u8* __restrict__ tileState, u32 segT, u32 nSeg,
                                                       const u8* __restrict__ segState)
{
    const int b = blockIdx.y;
    const u32 n = (v.len[b] <= v.cap[b]) ? v.len[b] : 0;
    const u32 t = blockIdx.x;
    const u32 base = t * MT;
    if (base >= n) return;
    __shared__ u8 S[256];
    S[threadIdx.x] = segState[((size_t)b * nSeg + t / segT) * 256 + tileState[((size_t)b * perTiles + t) * 256 + threadIdx.x]];
    __syncthreads();
    u8* d = v.dst[b];
    const u32 end = (base + MT < n) ? base + MT : n;
    if (((reinterpret_cast<uintptr_t>(d) & 3) == 0) && end == base + MT) {
        u32* d4 = reinterpret_cast<u32*>(d + base);
        for (u32 k = threadIdx.x; k < MT / 4; k += 256) {
            const u32 x = d4[k];
            d4[k] = (u32)S[x & 0xFF] | ((u32)S[(x >> 8) & 0xFF] << 8) | ((u32)S[(x >> 16) & 0xFF] << 16) | ((u32)S[x >> 24] << 24);
        }
    } else {
        for (u32 i = base + threadIdx.x; i < end; i += 256) d[i] = S[d[i]];
    }
}

__global__ void k_copy_ok(const u32* __restrict__ lens, const u32* __restrict__ caps, int nBlocks, u8* ok, u32* newLen)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    ok[b] = (lens[b] <= caps[b]) ? 1 : 0;       // SBRT.cpp:57-60
    newLen[b] = lens[b];
}

static XfView mk(const XfStage& st) { XfView v; v.src = st.src; v.dst = st.dst; v.len = st.len; v.cap = st.cap; return v; }

template <u32 MT>
static void mtft_forward_t(hipStream_t s, const XfStage& st)
{
    const XfView v = mk(st);
    const int perTiles = (int)((st.maxLen + MT - 1) / MT);
    u32* tileLast = st.scratchU32;                           // nBlocks * perTiles * 256
    const u32 segT = mtf_seg_tiles((u32)perTiles), nSeg = ((u32)perTiles + segT - 1) / segT;
    u32* segMax = tileLast + (size_t)st.nBlocks * perTiles * 256;                  // nBlocks * nSeg * 256
    const dim3 grid(perTiles, st.nBlocks);
    { KScope ks_("k_copy_ok"); hipLaunchKernelGGL(k_copy_ok, dim3((st.nBlocks + 255) / 256), dim3(256), 0, s, st.len, st.cap, st.nBlocks, st.ok, st.newLen); }
    { KScope ks_("k_mtf_f_last"); hipLaunchKernelGGL((k_mtf_f_last<MT>), grid, dim3(64), 0, s, v, perTiles, tileLast); }
    { KScope ks_("k_mtf_f_scan"); hipLaunchKernelGGL((k_mtf_f_scan<MT>), dim3(nSeg, st.nBlocks), dim3(256), 0, s, tileLast, perTiles, segT, nSeg, st.len, segMax);
      hipLaunchKernelGGL(k_mtf_f_scan2, dim3(st.nBlocks), dim3(256), 0, s, segMax, nSeg); }
    { KScope ks_("k_mtf_f_rank");
      if (g_mtfChainKnob.load()) hipLaunchKernelGGL((k_mtf_f_rank<MT>), grid, dim3(64), 0, s, v, perTiles, tileLast, segT, nSeg, segMax);
      else hipLaunchKernelGGL((k_mtf_f_rank_par<MT>), grid, dim3(64), 0, s, v, perTiles, tileLast, segT, nSeg, segMax); }
}

void launch_mtft_forward(hipStream_t s, const XfStage& st)
{
    if (mtf_tile_bytes(st.nBlocks, st.maxLen) == 1024u) mtft_forward_t<1024>(s, st);
    else mtft_forward_t<4096>(s, st);
}

template <u32 MT>
static void mtft_inverse_t(hipStream_t s, const XfStage& st)
{
    const XfView v = mk(st);
    const int perTiles = (int)((st.maxLen + MT - 1) / MT);
    u8* tilePerm = reinterpret_cast<u8*>(st.scratchU32);     // nBlocks * perTiles * 256 bytes
    const u32 segT = mtf_seg_tiles((u32)perTiles), nSeg = ((u32)perTiles + segT - 1) / segT;
    u8* segPerm = tilePerm + (size_t)st.nBlocks * perTiles * 256;                  // nBlocks * nSeg * 256 bytes
    { KScope ks_("k_copy_ok"); hipLaunchKernelGGL(k_copy_ok, dim3((st.nBlocks + 255) / 256), dim3(256), 0, s, st.len, st.cap, st.nBlocks, st.ok, st.newLen); }
    { KScope ks_("k_mtf_i_symbolic");
      if (g_mtfChainKnob.load()) hipLaunchKernelGGL((k_mtf_i_symbolic<MT>), dim3(perTiles, st.nBlocks), dim3(64), 0, s, v, perTiles, tilePerm);
      else hipLaunchKernelGGL((k_mtf_i_symbolic_par<MT>), dim3(perTiles, st.nBlocks), dim3(64), 0, s, v, perTiles, tilePerm); }
    { KScope ks_("k_mtf_i_compose"); hipLaunchKernelGGL((k_mtf_i_compose<MT>), dim3(nSeg, st.nBlocks), dim3(256), 0, s, tilePerm, perTiles, segT, nSeg, st.len, segPerm);
      hipLaunchKernelGGL(k_mtf_i_compose2, dim3(st.nBlocks), dim3(256), 0, s, segPerm, nSeg); }
    { KScope ks_("k_mtf_i_resolve"); hipLaunchKernelGGL((k_mtf_i_resolve<MT>), dim3(perTiles, st.nBlocks), dim3(256), 0, s, v, perTiles, tilePerm, segT, nSeg, segPerm); }
}

void launch_mtft_inverse(hipStream_t s, const XfStage& st)
{
    if (mtf_tile_bytes(st.nBlocks, st.maxLen) == 1024u) mtft_inverse_t<1024>(s, st);
    else mtft_inverse_t<4096>(s, st);
}

size_t mtft_scratch_u32(int nBlocks, u32 maxLen)
{
    // sized for the smaller tile (the larger table) whatever the launch will pick: the knob may change between this call and the
    // launch (another thread's knz_hip_tune), and the tables must hold either choice
    const u32 MT = 1024u;
    (void)mtf_tile_bytes;
    const size_t perTiles = ((size_t)maxLen + MT - 1) / MT;
    const u32 segT = mtf_seg_tiles((u32)perTiles);
    return (size_t)nBlocks * (perTiles + (perTiles + segT - 1) / segT) * 256 + 64;    // tile tables + segment tables
}

}  // namespace knz
