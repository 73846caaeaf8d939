// rANS order-0 (kanzi "ANS0") on gfx950.
//
// Reference being replaced (results are bit-identical):
//   encoder  entropy/ANSRangeEncoder.cpp:158-192 (encode), :264-287 (rebuildStatistics), :83-116
//            (updateFrequencies), :119-155 (encodeHeader), :194-261 (encodeChunk),
//            entropy/ANSRangeEncoder.hpp:92-131 ; entropy/EntropyUtils.cpp:57-89,131-245
//   decoder  entropy/ANSRangeDecoder.cpp:80-175 (decodeHeader), :177-216 (decode), :218-292 (decodeChunk)
//
// Mapping (not a translation of the CPU loops); the decoder lives in ans_dec.hip, order 1 in ans1.hip:
//   k_ans0_stats   one wave per 16 KiB chunk: 8 lane-selected private LDS histograms from 16 B/lane coalesced
//                  loads, wave-parallel normalizeFrequencies (4 symbols per lane, shuffle reductions and
//                  prefix sums reproduce the reference's order-dependent error spreading), encoder
//                  table (reciprocals) and the bit-granular chunk header built with LDS atomicOr
//                  (the table/header code is shared with order 1: ans_common.hpp).
//   k_ans0_encode  4 lanes per chunk = the 4 interleaved rANS states, 16 chunks per wave. The shared
//                  backward byte pointer of the reference becomes a ballot + popcount per step; input
//                  dwords are read one 16-step interval ahead, the table entry of step s+1 before step s
//                  is processed, and emissions are staged in LDS and written out as 128-byte lines.
#include "common.hpp"
#include "stages.hpp"
#include "ans_common.hpp"

namespace knz {

constexpr u32 ANS_TOP = 1u << 15;
constexpr u32 ANS_LR = 12;
constexpr u32 ANS_SCALE = 1u << ANS_LR;

// ------------------------------------------------------------------------------------------------
// stats + header
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ans0_stats(BlockView view, int maxChunks, ChunkDesc* __restrict__ desc,
                                                   uint2* __restrict__ encTab, u8* __restrict__ tmp)
{
    const int slot = blockIdx.x;
    const int b = slot / maxChunks;
    const int ci = slot - b * maxChunks;
    const u32 len = view.len[b];
    const u32 start = (u32)ci * ENT_CHUNK;
    if (start >= len) return;
    const int lane = lane_id();
    const u8* blk = view.ptr[b] + start;
    ChunkDesc* cd = desc + slot;

    if (len <= 32) {
        // ANSRangeEncoder.cpp:160-163 : tiny block stored raw (also the <= 15 byte copy-block mode)
        if (lane == 0) {
            cd->hdrBits = 0; cd->midLen = 0; cd->trailerLen = 0; cd->aux = 0;
            cd->nPieces = 1; cd->pieceBits[0] = 8 * len; cd->piecePtr[0] = blk;
        }
        return;
    }
    const u32 n = (len - start < ENT_CHUNK) ? (len - start) : ENT_CHUNK;

    // 8 private histograms, chosen by lane: one ds_add instruction never sends more than 8 lanes to the same
    // copy, which is what bounds the same-address serialisation on skewed data (text: 15 % spaces)
    __shared__ u32 hist[8][264];                 // row stride 264: copy c of a symbol sits 8c banks away, not in the same bank
    __shared__ u32 hdrw[HDR_WORDS];
    __shared__ u32 grpMax[64];
    for (int i = lane; i < 8 * 264; i += 64) (&hist[0][0])[i] = 0;
    for (int i = lane; i < (int)HDR_WORDS; i += 64) hdrw[i] = 0;
    grpMax[lane] = 0;
    __syncthreads();

    // ---- histogram (Global.cpp:170-221): 16 bytes per lane per iteration
    u32* myHist = hist[lane & 7];
    const u32 n16 = n & ~15u;
    const bool aligned = ((reinterpret_cast<uintptr_t>(blk) & 15) == 0);
    if (aligned) {
        const uint4* p4 = reinterpret_cast<const uint4*>(blk);
        for (u32 i = lane; i < (n16 >> 4); i += 64) {
            const uint4 v = p4[i];
            const u32 w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                atomicAdd(&myHist[w[k] & 0xFF], 1u);
                atomicAdd(&myHist[(w[k] >> 8) & 0xFF], 1u);
                atomicAdd(&myHist[(w[k] >> 16) & 0xFF], 1u);
                atomicAdd(&myHist[w[k] >> 24], 1u);
            }
        }
    } else {
        for (u32 i = lane; i < n16; i += 64) atomicAdd(&myHist[blk[i]], 1u);
    }
    for (u32 i = n16 + lane; i < n; i += 64) atomicAdd(&myHist[blk[i]], 1u);
    __syncthreads();

    // lane owns symbols 4*lane .. 4*lane+3
    u32 f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int s = 4 * lane + k;
        u32 acc = 0;
#pragma unroll
        for (int c = 0; c < 8; c++) acc += hist[c][s];
        f[k] = acc;
    }
    u32 present = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) present |= (f[k] != 0 ? 1u : 0u) << k;
    const u32 myCount = __popc(present);
    const u32 inclCount = wave_incl_scan(myCount);
    const u32 asz = (u32)__shfl((int)inclCount, 63, 64);
    const u32 rankBase = inclCount - myCount;      // alphabet index of my first present symbol

    if (n != ANS_SCALE) ans_normalize<ANS_LR>(lane, f, n, asz);

    __shared__ u32 grpOff[64];
    const u32 pos = ans_header_bits<ANS_LR>(lane, f, present, asz, rankBase, true, hdrw, grpMax, grpOff, 0);
    __syncthreads();

    // ---- write header buffer + encoder table
    u32* hdrOut = reinterpret_cast<u32*>(tmp + (size_t)slot * TMP_STRIDE);
    const u32 hdrWordsUsed = (pos + 31) >> 5;
    for (u32 i = lane; i < hdrWordsUsed; i += 64) hdrOut[i] = bswap32(hdrw[i]);

    if (asz > 1) ans_enc_table<ANS_LR>(lane, f, encTab + (size_t)slot * 256);
    if (lane == 0) {
        cd->hdrBits = pos; cd->midLen = 0; cd->trailerLen = 0; cd->nPieces = 0; cd->aux = asz;
    }
}

// ------------------------------------------------------------------------------------------------
// encode: 16 chunks per wave, 4 lanes (= 4 rANS states) per chunk
// ------------------------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ u32 quad_bcast(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xF, 0xF, false);
}

__global__ __launch_bounds__(64) void k_ans0_encode(BlockView view, int maxChunks, int nSlots, ChunkDesc* __restrict__ desc,
                                                    const uint2* __restrict__ encTab, u8* __restrict__ tmp)
{
    __shared__ uint2 tab[16][257];                   // 32 KiB; odd row stride: the same symbol of different chunks sits in different banks
    __shared__ uint4 stage[16][16];                  // 256 B of staged output per chunk
    const int lane = lane_id();
    const int g = lane >> 2;
    const int j = lane & 3;
    const int slotBase = blockIdx.x * 16;
    const int slot = slotBase + g;

    bool act = false;
    u32 n = 0;
    const u8* blk = nullptr;
    if (slot < nSlots) {
        const int b = slot / maxChunks;
        const int ci = slot - b * maxChunks;
        const u32 len = view.len[b];
        const u32 start = (u32)ci * ENT_CHUNK;
        if (len > 32 && start < len && desc[slot].aux > 1) {
            act = true;
            n = (len - start < ENT_CHUNK) ? (len - start) : ENT_CHUNK;
            blk = view.ptr[b] + start;
        }
    }
    // cooperative table load (all 16 groups)
    const u64 actMask = __ballot(act);
    for (int gg = 0; gg < 16; gg++) {
        if (!((actMask >> (4 * gg)) & 1)) continue;
        const uint2* src = encTab + (size_t)(slotBase + gg) * 256;
#pragma unroll
        for (int k = 0; k < 4; k++) tab[gg][lane + 64 * k] = src[lane + 64 * k];
    }
    __syncthreads();
    if (actMask == 0) return;

    const u32 end4 = n & ~3u;
    const u32 tail = n - end4;
    u8* pay = tmp + (size_t)slot * TMP_STRIDE + HDR_BYTES;
    // exclusive end of the payload area; shifted by one when the raw tail is odd so that every
    // 16-bit emission lands on an even address
    const u32 top = PAY_BYTES - (tail & 1);
    // Emissions are staged in a 256-byte LDS ring per chunk (payload offset & 255) and written out in
    // 128-byte lines by the chunk's own 4 lanes; the input is read 16 steps ahead (4 dwords per lane).
    u8* stg = reinterpret_cast<u8*>(stage[g]);
    if (act && j == 0) {
        for (u32 t = 0; t < tail; t++) stg[(top - tail + t) & 255] = blk[end4 + t];   // ANSRangeEncoder.cpp:204-205
    }
    u32 q = top - tail;                               // current (exclusive) low end of emitted bytes
    u32 Hi = PAY_BYTES;                               // payload bytes [Hi, PAY_BYTES) are already in global memory
    u32 st = ANS_TOP;
    const u32 steps = act ? (end4 >> 2) : 0;
    const u32 maxSteps = wave_max(steps);
    const uint2* mytab = tab[g];
    const u32 grpShift = (u32)(lane & ~3);
    const u32 lowMask = (1u << j) - 1u;
    const u32 jsh = 8u * (3u - (u32)j);
    // step s encodes dword (end4/4 - 1 - s); lanes without a chunk read a harmless valid address instead
    const u32* blkw = act ? reinterpret_cast<const u32*>(blk) : reinterpret_cast<const u32*>(encTab);
    const int nDw = act ? (int)(end4 >> 2) : 0;
    // dwords of interval k (steps 16k .. 16k+15): lane j holds dwords nDw - 16(k+1) + 4j + c, c = 0..3 (clamped at 0)
    auto load_interval = [&](u32 k, u32 r[4]) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            int idx = nDw - 16 * (int)(k + 1) + 4 * (int)j + c;
            idx = idx < 0 ? 0 : idx;
            r[c] = blkw[idx];
        }
    };
    auto flush_line = [&]() {
        // payload bytes [Hi - 128, Hi) are final: lane j writes 32 of them
        const u32 off = Hi - 128 + 32u * (u32)j;
        const uint4 v0 = *reinterpret_cast<const uint4*>(stg + (off & 255));
        const uint4 v1 = *reinterpret_cast<const uint4*>(stg + ((off + 16) & 255));
        *reinterpret_cast<uint4*>(pay + off) = v0;
        *reinterpret_cast<uint4*>(pay + off + 16) = v1;
    };
    // one step with the table entry already in registers (the entry of the next step is fetched before this
    // one is processed, so the LDS latency is off the state chain)
    auto step = [&](u32 s, uint2 e) {
        const bool on = s < steps;
        const u32 fr = e.y & 0x1FFF;
        const u32 sh = (e.y >> 13) & 0xF;
        const u32 bias = e.y >> 17;
        const bool flag = on && (st >= (fr << 19));
        const u64 m = __ballot(flag);
        const u32 grp = (u32)(m >> grpShift) & 0xF;
        const u32 before = __popc(grp & lowMask);
        const u32 a = q - 2 * (before + 1);
        if (flag) *reinterpret_cast<u16*>(stg + (a & 255)) = (u16)(((st >> 8) & 0xFF) | ((st & 0xFF) << 8));
        st = flag ? (st >> 16) : st;
        q -= 2 * __popc(grp);
        const u32 qd = __umulhi(st, e.x) >> sh;
        const u32 stn = st + bias + qd * (ANS_SCALE - fr);
        st = on ? stn : st;
    };
    // dword d (0..15) of an interval held as r[d & 3] of lane d >> 2
    auto entry_of = [&](const u32 r[4], int d) -> uint2 {
        u32 w;
        switch (d >> 2) {
        case 0: w = quad_bcast<0>(r[d & 3]); break;
        case 1: w = quad_bcast<1>(r[d & 3]); break;
        case 2: w = quad_bcast<2>(r[d & 3]); break;
        default: w = quad_bcast<3>(r[d & 3]); break;
        }
        return mytab[(w >> jsh) & 0xFF];
    };
    u32 cur[4], nxt[4];
    load_interval(0, cur);
    const u32 nIntervals = (maxSteps + 15) >> 4;
    uint2 e = entry_of(cur, 15);                      // step 16k+u uses dword 15-u of interval k
    for (u32 k = 0; k < nIntervals; k++) {
        load_interval(k + 1, nxt);
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const uint2 en = (u < 15) ? entry_of(cur, 14 - u) : entry_of(nxt, 15);
            step(16 * k + (u32)u, e);
            e = en;
        }
        KNZ_WAVE_ORDER();       // (the ring stores of every lane before the reads below)
        if (act && q + 128 <= Hi) { flush_line(); Hi -= 128; }
#pragma unroll
        for (int c = 0; c < 4; c++) cur[c] = nxt[c];
    }
    // remaining staged bytes [q, Hi): 16-bit units, interleaved over the chunk's 4 lanes
    KNZ_WAVE_ORDER();
    if (act) {
        for (u32 a = q + 2 * (u32)j; a < Hi; a += 8)
            *reinterpret_cast<u16*>(pay + a) = *reinterpret_cast<const u16*>(stg + (a & 255));
    }

    // varint(size) + 4 states -> mid ; payload piece
    const u32 s0 = (u32)__shfl((int)st, (g << 2) + 0, 64);
    const u32 s1 = (u32)__shfl((int)st, (g << 2) + 1, 64);
    const u32 s2 = (u32)__shfl((int)st, (g << 2) + 2, 64);
    const u32 s3 = (u32)__shfl((int)st, (g << 2) + 3, 64);
    if (act && j == 0) {
        ChunkDesc* cd = desc + slot;
        const u32 sz = top - q;
        u8 mid[24];
        u32 ml = 0;
        u32 v = sz;
        while (v >= 128) { mid[ml++] = (u8)(0x80 | (v & 0x7F)); v >>= 7; }
        mid[ml++] = (u8)v;
        const u32 sts[4] = { s0, s1, s2, s3 };
        for (int k = 0; k < 4; k++) {
            mid[ml++] = (u8)(sts[k] >> 24); mid[ml++] = (u8)(sts[k] >> 16);
            mid[ml++] = (u8)(sts[k] >> 8);  mid[ml++] = (u8)sts[k];
        }
        u32 mw[6] = { 0, 0, 0, 0, 0, 0 };
        for (u32 i = 0; i < ml; i++) mw[i >> 2] |= (u32)mid[i] << (8 * (i & 3));
        for (int i = 0; i < 6; i++) cd->mid[i] = mw[i];
        cd->midLen = ml;
        if (sz) { cd->nPieces = 1; cd->pieceBits[0] = 8 * sz; cd->piecePtr[0] = pay + q; }
    }
}


void launch_ans0_encode(hipStream_t s, BlockView view, int nBlocks, int maxChunks, ChunkDesc* desc, uint2* encTab, u8* tmp)
{
    const int nSlots = nBlocks * maxChunks;
    { KScope ks_("k_ans0_stats"); hipLaunchKernelGGL(k_ans0_stats, dim3(nSlots), dim3(64), 0, s, view, maxChunks, desc, encTab, tmp); }
    { KScope ks_("k_ans0_encode"); hipLaunchKernelGGL(k_ans0_encode, dim3((nSlots + 15) / 16), dim3(64), 0, s, view, maxChunks, nSlots, desc, encTab, tmp); }
}

}  // namespace knz
