// FPAQ on gfx950: kanzi's order-0 binary arithmetic coder with 4 x 256 adaptive bit probabilities selected by the two top
// bits of the previous byte.
//
// Reference being replaced (bit-identical streams): entropy/FPAQEncoder.cpp:58-110, FPAQEncoder.hpp:72-94 (encodeBit / flush),
// entropy/FPAQDecoder.cpp:62-120, FPAQDecoder.hpp:74-117. A block is coded in sub-chunks of 4 MiB (var-int byte count, payload,
// 56-bit tail); the interval [low, high] and the probabilities carry through the whole block.
//
// What the format allows to run in parallel, and what it does not:
//   * The probability a bit is coded with depends only on the earlier bits seen in the same context (previous byte >> 6, bit
//     prefix of the current byte), and the encoder knows every bit up front. Phase 1 (k_fpaq_probs) therefore runs the
//     4 x 8 (context class, tree level) families of probability chains side by side, one wave each over the whole block: levels
//     0-2 (1, 2, 4 nodes: every byte hits them) keep their probabilities in scalar registers and walk the matching lanes of a
//     64-byte tile in order; levels 3-7 rank the lanes of a tile that hit the same node (ballot matching) and apply them in
//     rank order, all nodes at once, with the probabilities in LDS. Result: the 16-bit probability of every coded bit, 16
//     bytes per input byte, in coding order.
//   * The interval recurrence is one dependent chain per block (the width of the interval decides when 32 bits leave, and
//     that decides the next width). Phase 2 (k_fpaq_code) is that chain and nothing else: wave-uniform arithmetic (scalar
//     registers) on bits and probabilities that the wave's lanes have loaded a tile ahead, 64 bytes at a time; flushed words
//     go to an LDS ring and leave as coalesced stores.
//     Both phases are ONE launch (round 6): 32 family waves and the coding wave of a block run side by side, the coding wave starts at
//     tile 0 and follows the families' progress counters (FPAQ_PUBLISH tiles at a time, release / acquire at device scope) instead
//     of waiting for the last probability of the block -- phase 1 (2.1 s of a 9.7 s encode on 30 blocks of 32 MiB) is hidden under phase 2.
//   * The decoder has no such split (the next context is the decoded bit): one wave per block, wave-uniform chain; the
//     probabilities live in LDS as (even, odd child) pairs so that both candidates for the next step arrive with one read
//     issued before the bit is known, the payload is pre-shifted to 32-bit units held one per lane (64 units per load).
#include "common.hpp"
#include "stages.hpp"

namespace knz {

#ifdef KNZ_EMU_FPAQ_CHUNK          // CPU emulation tests only: a small sub-chunk, to cross sub-chunk borders with small inputs
constexpr u32 FPAQ_CHUNK = KNZ_EMU_FPAQ_CHUNK;
#else
constexpr u32 FPAQ_CHUNK = 4u << 20;
#endif
constexpr u64 FPAQ_TOP = 0x00FFFFFFFFFFFFFFull;
constexpr u64 FPAQ_MASK24 = 0x0000000000FFFFFFull;
constexpr u64 FPAQ_MASK32 = 0x00000000FFFFFFFFull;
constexpr u64 FPAQ_MASK56 = 0x00FFFFFFFFFFFFFFull;

__device__ __forceinline__ u32 fpaq_rl(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ u32 fpaq_uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }

// FPAQEncoder.hpp:75-84: towards 0 after a 0 bit, towards 65536 after a 1 bit, by 1/64 of the distance
__device__ __forceinline__ u32 fpaq_update(u32 p, u32 bit)
{
    return bit ? ((p - (u32)(((int)p - 65536 + 64) >> 6)) & 0xFFFFu) : (p - (p >> 6));
}

// ---- progress of the 32 families of a block, in tiles of 64 bytes (ctrl[FPAQ_CTRL_PROG + 32 b + family]; 0xFFFFFFFF: done) ----------
constexpr u32 FPAQ_PUBLISH = 1024;       // tiles between two publications (64 KiB of the block: a release fence writes the L2 back)
constexpr u32 FPAQ_CTRL_PROG = 64;       // ctrl[0] = ticket counter of the launch
#ifdef KNZ_EMU
__device__ __forceinline__ u32 fpaq_ld_dev(const u32* p) { return *p; }
__device__ __forceinline__ void fpaq_st_dev(u32* p, u32 v) { *p = v; }
#define FPAQ_SPIN() do { fprintf(stderr, "k_fpaq_encode: the families of this block have not run\n"); abort(); } while (0)
#else
__device__ __forceinline__ u32 fpaq_ld_dev(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fpaq_st_dev(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define FPAQ_SPIN() __builtin_amdgcn_s_sleep(8)
#endif

// what the family has stored so far becomes visible to the other compute units, then the counter says so
__device__ __forceinline__ void fpaq_publish(u32* prog, u32 tilesDone)
{
    __threadfence();
    if (lane_id() == 0) fpaq_st_dev(prog, tilesDone);
}

// the coding wave: wait until all 32 families of the block have published `need` tiles; returns what they have published
__device__ __forceinline__ u32 fpaq_await(const u32* progBlock, u32 need)
{
    const int lane = lane_id();
    u32 have;
    for (;;) {
        u32 v = (lane < 32) ? fpaq_ld_dev(progBlock + lane) : 0xFFFFFFFFu;
        for (int o = 32; o > 0; o >>= 1) { const u32 t = (u32)__shfl_xor((int)v, o, 64); v = t < v ? t : v; }
        have = fpaq_uni(v);
        if (have >= need) break;
        FPAQ_SPIN();
    }
    __threadfence();                      // (acquire: nothing read before this point stands in for the probabilities read behind it)
    return have;
}

// ------------------------------------------------------------------------------------------------
// encoder, phase 1: the probability every bit will be coded with
// ------------------------------------------------------------------------------------------------
// family = context class * 8 + tree level. probs[(b * pStride) + 8 i + level] for byte i of block b.
__device__ __forceinline__ void fpaq_probs_family(BlockView view, int b, u32 family, u16* __restrict__ probs, u64 pStride, u32* __restrict__ prog, u16* pr)
{
    const u32 c2 = family >> 3, level = family & 7;
    const int lane = lane_id();
    const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const u32 count = view.len[b];
    const u8* blk = view.ptr[b];
    u16* out = probs + (u64)b * pStride;
    for (int q = lane; q < 128; q += 64) pr[q] = 32768;
    u32 p0 = 32768, p1 = 32768, p2 = 32768, p3 = 32768;
    // the bytes of the next tile (and the byte before each) are loaded while this tile is worked on
    u32 nByte = 0, nPrev = 0;
    if ((u32)lane < count) { nByte = blk[lane]; nPrev = lane ? blk[lane - 1] : 0u; }
    for (u32 i0 = 0; i0 < count; i0 += 64) {
        const u32 i = i0 + (u32)lane;
        const bool valid = i < count;
        const u32 byte = nByte;
        const u32 prev2 = (valid && (i % FPAQ_CHUNK) != 0) ? (nPrev >> 6) : 0u;     // every sub-chunk starts in class 0
        { const u32 j = i + 64; nByte = 0; nPrev = 0; if (j < count) { nByte = blk[j]; nPrev = blk[j - 1]; } }
        const bool match = valid && prev2 == c2;
        const u32 node = ((byte | 256u) >> (8 - level)) - (1u << level);
        const u32 bit = (byte >> (7 - level)) & 1u;
        if (i0 && ((i0 >> 6) % FPAQ_PUBLISH) == 0) fpaq_publish(prog, i0 >> 6);        // the tiles in front of this one are stored
        unsigned long long m = __ballot(match);
        if (m == 0) continue;
        u32 myP = 0;
        if (level <= 2) {
            const unsigned long long mb = __ballot(match && bit), n0 = __ballot(match && (node & 1u)), n1 = __ballot(match && (node & 2u));
            while (m) {
                const int l = __ffsll((long long)m) - 1;
                m &= m - 1;
                const u32 nd = (u32)((n0 >> l) & 1) | ((u32)((n1 >> l) & 1) << 1);
                const u32 bt = (u32)((mb >> l) & 1);
                u32 p = (nd == 0) ? p0 : (nd == 1) ? p1 : (nd == 2) ? p2 : p3;
                if (lane == l) myP = p;
                p = fpaq_update(p, bt);
                if (nd == 0) p0 = p; else if (nd == 1) p1 = p; else if (nd == 2) p2 = p; else p3 = p;
            }
        } else {
            unsigned long long peers = m;
            for (u32 k = 0; k < level; k++) {
                const bool one = (node >> k) & 1u;
                const unsigned long long bal = __ballot(match && one);
                peers &= one ? bal : ~bal;
            }
            const u32 rank = (u32)__popcll(peers & ltMask);
            unsigned long long pending = m;
            for (u32 r = 0; pending; r++) {
                const bool act = match && rank == r;
                if (act) {
                    const u32 p = pr[node];
                    myP = p;
                    pr[node] = (u16)fpaq_update(p, bit);
                }
                pending &= ~__ballot(act);
            }
        }
        if (match) out[(u64)i * 8 + level] = (u16)myP;
    }
    fpaq_publish(prog, 0xFFFFFFFFu);
}

// ------------------------------------------------------------------------------------------------
// encoder, phase 2: the interval recurrence, one wave per block
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fpaq_desc_finish(ChunkDesc& cd, u32 index, const u8* buf, u64 low)
{
    cd.hdrBits = 0; cd.aux = 0;
    u8 mid[8];
    u32 ml = 0;
    u32 v = index;
    while (v >= 128) { mid[ml++] = (u8)(0x80 | (v & 0x7F)); v >>= 7; }
    mid[ml++] = (u8)v;
    u32 mw[6] = { 0, 0, 0, 0, 0, 0 };
    for (u32 i = 0; i < ml; i++) mw[i >> 2] |= (u32)mid[i] << (8 * (i & 3));
    for (int i = 0; i < 6; i++) cd.mid[i] = mw[i];
    cd.midLen = ml;
    cd.nPieces = index ? 1 : 0;
    cd.pieceBits[0] = 8 * index;
    cd.piecePtr[0] = buf;
    // 56 bits of low | 0xFFFFFF after every sub-chunk; the last one is written by dispose() (FPAQEncoder.cpp:92-110)
    const u64 tail = (low | FPAQ_MASK24) & FPAQ_MASK56;
    u32 tw[2] = { 0, 0 };
    for (int k = 0; k < 7; k++) { const u32 byte = (u32)((tail >> (48 - 8 * k)) & 0xFF); tw[k >> 2] |= byte << (8 * (k & 3)); }
    cd.trailer[0] = tw[0]; cd.trailer[1] = tw[1];
    cd.trailerLen = 7;
}


__device__ __forceinline__ void fpaq_code_block(BlockView view, int b, const u32* __restrict__ origLen, u32 copyThreshold, int maxChunks,
                                                ChunkDesc* __restrict__ desc, u8* __restrict__ tmp, u64 tmpStride, const u16* __restrict__ probs, u64 pStride,
                                                const u32* __restrict__ progBlock, u32* ring)
{
    const int lane = lane_id();
    const u32 count = view.len[b];
    const u8* blk = view.ptr[b];
    ChunkDesc* cds = desc + (size_t)b * maxChunks;
    if (origLen[b] <= copyThreshold) {
        // copy block: entropy type forced to NONE (io/CompressedOutputStream.cpp:691-695)
        if (lane == 0) {
            ChunkDesc& cd = cds[0];
            cd.hdrBits = 0; cd.midLen = 0; cd.trailerLen = 0; cd.aux = 0;
            cd.nPieces = 1; cd.pieceBits[0] = 8 * count; cd.piecePtr[0] = blk;
        }
        return;
    }
    const uint4* pv = reinterpret_cast<const uint4*>(probs + (u64)b * pStride);   // 8 probabilities = 16 bytes per input byte
    u64 low = 0, high = FPAQ_TOP;
    u32 startChunk = 0;
    int ci = 0;
    u32 have = 0;                                                 // tiles of the block whose probabilities are known to be there
    while (startChunk < count) {
        const u32 chunkSize = (FPAQ_CHUNK < count - startChunk) ? FPAQ_CHUNK : count - startChunk;
        const u32 endChunk = startChunk + chunkSize;
        u8* buf = tmp + ((size_t)b * maxChunks + ci) * tmpStride;
        u32 index = 0;
        // the tile after the one being coded is already loaded
        u32 nByte = 0;
        uint4 nPw; nPw.x = nPw.y = nPw.z = nPw.w = 0;
        if ((startChunk >> 6) + 1 > have) have = fpaq_await(progBlock, (startChunk >> 6) + 1);
        { const u32 i = startChunk + (u32)lane; if (i < endChunk) { nByte = blk[i]; nPw = pv[i]; } }
        for (u32 i0 = startChunk; i0 < endChunk; i0 += 64) {
            const u32 byte = nByte;
            const uint4 pw = nPw;
            if (i0 + 64 < endChunk && (i0 >> 6) + 2 > have) have = fpaq_await(progBlock, (i0 >> 6) + 2);
            { const u32 i = i0 + 64 + (u32)lane; nByte = 0; if (i < endChunk) { nByte = blk[i]; nPw = pv[i]; } }
            const u32 nb = (endChunk - i0 < 64) ? endChunk - i0 : 64;
            u32 cnt = 0;
            for (u32 l = 0; l < nb; l++) {
                const u32 bv = fpaq_rl(byte, l);
                const u32 q[4] = { fpaq_rl(pw.x, l), fpaq_rl(pw.y, l), fpaq_rl(pw.z, l), fpaq_rl(pw.w, l) };
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const u64 p = (k & 1) ? (q[k >> 1] >> 16) : (q[k >> 1] & 0xFFFFu);
                    const u64 mid = low + ((((high - low) >> 8) * p) >> 8);
                    const bool one = (bv >> (7 - k)) & 1u;
                    high = one ? mid : high;                       // selects, not branches: the bit is as good as random
                    low = one ? low : mid + 1;
                    const u64 x = low ^ high;
                    if ((((u32)(x >> 32)) | ((u32)x >> 24)) == 0) {  // top 32 of the 56 bits agree: they leave
                        if (lane == 0) ring[cnt] = (u32)(high >> 24);
                        cnt++;
                        low <<= 32;
                        high = (high << 32) | FPAQ_MASK32;
                    }
                }
            }
            __syncthreads();
            for (u32 qd = (u32)lane; qd < cnt; qd += 64) reinterpret_cast<u32*>(buf + index)[qd] = bswap32(ring[qd]);
            __syncthreads();
            index += 4 * cnt;
        }
        if (lane == 0) fpaq_desc_finish(cds[ci], index, buf, low);
        startChunk = endChunk;
        ci++;
    }
}

// One launch for both phases: 33 waves per block, roles by TICKET (drawn when the workgroup starts), 32 families first, then the
// coding wave -- so the coding wave of a block only ever waits for workgroups that started before it, whatever order the dispatcher
// starts workgroups in.
__global__ __launch_bounds__(64) void k_fpaq_encode(BlockView view, const u32* __restrict__ origLen, u32 copyThreshold, int nBlocks, int maxChunks,
                                                    ChunkDesc* __restrict__ desc, u8* __restrict__ tmp, u64 tmpStride, u16* __restrict__ probs, u64 pStride,
                                                    u32* __restrict__ ctrl)
{
    __shared__ u32 ring[528];
    __shared__ u16 pr[128];
    __shared__ u32 sTicket;
    if (threadIdx.x == 0) sTicket = atomicAdd(&ctrl[0], 1u);
    __syncthreads();
    const u32 t = sTicket;
    const int b = (int)(t / 33u);
    const u32 role = t % 33u;
    if (b >= nBlocks) return;
    u32* progBlock = ctrl + FPAQ_CTRL_PROG + 32u * (u32)b;
    const bool copyBlock = origLen[b] <= copyThreshold;
    if (role < 32) { if (!copyBlock) fpaq_probs_family(view, b, role, probs, pStride, progBlock + role, pr); }
    else fpaq_code_block(view, b, origLen, copyThreshold, maxChunks, desc, tmp, tmpStride, probs, pStride, progBlock, ring);
}

// ------------------------------------------------------------------------------------------------
// decoder: one wave per block, wave-uniform chain
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_fpaq_decode(BitSrc src, DecBlock* __restrict__ blocks, u8* const* __restrict__ outPtr)
{
    const int b = blockIdx.x;
    const int lane = lane_id();
    // probabilities as 16-bit values, read as pairs: word c2 * 128 + node holds the children 2 node (low half) and 2 node + 1
    __shared__ u32 pr32[512];
    // the 16-bit stores below go to the words the 32-bit loads read: say so, or type-based alias analysis lets the compiler keep
    // a loaded pair across a store (it hoisted the read of the root pair out of the sub-chunk loop)
    typedef u16 __attribute__((may_alias)) u16_alias;
    u16_alias* pr16 = reinterpret_cast<u16_alias*>(pr32);
    for (int i = lane; i < 512; i += 64) pr32[i] = 0x80008000u;
    __syncthreads();
    DecBlock& db = blocks[b];
    if (db.error) return;
    BitSrc s = src;
    s.limitBits = db.payloadBit + ((db.bits + 7) & ~7ull);
    u64 pos = db.entropyBit;
    const u32 count = db.preLen;
    u8* block = outPtr[b];
    if (db.copyBlock) {
        bool bad = pos + 8ull * count > s.limitBits;
        if (!bad) for (u32 i = (u32)lane; i < count; i += 64) block[i] = (u8)peek_bits(s, pos + 8ull * i, 8);
        if (lane == 0) { if (bad) db.error = KNZ_ERR_PROCESS_BLOCK; db.usedBits = bad ? (s.limitBits - db.entropyBit) : 8ull * count; }
        return;
    }
    const u64 lastWord = ((src.nBytes + 3) >> 2) - 1;
    u64 low = 0, high = FPAQ_TOP, current = 0;
    u32 startChunk = 0;
    bool fail = false;
    while (startChunk < count && !fail) {
        int err = 0;
        const u32 szBytes = take_varint(s, pos, err);
        if (err) { fail = true; break; }
        if (szBytes >= 2 * count) { fail = true; break; }          // FPAQDecoder.cpp:75-76
        current = ((u64)take_bits(s, pos, 24, err) << 32) | take_bits(s, pos, 32, err);
        const u64 payBit = pos;
        pos += 8ull * szBytes;
        if (err || pos > s.limitBits) { fail = true; break; }
        // payload as 32-bit units in stream order, unit u = bits [payBit + 32 u, + 32): window = 64 units, one per lane
        const u64 wbase = payBit >> 5;
        const u32 sh = (u32)(payBit & 31);
        auto loadWin = [&](u32 unit0) -> u32 {
            const u64 w = wbase + unit0 + (u32)lane;
            const u32 a = bswap32(src.words[w < lastWord ? w : lastWord]);
            const u32 c = bswap32(src.words[w + 1 < lastWord ? w + 1 : lastWord]);
            return sh ? ((a << sh) | (c >> (32 - sh))) : a;
        };
        u32 winBase = 0;
        u32 winCur = loadWin(0), winNext = loadWin(64);
        u32 index = 0;
        const u32 chunkSize = (FPAQ_CHUNK < count - startChunk) ? FPAQ_CHUNK : count - startChunk;
        const u32 endChunk = startChunk + chunkSize;
        u32 c2 = 0;
        u32 pRoot = fpaq_uni(pr32[0]) >> 16;                      // node 1 of class 0
        for (u32 i0 = startChunk; i0 < endChunk && !fail; i0 += 64) {
            const u32 nb = (endChunk - i0 < 64) ? endChunk - i0 : 64;
            u32 myByte = 0;
            for (u32 l = 0; l < nb; l++) {
                u32 ctx = 1;
                u32 p = pRoot;
                u32 nextRoot = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    // both children of the node, before the bit is known
                    u32 pair = 0;
                    if (k < 7) pair = pr32[c2 * 128 + ctx];
                    const u64 split = ((((high - low) >> 8) * (u64)p) >> 8) + low;
                    const bool one = split >= current;
                    const u32 np = fpaq_update(p, one ? 1u : 0u);
                    pr16[c2 * 256 + ctx] = (u16)np;                 // every lane stores the same value: no exec-mask detour on the chain
                    high = one ? split : high;
                    low = one ? low : split + 1;
                    ctx = 2 * ctx + (one ? 1u : 0u);
                    if (k < 7) { pair = fpaq_uni(pair); p = one ? (pair >> 16) : (pair & 0xFFFFu); }
                    if (k == 1) nextRoot = pr32[(ctx & 3) * 128];          // the class of the next byte is known after two bits
                    const u64 x = low ^ high;
                    if ((((u32)(x >> 32)) | ((u32)x >> 24)) == 0) {
                        low = (low << 32) & FPAQ_MASK56;
                        high = ((high << 32) | FPAQ_MASK32) & FPAQ_MASK56;
                        if (index + 4 > szBytes) {
                            current = (current << 32) & FPAQ_MASK56;
                            index = szBytes + 1;
                        } else {
                            const u32 u = index >> 2;
                            if (u - winBase >= 64) { winCur = winNext; winBase += 64; winNext = loadWin(winBase + 64); }
                            const u64 val = fpaq_rl(winCur, (u - winBase) & 63);
                            current = ((current << 32) | val) & FPAQ_MASK56;
                            index += 4;
                        }
                    }
                }
                if ((u32)lane == l) myByte = ctx & 0xFF;
                if (index > szBytes) { fail = true; break; }
                // next byte: class = top two bits of this one; its root probability was read after bit 1, but the root of the
                // same class may have been this byte's own root (updated at bit 0, before that read) -- LDS keeps program order
                c2 = (ctx & 0xFF) >> 6;
                pRoot = fpaq_uni(nextRoot) >> 16;
            }
            if ((u32)lane < nb && !fail) block[i0 + (u32)lane] = (u8)myByte;
            else if (fail) { /* partial tile: the block is rejected anyway */ }
        }
        if (index > szBytes) fail = true;
        startChunk = endChunk;
    }
    if (lane == 0) {
        if (fail) db.error = KNZ_ERR_PROCESS_BLOCK;
        db.usedBits = pos - db.entropyBit;
    }
}

static size_t fpaq_probs_only_bytes(int nBlocks, u64 S) { return ((size_t)nBlocks * (size_t)((S + 63) & ~63ull) * 16 + 255) & ~(size_t)255; }
// probabilities + the control words of the launch (ticket, progress counters)
size_t fpaq_probs_bytes(int nBlocks, u64 S) { return fpaq_probs_only_bytes(nBlocks, S) + 4 * ((size_t)FPAQ_CTRL_PROG + 32 * (size_t)nBlocks) + 256; }

void launch_fpaq_encode(hipStream_t s, BlockView view, const u32* origLen, u32 copyThreshold, int nBlocks, int maxChunks, ChunkDesc* desc, u8* tmp, u64 tmpStride,
                        u16* probs, u64 S)
{
    const u64 pStride = ((S + 63) & ~63ull) * 8;
    hipMemsetAsync(desc, 0, sizeof(ChunkDesc) * (size_t)nBlocks * maxChunks, s);
    u32* ctrl = reinterpret_cast<u32*>(reinterpret_cast<u8*>(probs) + fpaq_probs_only_bytes(nBlocks, S));
    hipMemsetAsync(ctrl, 0, 4 * ((size_t)FPAQ_CTRL_PROG + 32 * (size_t)nBlocks), s);
    { KScope ks_("k_fpaq_encode"); hipLaunchKernelGGL(k_fpaq_encode, dim3(33u * (unsigned)nBlocks), dim3(64), 0, s, view, origLen, copyThreshold, nBlocks, maxChunks, desc, tmp, tmpStride,
                                                      probs, pStride, ctrl); }
}

void launch_fpaq_decode(hipStream_t s, BitSrc src, DecBlock* blocks, int nBlocks, u8* const* outPtr)
{
    { KScope ks_("k_fpaq_decode"); hipLaunchKernelGGL(k_fpaq_decode, dim3(nBlocks), dim3(64), 0, s, src, blocks, outPtr); }
}

}  // namespace knz
