// The stage whose reference definition is a single dependency chain per block. It runs as ONE LANE
// PER BLOCK (blocks of a batch in parallel), i.e. they are latency-bound, not bandwidth-bound:
//
//   FPAQ  entropy/FPAQEncoder.cpp:58-110, FPAQEncoder.hpp:72-94 ; FPAQDecoder.cpp:62-120, .hpp:74-117
//         56-bit binary arithmetic coder whose interval and 1024 adaptive probabilities carry across
//         every bit of the block (SURVEY.md section 7 "hard parts": no bit-exact parallel form for the decoder).
//
// They exist so that every BASELINE config runs end-to-end on the device with a bit-exact stream;
// DESIGN.md reports them as latency-bound.
#include "common.hpp"
#include "stages.hpp"

namespace knz {

// ================================================================================================
// FPAQ
// ================================================================================================
constexpr u32 FPAQ_CHUNK = 4u << 20;
constexpr u64 FPAQ_TOP = 0x00FFFFFFFFFFFFFFull;
constexpr u64 MASK_0_24 = 0x0000000000FFFFFFull;
constexpr u64 MASK_0_32 = 0x00000000FFFFFFFFull;
constexpr u64 MASK_0_56 = 0x00FFFFFFFFFFFFFFull;
constexpr int PSCALE = 65536;

__device__ __forceinline__ void fpaq_enc_bit(u64& low, u64& high, u8* buf, u32& index, int bit, u16& prob)
{
    if (bit == 0) {
        low = low + ((((high - low) >> 8) * (u64)prob) >> 8) + 1;
        prob = (u16)(prob - (u16)(prob >> 6));
    } else {
        high = low + ((((high - low) >> 8) * (u64)prob) >> 8);
        prob = (u16)(prob - (u16)(((int)prob - PSCALE + 64) >> 6));
    }
    if (((low ^ high) >> 24) == 0) {
        const u32 v = (u32)(high >> 24);
        *reinterpret_cast<u32*>(buf + index) = bswap32(v);     // index is a multiple of 4, buf 4-byte aligned
        index += 4;
        low <<= 32;
        high = (high << 32) | MASK_0_32;
    }
}

// one workgroup (lane 0 active) per block
__global__ __launch_bounds__(64) void k_fpaq_encode(BlockView view, const u32* __restrict__ origLen, u32 copyThreshold, int maxChunks,
                                                    ChunkDesc* __restrict__ desc, u8* __restrict__ tmp, u64 tmpStride)
{
    const int b = blockIdx.x;
    __shared__ u16 probs[4][256];
    for (int i = threadIdx.x; i < 1024; i += 64) (&probs[0][0])[i] = PSCALE >> 1;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const u32 count = view.len[b];
    const u8* blk = view.ptr[b];
    ChunkDesc* cds = desc + (size_t)b * maxChunks;
    if (origLen[b] <= copyThreshold) {
        // copy block: entropy type forced to NONE (io/CompressedOutputStream.cpp:691-695)
        ChunkDesc& cd = cds[0];
        cd.hdrBits = 0; cd.midLen = 0; cd.trailerLen = 0; cd.aux = 0;
        cd.nPieces = 1; cd.pieceBits[0] = 8 * count; cd.piecePtr[0] = blk;
        return;
    }
    u64 low = 0, high = FPAQ_TOP;
    u32 startChunk = 0;
    int ci = 0;
    while (startChunk < count) {
        const u32 chunkSize = (FPAQ_CHUNK < count - startChunk) ? FPAQ_CHUNK : count - startChunk;
        u8* buf = tmp + ((size_t)b * maxChunks + ci) * tmpStride;
        u32 index = 0;
        const u32 endChunk = startChunk + chunkSize;
        u16* p = probs[0];
        for (u32 i = startChunk; i < endChunk; i++) {
            const int val = blk[i];
            const int bits = val + 256;
            fpaq_enc_bit(low, high, buf, index, val & 0x80, p[1]);
            fpaq_enc_bit(low, high, buf, index, val & 0x40, p[bits >> 7]);
            fpaq_enc_bit(low, high, buf, index, val & 0x20, p[bits >> 6]);
            fpaq_enc_bit(low, high, buf, index, val & 0x10, p[bits >> 5]);
            fpaq_enc_bit(low, high, buf, index, val & 0x08, p[bits >> 4]);
            fpaq_enc_bit(low, high, buf, index, val & 0x04, p[bits >> 3]);
            fpaq_enc_bit(low, high, buf, index, val & 0x02, p[bits >> 2]);
            fpaq_enc_bit(low, high, buf, index, val & 0x01, p[bits >> 1]);
            p = probs[val >> 6];
        }
        ChunkDesc& cd = cds[ci];
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
        const u64 tail = (low | MASK_0_24) & MASK_0_56;
        u32 tw[2] = { 0, 0 };
        for (int k = 0; k < 7; k++) { const u32 byte = (u32)((tail >> (48 - 8 * k)) & 0xFF); tw[k >> 2] |= byte << (8 * (k & 3)); }
        cd.trailer[0] = tw[0]; cd.trailer[1] = tw[1];
        cd.trailerLen = 7;
        startChunk += chunkSize;
        ci++;
    }
}

__global__ __launch_bounds__(64) void k_fpaq_decode(BitSrc src, DecBlock* __restrict__ blocks, u8* const* __restrict__ outPtr)
{
    const int b = blockIdx.x;
    __shared__ u16 probs[4][256];
    for (int i = threadIdx.x; i < 1024; i += 64) (&probs[0][0])[i] = PSCALE >> 1;
    __syncthreads();
    if (threadIdx.x != 0) return;
    DecBlock& db = blocks[b];
    if (db.error) return;
    BitSrc s = src;
    s.limitBits = db.payloadBit + ((db.bits + 7) & ~7ull);
    u64 pos = db.entropyBit;
    const u32 count = db.preLen;
    u8* block = outPtr[b];
    int err = 0;
    if (db.copyBlock) {
        for (u32 i = 0; i < count; i++) block[i] = (u8)take_bits(s, pos, 8, err);
        if (err) db.error = KNZ_ERR_PROCESS_BLOCK;
        db.usedBits = pos - db.entropyBit;
        return;
    }
    u64 low = 0, high = FPAQ_TOP, current = 0;
    u32 startChunk = 0;
    bool fail = false;
    while (startChunk < count && !fail) {
        const u32 szBytes = take_varint(s, pos, err);
        if (err) { fail = true; break; }
        if (szBytes >= 2 * count) { fail = true; break; }          // FPAQDecoder.cpp:75-76
        current = ((u64)take_bits(s, pos, 24, err) << 32) | take_bits(s, pos, 32, err);
        const u64 payBit = pos;
        pos += 8ull * szBytes;
        if (err || pos > s.limitBits) { fail = true; break; }
        u32 index = 0;
        const u32 chunkSize = (FPAQ_CHUNK < count - startChunk) ? FPAQ_CHUNK : count - startChunk;
        const u32 endChunk = startChunk + chunkSize;
        u16* p = probs[0];
        for (u32 i = startChunk; i < endChunk; i++) {
            int ctx = 1;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const u64 split = ((((high - low) >> 8) * (u64)p[ctx]) >> 8) + low;
                if (split >= current) {
                    high = split;
                    p[ctx] = (u16)(p[ctx] - (u16)(((int)p[ctx] - PSCALE + 64) >> 6));
                    ctx += ctx + 1;
                } else {
                    low = split + 1;
                    p[ctx] = (u16)(p[ctx] - (u16)(p[ctx] >> 6));
                    ctx += ctx;
                }
                if (((low ^ high) >> 24) == 0) {
                    low = (low << 32) & MASK_0_56;
                    high = ((high << 32) | MASK_0_32) & MASK_0_56;
                    if (index + 4 > szBytes) {
                        current = (current << 32) & MASK_0_56;
                        index = szBytes + 1;
                    } else {
                        const u64 val = peek_bits(src, payBit + 8ull * index, 32);
                        current = ((current << 32) | val) & MASK_0_56;
                        index += 4;
                    }
                }
            }
            block[i] = (u8)ctx;
            if (index > szBytes) { fail = true; break; }
            p = probs[(ctx & 0xFF) >> 6];
        }
        if (index > szBytes) fail = true;
        startChunk = endChunk;
    }
    if (fail) db.error = KNZ_ERR_PROCESS_BLOCK;
    db.usedBits = pos - db.entropyBit;
}

void launch_fpaq_encode(hipStream_t s, BlockView view, const u32* origLen, u32 copyThreshold, int nBlocks, int maxChunks, ChunkDesc* desc, u8* tmp, u64 tmpStride)
{
    hipMemsetAsync(desc, 0, sizeof(ChunkDesc) * (size_t)nBlocks * maxChunks, s);
    { KScope ks_("k_fpaq_encode"); hipLaunchKernelGGL(k_fpaq_encode, dim3(nBlocks), dim3(64), 0, s, view, origLen, copyThreshold, maxChunks, desc, tmp, tmpStride); }
}

void launch_fpaq_decode(hipStream_t s, BitSrc src, DecBlock* blocks, int nBlocks, u8* const* outPtr)
{
    { KScope ks_("k_fpaq_decode"); hipLaunchKernelGGL(k_fpaq_decode, dim3(nBlocks), dim3(64), 0, s, src, blocks, outPtr); }
}

}  // namespace knz
