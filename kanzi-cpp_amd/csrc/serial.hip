// Stages whose reference definition is a single dependency chain per block. They run as ONE LANE
// PER BLOCK (blocks of a batch in parallel), i.e. they are latency-bound, not bandwidth-bound:
//
//   FPAQ  entropy/FPAQEncoder.cpp:58-110, FPAQEncoder.hpp:72-94 ; FPAQDecoder.cpp:62-120, .hpp:74-117
//         56-bit binary arithmetic coder whose interval and 1024 adaptive probabilities carry across
//         every bit of the block (SURVEY.md section 7 "hard parts": no bit-exact parallel form for the decoder).
//   RLT   transform/RLT.cpp:39-221, :223-245, :247-369. Kept serial in round 1 (its 4-byte stride scan and
//         MAX_RUN splitting are reproduced literally); a scan-based version is future work.
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

// ================================================================================================
// RLT
// ================================================================================================
constexpr int RLT_ENC1 = 224;
constexpr int RLT_ENC2 = (255 - RLT_ENC1) << 8;
constexpr int RLT_THRESHOLD = 3;
constexpr int RLT_MAX_RUN = 0xFFFF + RLT_ENC2 + RLT_THRESHOLD - 1;
constexpr int RLT_MAX_RUN4 = RLT_MAX_RUN - 4;

__device__ int rlt_emit_run(u8* dst, int run, u8 escape, u8 val)
{
    dst[0] = val;
    dst[1] = 0;
    int dstIdx = (val == escape) ? 2 : 1;
    dst[dstIdx++] = escape;
    run -= RLT_THRESHOLD;
    if (run >= RLT_ENC1) {
        if (run < RLT_ENC2) { run -= RLT_ENC1; dst[dstIdx++] = (u8)(RLT_ENC1 + (run >> 8)); }
        else { run -= RLT_ENC2; dst[dstIdx++] = 0xFF; dst[dstIdx++] = (u8)(run >> 8); }
    }
    dst[dstIdx] = (u8)run;
    return dstIdx + 1;
}

// Global.cpp:354-397 (only the classes RLT cares about: DNA / BASE64 refuse the transform)
__device__ int rlt_refuses(int count, const u32* f)
{
    const char DNA[] = "acgntuACGNTU";
    const char NUM[] = "0123456789+-*/=,.:; ";
    const char B64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    int sum = 0;
    for (int i = 0; i < 12; i++) sum += (int)f[(u8)DNA[i]];
    if (sum > (count - count / 12)) return 1;                 // DNA
    sum = 0;
    for (int i = 0; i < 20; i++) sum += (int)f[(u8)NUM[i]];
    if (sum == count) return 0;                               // NUMERIC
    sum = (f[0x3D] == 1) ? 1 : 0;
    for (int i = 0; i < 64; i++) sum += (int)f[(u8)B64[i]];
    if (sum == count) return 1;                               // BASE64
    return 0;
}

__global__ __launch_bounds__(64) void k_rlt_forward(XfStage st)
{
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const int count = (int)st.len[b];
    st.ok[b] = 0; st.newLen[b] = 0;
    if (count == 0) { st.ok[b] = 1; return; }
    if (count < 16) return;
    const int maxEnc = (count <= 512) ? count + 32 : count;
    if ((int)st.cap[b] < maxEnc) return;
    const u8* src = st.src[b];
    u8* dst = st.dst[b];
    const int etype = st.entropyType;
    const bool findBestEscape = !(etype == KNZ_E_NONE || etype == KNZ_E_ANS0 || etype == KNZ_E_HUFFMAN || etype == 4);
    u8 escape = 0xFB;
    if (findBestEscape) {
        __shared__ u32 freqs[256];
        for (int i = 0; i < 256; i++) freqs[i] = 0;
        for (int i = 0; i < count; i++) freqs[src[i]]++;
        if (rlt_refuses(count, freqs)) return;
        int minIdx = 0;
        if (freqs[minIdx] > 0) {
            for (int i = 1; i < 256; i++) {
                if (freqs[i] < freqs[minIdx]) { minIdx = i; if (freqs[i] == 0) break; }
            }
        }
        escape = (u8)minIdx;
    }
    int srcIdx = 0, dstIdx = 0;
    const int srcEnd = count, srcEnd4 = srcEnd - 4, dstEnd = (int)st.cap[b];
    int res = 1, run = 0;
    u8 prev = src[srcIdx++];
    dst[dstIdx++] = escape;
    dst[dstIdx++] = prev;
    if (prev == escape) dst[dstIdx++] = 0;
    while (true) {
        if (prev == src[srcIdx]) {
            const u32 v = 0x01010101u * (u32)prev;
            const u32 w = (u32)src[srcIdx] | ((u32)src[srcIdx + 1] << 8) | ((u32)src[srcIdx + 2] << 16) | ((u32)src[srcIdx + 3] << 24);
            const u32 diff = w ^ v;
            if (diff == 0) {
                srcIdx += 4; run += 4;
                if ((run < RLT_MAX_RUN4) && (srcIdx < srcEnd4)) continue;
            } else {
                const int n = (__ffs((int)diff) - 1) >> 3;
                srcIdx += n;
                run += n;
            }
        }
        if (run > RLT_THRESHOLD) {
            if (dstIdx + 6 >= dstEnd) { res = 0; break; }
            dstIdx += rlt_emit_run(&dst[dstIdx], run, escape, prev);
        } else if (prev != escape) {
            if (dstIdx + run >= dstEnd) { res = 0; break; }
            if (run-- > 0) dst[dstIdx++] = prev;
            while (run-- > 0) dst[dstIdx++] = prev;
        } else {
            if (dstIdx + (2 * run) >= dstEnd) { res = 0; break; }
            while (run-- > 0) { dst[dstIdx++] = escape; dst[dstIdx++] = 0; }
        }
        prev = src[srcIdx];
        srcIdx++;
        run = 1;
        if (srcIdx >= srcEnd4) break;
    }
    if (res) {
        if (prev != escape) {
            if (dstIdx + run < dstEnd) while (run-- > 0) dst[dstIdx++] = prev;
        } else {
            if (dstIdx + (2 * run) < dstEnd) while (run-- > 0) { dst[dstIdx++] = escape; dst[dstIdx++] = 0; }
        }
        while ((srcIdx < srcEnd) && (dstIdx < dstEnd)) {
            if (src[srcIdx] == escape) {
                if (dstIdx + 2 >= dstEnd) { res = 0; break; }
                dst[dstIdx++] = escape;
                dst[dstIdx++] = 0;
                srcIdx++;
                continue;
            }
            dst[dstIdx++] = src[srcIdx++];
        }
        res &= (srcIdx == srcEnd) ? 1 : 0;
    }
    st.ok[b] = (res && (dstIdx < srcIdx)) ? 1 : 0;
    st.newLen[b] = (u32)dstIdx;
}

__global__ __launch_bounds__(64) void k_rlt_inverse(XfStage st)
{
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    const int count = (int)st.len[b];
    st.ok[b] = 0; st.newLen[b] = 0;
    if (count == 0) { st.ok[b] = 1; return; }
    const u8* src = st.src[b];
    u8* dst = st.dst[b];
    int srcIdx = 0, dstIdx = 0;
    const int srcEnd = count, dstEnd = (int)st.cap[b];
    int res = 1;
    const u8 escape = src[srcIdx++];
    if ((srcIdx < srcEnd) && (src[srcIdx] == escape)) {
        srcIdx++;
        if ((srcIdx < srcEnd) && (src[srcIdx] != 0)) return;
        if (dstIdx >= dstEnd) return;
        dst[dstIdx++] = escape;
        srcIdx++;
    }
    while (srcIdx < srcEnd) {
        // literal span up to the next escape
        int q = srcIdx;
        while (q < srcEnd && src[q] != escape) q++;
        const int literalLen = q - srcIdx;
        if (literalLen > 0) {
            if (literalLen > dstEnd - dstIdx) { res = 0; break; }
            for (int k = 0; k < literalLen; k++) dst[dstIdx + k] = src[srcIdx + k];
            srcIdx += literalLen;
            dstIdx += literalLen;
        }
        if (srcIdx >= srcEnd) break;
        srcIdx++;
        if (srcIdx >= srcEnd) { res = 0; break; }
        int run = src[srcIdx++];
        if (run == 0) {
            if (dstIdx >= dstEnd) { res = 0; break; }
            dst[dstIdx++] = escape;
            continue;
        }
        if (run == 0xFF) {
            if (srcIdx + 1 >= srcEnd) { res = 0; break; }
            run = ((int)src[srcIdx] << 8) | (int)src[srcIdx + 1];
            srcIdx += 2;
            run += RLT_ENC2;
        } else if (run >= RLT_ENC1) {
            if (srcIdx >= srcEnd) { res = 0; break; }
            run = ((run - RLT_ENC1) << 8) | (int)src[srcIdx];
            srcIdx++;
            run += RLT_ENC1;
        }
        run += (RLT_THRESHOLD - 1);
        if ((dstIdx + run > dstEnd) || (run > RLT_MAX_RUN)) { res = 0; break; }
        if (dstIdx == 0) { res = 0; break; }
        const u8 v = dst[dstIdx - 1];
        for (int k = 0; k < run; k++) dst[dstIdx + k] = v;
        dstIdx += run;
    }
    st.ok[b] = (res && (srcIdx == srcEnd)) ? 1 : 0;
    st.newLen[b] = (u32)dstIdx;
}

void launch_rlt_forward(hipStream_t s, const XfStage& st) { KScope ks_("k_rlt_forward"); hipLaunchKernelGGL(k_rlt_forward, dim3(st.nBlocks), dim3(64), 0, s, st); }
void launch_rlt_inverse(hipStream_t s, const XfStage& st) { KScope ks_("k_rlt_inverse"); hipLaunchKernelGGL(k_rlt_inverse, dim3(st.nBlocks), dim3(64), 0, s, st); }

}  // namespace knz
