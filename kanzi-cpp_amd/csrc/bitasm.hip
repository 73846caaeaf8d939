// Bit-granular assembly of per-chunk entropy output into the block-ordered stream.
//
// Replaces, for a whole batch of blocks at once, what the reference does serially through
// DefaultOutputBitStream::writeBits (bitstream/DefaultOutputBitStream.cpp:42-128) inside every
// entropy coder and at the end of EncodingTask::run (io/CompressedOutputStream.cpp:852-864):
//   1. k_block_sum   : per block, exclusive prefix sum of chunk bit lengths (LDS scan)
//   2. k_block_scan  : over blocks, 5-bit/lw-bit length prefixes -> absolute bit offsets
//   3. k_assemble    : every chunk funnel-shifts its pieces to its final bit position; interior
//                      32-bit words are plain coalesced stores, boundary words are atomicOr'ed
//                      into the zero-initialised output.
#include "common.hpp"
#include "stages.hpp"
#include <algorithm>

namespace knz {

// ---------------------------------------------------------------- wave-cooperative bit copy
// Copies nbits (MSB-first from src[0]) to bit position dstBit of the big-endian word stream dst.
__device__ void wave_copy_bits(u32* __restrict__ dst, u64 dstBit, const u8* __restrict__ src, u64 nbits)
{
    if (nbits == 0) return;
    const int lane = lane_id();
    const u64 w0 = dstBit >> 5;
    const u64 w1 = (dstBit + nbits - 1) >> 5;
    const u64 srcBytes = (nbits + 7) >> 3;
    for (u64 w = w0 + lane; w <= w1; w += 64) {
        const u64 wb = w << 5;
        const u64 lo = wb > dstBit ? wb : dstBit;
        const u64 hiEnd = dstBit + nbits;
        const u64 hi = (wb + 32) < hiEnd ? (wb + 32) : hiEnd;
        const u32 nb = (u32)(hi - lo);
        const u64 s = lo - dstBit;
        const u64 b0 = s >> 3;
        const u32 sh = (u32)(s & 7);
        // 40 source bits starting at byte b0 (big-endian): one (unaligned) dword + one byte when they are inside
        // the piece, single bytes only at its tail
        u64 win = 0;
        if (b0 + 5 <= srcBytes) {
            u32 x;
            __builtin_memcpy(&x, src + b0, 4);
            win = ((u64)bswap32(x) << 8) | (u64)src[b0 + 4];
        } else {
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const u64 idx = b0 + i;
                const u64 v = idx < srcBytes ? (u64)src[idx] : 0ull;
                win |= v << (32 - 8 * i);
            }
        }
        const u32 val = (u32)((win >> (40 - sh - nb)) & (nb == 32 ? 0xFFFFFFFFull : ((1ull << nb) - 1)));
        const u32 word = val << (u32)((wb + 32) - hi);
        if (nb == 32) dst[w] = bswap32(word);
        else if (word) atomicOr(&dst[w], bswap32(word));
    }
}

// Inline bytes (held in registers/struct) appended by one lane.
__device__ void lane_put_bytes(u32* dst, u64 dstBit, const u32* words, u32 nbytes)
{
    for (u32 i = 0; i < nbytes; i++) {
        const u32 b = (words[i >> 2] >> (8 * (i & 3))) & 0xFF;
        or_bits_mem(dst, dstBit + 8ull * i, b, 8);
    }
}

// ---------------------------------------------------------------- framing helpers
// Block header bits: mode byte (+ skip byte when > 4 transforms) + length + checksum
// (io/CompressedOutputStream.cpp:757-807)
__device__ __forceinline__ u32 block_data_size(u32 postLen)
{
    return (postLen < 256) ? 1u : (u32)(ilog2_u32(postLen) >> 3) + 1u;
}


__global__ __launch_bounds__(256) void k_block_sum(ChunkDesc* __restrict__ desc, BlockInfo* __restrict__ info,
                                                    const u32* __restrict__ blockLen, int maxChunks, u32 chunkSize, u32 slotMul)
{
    const int b = blockIdx.x;
    const u32 len = blockLen[b];
    const u32 nChunks = ((len + chunkSize - 1) / chunkSize) * slotMul;
    __shared__ u64 waveTot[4];
    __shared__ u64 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    ChunkDesc* d = desc + (size_t)b * maxChunks;
    for (u32 base = 0; base < nChunks; base += 256) {
        const u32 c = base + threadIdx.x;
        u64 bits = 0;
        if (c < nChunks) {
            const ChunkDesc& cd = d[c];
            bits = (u64)cd.hdrBits + 8ull * cd.midLen + 8ull * cd.trailerLen;
            for (u32 k = 0; k < cd.nPieces; k++) bits += cd.pieceBits[k];
        }
        const u64 incl = wave_incl_scan64(bits);
        const int wv = threadIdx.x >> 6;
        if (lane_id() == 63) waveTot[wv] = incl;
        __syncthreads();
        u64 off = carry;
        for (int k = 0; k < wv; k++) off += waveTot[k];
        if (c < nChunks) {
            d[c].relBit = off + incl - bits;
            d[c].totalBits = bits;
        }
        __syncthreads();
        if (threadIdx.x == 255) carry = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) info[b].payloadBits = carry;
}

__global__ void k_block_scan(BlockInfo* __restrict__ info, const u32* __restrict__ blockLen, const u32* __restrict__ origLen, int nBlocks,
                             FrameParams fp, u64* __restrict__ totalBits)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u64 cur = fp.prologueBits;
    for (int b = 0; b < nBlocks; b++) {
        BlockInfo bi = info[b];
        if (fp.framing) {
            const u32 postLen = blockLen[b];
            const u32 ds = block_data_size(postLen);
            // (more than four transforms: the skip flags get a byte of their own, except in copy blocks; io/CompressedOutputStream.cpp:791-799)
            const u32 skipByte = (fp.nTransforms > 4 && origLen[b] > 15) ? 8u : 0u;
            bi.hdrBits = 8u + skipByte + 8u * ds + (u32)fp.checksumBits;
            bi.written = (u64)bi.hdrBits + bi.payloadBits;
            bi.lw = (bi.written < 8) ? 3u : (u32)ilog2_u32((u32)(bi.written >> 3)) + 4u;
            bi.bitOff = cur;
            cur += 5u + bi.lw + bi.written;
        } else {
            bi.hdrBits = 0;
            bi.written = bi.payloadBits;
            bi.lw = 0;
            bi.bitOff = cur;
            cur += bi.written;
        }
        info[b] = bi;
    }
    if (fp.framing && fp.finish) cur += 8;
    *totalBits = cur;
}

// One wave per chunk slot. Chunk 0 of every block also writes the block's framing fields.
__global__ __launch_bounds__(64) void k_assemble(const ChunkDesc* __restrict__ desc, const BlockInfo* __restrict__ info,
                                                 const u32* __restrict__ blockLen, const u32* __restrict__ origLen,
                                                 const u8* __restrict__ skipFlags, const u64* __restrict__ checksums, const u8* __restrict__ hdrBase,
                                                 int maxChunks, u32 chunkSize, u32 slotMul, u32 hdrStride, FrameParams fp, u32* __restrict__ out)
{
    const int slot = blockIdx.x;
    const int b = slot / maxChunks;
    const int ci = slot - b * maxChunks;
    const u32 len = blockLen[b];
    const u32 nChunks = ((len + chunkSize - 1) / chunkSize) * slotMul;
    if ((u32)ci >= nChunks) return;
    const BlockInfo bi = info[b];
    const int lane = lane_id();
    u64 pos = bi.bitOff;
    if (fp.framing) {
        if (ci == 0 && lane == 0) {
            // 5 bits lw-3, lw bits written (io/CompressedOutputStream.cpp:852-853)
            or_bits_mem(out, pos, bi.lw - 3, 5);
            or_bits_mem(out, pos + 5, bi.written, bi.lw);
            u64 p = pos + 5 + bi.lw;
            const u32 ds = block_data_size(len);
            u32 mode = ((ds - 1) & 3) << 5;
            const u32 skip = skipFlags[b];
            const bool copy = origLen[b] <= 15;                // copy block (SMALL_BLOCK_SIZE, :691-695)
            if (copy) mode |= 0x80;
            if (copy || fp.nTransforms <= 4) { mode |= (skip >> 4); or_bits_mem(out, p, mode, 8); p += 8; }
            else { mode |= 0x10; or_bits_mem(out, p, mode, 8); or_bits_mem(out, p + 8, skip, 8); p += 16; }
            or_bits_mem(out, p, len, 8 * ds); p += 8 * ds;
            if (fp.checksumBits == 32) or_bits_mem(out, p, checksums[b] & 0xFFFFFFFFull, 32);
            else if (fp.checksumBits == 64) { or_bits_mem(out, p, checksums[b] >> 32, 32); or_bits_mem(out, p + 32, checksums[b] & 0xFFFFFFFFull, 32); }
        }
        pos += 5 + bi.lw + bi.hdrBits;
    }
    const ChunkDesc& cd = desc[slot];
    pos += cd.relBit;
    // hdr bits
    wave_copy_bits(out, pos, hdrBase + (size_t)slot * hdrStride, cd.hdrBits);
    pos += cd.hdrBits;
    if (lane == 0) lane_put_bytes(out, pos, cd.mid, cd.midLen);
    pos += 8ull * cd.midLen;
    for (u32 k = 0; k < cd.nPieces; k++) {
        wave_copy_bits(out, pos, cd.piecePtr[k], cd.pieceBits[k]);
        pos += cd.pieceBits[k];
    }
    if (lane == 0) lane_put_bytes(out, pos, cd.trailer, cd.trailerLen);
}

__global__ void k_put_prologue(u32* out, const u8* prologue, u32 bits)
{
    wave_copy_bits(out, 0, prologue, bits);
}


// ---------------------------------------------------------------- decode side: block walk
// Walks the 5-bit/lw-bit length prefixes (io/CompressedInputStream.cpp:823-856) and parses each
// block's mode byte / length / checksum (:875-919). Serial, one thread: a stream has tens of blocks.
struct WalkResult {
    u64 endBit;
    int64_t nBlocks;
    int32_t ended;       // end marker seen
    int32_t error;
};

__global__ void k_walk_blocks(BitSrc src, u64 startBit, int64_t maxBlocks, int framing, u32 rawLen, int checksumBits,
                              u32 blockSize, DecBlock* __restrict__ blocks, WalkResult* __restrict__ res)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u64 pos = startBit;
    int64_t nb = 0;
    int err = 0;
    int ended = 0;
    int error = 0;
    if (!framing) {
        // per-stage API: one "block" of rawLen bytes, entropy bits start at startBit
        DecBlock db;
        db.payloadBit = startBit; db.bits = src.limitBits - startBit; db.entropyBit = startBit; db.checksum = 0;
        db.preLen = rawLen; db.skipFlags = 0; db.copyBlock = 0; db.error = 0; db.usedBits = 0;
        blocks[0] = db;
        res->endBit = startBit; res->nBlocks = 1; res->ended = 0; res->error = 0;
        return;
    }
    while (nb < maxBlocks) {
        const u32 lr = 3 + take_bits(src, pos, 5, err);
        if (err) { error = KNZ_ERR_READ_FILE; break; }
        u64 len = 0;
        if (lr > 32) { len = (u64)take_bits(src, pos, lr - 32, err) << 32; len |= take_bits(src, pos, 32, err); }
        else len = take_bits(src, pos, lr, err);
        if (err) { error = KNZ_ERR_READ_FILE; break; }
        if (len == 0) { ended = 1; break; }
        if (len > (1ull << 34)) { error = KNZ_ERR_BLOCK_SIZE; break; }
        if (pos + len > src.limitBits) { error = KNZ_ERR_READ_FILE; break; }
        DecBlock db;
        db.payloadBit = pos; db.bits = len; db.error = 0; db.usedBits = 0; db.checksum = 0;
        u64 p = pos;
        BitSrc bs = src;
        bs.limitBits = pos + len;
        int e2 = 0;
        const u32 mode = take_bits(bs, p, 8, e2);
        db.copyBlock = (mode & 0x80) ? 1u : 0u;
        if (db.copyBlock) db.skipFlags = 0xFF;
        else if (mode & 0x10) db.skipFlags = take_bits(bs, p, 8, e2);
        else db.skipFlags = ((mode << 4) | 0x0F) & 0xFF;
        const u32 ds = 1 + ((mode >> 5) & 3);
        db.preLen = take_bits(bs, p, 8 * ds, e2);
        const u32 blkLen = (blockSize + 512 > blockSize + (blockSize >> 4)) ? blockSize + 512 : blockSize + (blockSize >> 4);
        u32 mts = blkLen + blkLen / 2;
        if (mts < 2048) mts = 2048;
        if (mts > (1u << 30)) mts = 1u << 30;
        if (e2 || db.preLen == 0 || db.preLen > mts) db.error = KNZ_ERR_READ_FILE;
        if (checksumBits == 32) db.checksum = take_bits(bs, p, 32, e2);
        else if (checksumBits == 64) { db.checksum = (u64)take_bits(bs, p, 32, e2) << 32; db.checksum |= take_bits(bs, p, 32, e2); }
        if (e2 && !db.error) db.error = KNZ_ERR_PROCESS_BLOCK;
        db.entropyBit = p;
        blocks[nb] = db;
        pos += len;
        nb++;
    }
    res->endBit = pos;
    res->nBlocks = nb;
    res->ended = ended;
    res->error = error;
}

void launch_walk_blocks(hipStream_t s, BitSrc src, u64 startBit, int64_t maxBlocks, int framing, u32 rawLen, int checksumBits,
                        u32 blockSize, DecBlock* blocks, void* res)
{
    { KScope ks_("k_walk_blocks"); hipLaunchKernelGGL(k_walk_blocks, dim3(1), dim3(64), 0, s, src, startBit, maxBlocks, framing, rawLen, checksumBits,
                       blockSize, blocks, reinterpret_cast<WalkResult*>(res)); }
}

void launch_block_sum(hipStream_t s, ChunkDesc* desc, BlockInfo* info, const u32* blockLen, int nBlocks, int maxChunks, u32 chunkSize, u32 slotMul)
{
    { KScope ks_("k_block_sum"); hipLaunchKernelGGL(k_block_sum, dim3(nBlocks), dim3(256), 0, s, desc, info, blockLen, maxChunks, chunkSize, slotMul); }
}

void launch_block_scan(hipStream_t s, BlockInfo* info, const u32* blockLen, const u32* origLen, int nBlocks, FrameParams fp, u64* totalBits)
{
    { KScope ks_("k_block_scan"); hipLaunchKernelGGL(k_block_scan, dim3(1), dim3(64), 0, s, info, blockLen, origLen, nBlocks, fp, totalBits); }
}

void launch_assemble(hipStream_t s, const ChunkDesc* desc, const BlockInfo* info, const u32* blockLen, const u32* origLen, const u8* skipFlags,
                     const u64* checksums, const u8* hdrBase, int nBlocks, int maxChunks, u32 chunkSize, u32 slotMul, u32 hdrStride,
                     FrameParams fp, u32* out)
{
    { KScope ks_("k_assemble"); hipLaunchKernelGGL(k_assemble, dim3(nBlocks * maxChunks), dim3(64), 0, s, desc, info, blockLen, origLen, skipFlags, checksums,
                       hdrBase, maxChunks, chunkSize, slotMul, hdrStride, fp, out); }
}

__global__ void k_init_blocks(u64 n, u32 blockSize, int nBlocks, u32* origLen, u32* blockLen)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    const u64 off = (u64)b * blockSize;
    const u32 len = (n - off < blockSize) ? (u32)(n - off) : blockSize;
    origLen[b] = len;
    blockLen[b] = len;
}

void launch_init_blocks(hipStream_t s, u64 n, u32 blockSize, int nBlocks, u32* origLen, u32* blockLen)
{
    { KScope ks_("k_init_blocks"); hipLaunchKernelGGL(k_init_blocks, dim3((nBlocks + 255) / 256), dim3(256), 0, s, n, blockSize, nBlocks, origLen, blockLen); }
}

// Reject blocks whose declared length cannot be written (guards every later kernel).
__global__ void k_check_prelen(DecBlock* blocks, int nBlocks, u32 maxPre, u64 outCap, u64 outStride)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    DecBlock& db = blocks[b];
    if (db.error) return;
    if (db.preLen > maxPre || (u64)b * outStride + db.preLen > outCap) db.error = KNZ_ERR_PROCESS_BLOCK;
}

void launch_check_prelen(hipStream_t s, DecBlock* blocks, int nBlocks, u32 maxPre, u64 outCap, u64 outStride)
{
    { KScope ks_("k_check_prelen"); hipLaunchKernelGGL(k_check_prelen, dim3((nBlocks + 255) / 256), dim3(256), 0, s, blocks, nBlocks, maxPre, outCap, outStride); }
}

// out = r zero bits (0 < r < 8) followed by the nbits bits of in, both MSB-first byte streams; out gets (r + nbits + 7) / 8 bytes
// (bits of the last byte beyond the stream are zero, as they are in `in`)
__global__ __launch_bounds__(256) void k_shift_bits(const u8* __restrict__ in, u64 nbits, u32 r, u8* __restrict__ out)
{
    const u64 inBytes = (nbits + 7) >> 3, outBytes = (nbits + r + 7) >> 3;
    for (u64 q = (u64)blockIdx.x * 256 + threadIdx.x; q < outBytes; q += (u64)gridDim.x * 256) {
        const u32 a = (q > 0 && q - 1 < inBytes) ? in[q - 1] : 0u;
        const u32 b = (q < inBytes) ? in[q] : 0u;
        out[q] = (u8)((a << (8 - r)) | (b >> r));
    }
}

void launch_shift_bits(hipStream_t s, const u8* in, u64 nbits, u32 r, u8* out)
{
    const u64 outBytes = (nbits + r + 7) >> 3;
    const unsigned grid = (unsigned)std::min<u64>((outBytes + 255) / 256, 65536);
    { KScope ks_("k_shift_bits"); hipLaunchKernelGGL(k_shift_bits, dim3(grid ? grid : 1), dim3(256), 0, s, in, nbits, r, out); }
}

void launch_put_prologue(hipStream_t s, u32* out, const u8* d_prologue, u32 bits)
{
    { KScope ks_("k_put_prologue"); hipLaunchKernelGGL(k_put_prologue, dim3(1), dim3(64), 0, s, out, d_prologue, bits); }
}

}  // namespace knz
