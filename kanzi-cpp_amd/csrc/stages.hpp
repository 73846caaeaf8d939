// Launchers of the device stages (defined in the .hip files next to this header).
#pragma once
#include "common.hpp"

namespace knz {

struct FrameParams {
    int framing;          // 0: raw entropy output of block 0 only (per-stage API), 1: block framing
    int nTransforms;      // number of transform slots in the sequence
    int checksumBits;
    int finish;
    u32 prologueBits;
};

// bitasm.hip
void launch_init_blocks(hipStream_t s, u64 n, u32 blockSize, int nBlocks, u32* origLen, u32* blockLen, u8* skipFlags, u8 skipInit);
void launch_block_sum(hipStream_t s, ChunkDesc* desc, BlockInfo* info, const u32* blockLen, int nBlocks, int maxChunks, u32 chunkSize);
void launch_block_scan(hipStream_t s, BlockInfo* info, const u32* blockLen, int nBlocks, FrameParams fp, u64* totalBits);
void launch_assemble(hipStream_t s, const ChunkDesc* desc, const BlockInfo* info, const u32* blockLen, const u32* origLen,
                     const u8* skipFlags, const u64* checksums, const u8* hdrBase, int nBlocks, int maxChunks, u32 chunkSize,
                     FrameParams fp, u32* out);
void launch_put_prologue(hipStream_t s, u32* out, const u8* d_prologue, u32 bits);
void launch_walk_blocks(hipStream_t s, BitSrc src, u64 startBit, int64_t maxBlocks, int framing, u32 rawLen, int checksumBits,
                        u32 blockSize, DecBlock* blocks, void* res);
void launch_check_prelen(hipStream_t s, DecBlock* blocks, int nBlocks, u32 maxPre, u64 outCap, u64 outStride);

// none.hip
void launch_none_encode(hipStream_t s, BlockView view, int nBlocks, int maxChunks, ChunkDesc* desc);
void launch_none_decode(hipStream_t s, BitSrc src, DecBlock* blocks, int nBlocks, u8* out, u64 outStride);

// ans.hip
void launch_ans0_encode(hipStream_t s, BlockView view, int nBlocks, int maxChunks, ChunkDesc* desc, uint2* encTab, u8* tmp);
void launch_ans0_decode(hipStream_t s, BitSrc src, DecBlock* blocks, int nBlocks, int maxChunks, void* chunkMeta, u8* out, u64 outStride);
size_t ans0_dec_chunk_bytes();

}  // namespace knz
