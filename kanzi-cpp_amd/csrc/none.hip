// NONE entropy codec (entropy/NullEntropyEncoder.hpp, NullEntropyDecoder.hpp): the block's bytes are
// copied to / from the bit stream unchanged, at whatever bit offset the framing gives them.
#include "common.hpp"
#include "stages.hpp"

namespace knz {

__global__ void k_none_encode(BlockView view, int nBlocks, int maxChunks, ChunkDesc* desc)
{
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= nBlocks * maxChunks) return;
    const int b = slot / maxChunks;
    const int ci = slot - b * maxChunks;
    const u32 len = view.len[b];
    const u32 start = (u32)ci * ENT_CHUNK;
    if (start >= len) return;
    ChunkDesc* cd = desc + slot;
    const u32 n = (len - start < ENT_CHUNK) ? (len - start) : ENT_CHUNK;
    cd->hdrBits = 0; cd->midLen = 0; cd->trailerLen = 0; cd->aux = 0;
    cd->nPieces = 1; cd->pieceBits[0] = 8 * n; cd->piecePtr[0] = view.ptr[b] + start;
}

__global__ __launch_bounds__(256) void k_none_decode(BitSrc src, DecBlock* blocks, u8* const* outPtr)
{
    const int b = blockIdx.y;
    DecBlock& db = blocks[b];
    if (db.error) return;
    const u32 n = db.preLen;
    const u64 limit = db.payloadBit + ((db.bits + 7) & ~7ull);
    if (db.entropyBit + 8ull * n > limit) { if (threadIdx.x == 0 && blockIdx.x == 0) db.error = KNZ_ERR_PROCESS_BLOCK; return; }
    u8* dst = outPtr[b];
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[i] = (u8)peek_bits(src, db.entropyBit + 8ull * i, 8);
    if (threadIdx.x == 0 && blockIdx.x == 0) db.usedBits = 8ull * n;
}

void launch_none_encode(hipStream_t s, BlockView view, int nBlocks, int maxChunks, ChunkDesc* desc)
{
    const int nSlots = nBlocks * maxChunks;
    { KScope ks_("k_none_encode"); hipLaunchKernelGGL(k_none_encode, dim3((nSlots + 255) / 256), dim3(256), 0, s, view, nBlocks, maxChunks, desc); }
}

void launch_none_decode(hipStream_t s, BitSrc src, DecBlock* blocks, int nBlocks, u8* const* outPtr)
{
    { KScope ks_("k_none_decode"); hipLaunchKernelGGL(k_none_decode, dim3(64, nBlocks), dim3(256), 0, s, src, blocks, outPtr); }
}

}  // namespace knz
