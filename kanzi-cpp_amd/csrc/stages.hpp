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
void launch_init_blocks(hipStream_t s, u64 n, u32 blockSize, int nBlocks, u32* origLen, u32* blockLen);
// A block owns ceil(len / chunkSize) * slotMul consecutive ChunkDesc slots (slotMul > 1: ANS1, one slot per context table).
void launch_block_sum(hipStream_t s, ChunkDesc* desc, BlockInfo* info, const u32* blockLen, int nBlocks, int maxChunks, u32 chunkSize, u32 slotMul);
void launch_block_scan(hipStream_t s, BlockInfo* info, const u32* blockLen, const u32* origLen, int nBlocks, FrameParams fp, u64* totalBits);
void launch_assemble(hipStream_t s, const ChunkDesc* desc, const BlockInfo* info, const u32* blockLen, const u32* origLen,
                     const u8* skipFlags, const u64* checksums, const u8* hdrBase, int nBlocks, int maxChunks, u32 chunkSize,
                     u32 slotMul, u32 hdrStride, FrameParams fp, u32* out);
void launch_put_prologue(hipStream_t s, u32* out, const u8* d_prologue, u32 bits);
void launch_shift_bits(hipStream_t s, const u8* in, u64 nbits, u32 r, u8* out);
void launch_walk_blocks(hipStream_t s, BitSrc src, u64 startBit, int64_t maxBlocks, int framing, u32 rawLen, int checksumBits,
                        u32 blockSize, DecBlock* blocks, void* res);
void launch_check_prelen(hipStream_t s, DecBlock* blocks, int nBlocks, u32 maxPre, u64 outCap, u64 outStride);

// none.hip
void launch_none_encode(hipStream_t s, BlockView view, int nBlocks, int maxChunks, ChunkDesc* desc);
void launch_none_decode(hipStream_t s, BitSrc src, DecBlock* blocks, int nBlocks, u8* const* outPtr);

// ans.hip
void launch_ans0_encode(hipStream_t s, BlockView view, int nBlocks, int maxChunks, ChunkDesc* desc, uint2* encTab, u8* tmp);
void launch_ans0_decode(hipStream_t s, BitSrc src, DecBlock* blocks, int nBlocks, int maxChunks, void* chunkMeta, u8* const* outPtr);
size_t ans0_dec_chunk_bytes();

// ans1.hip (order-1 rANS: 4 MiB chunks, 256 context tables per chunk)
constexpr u32 ANS1_CHUNK = 4u << 20;
constexpr u32 ANS1_SLOTS = 257;              // ChunkDesc slots per chunk: 256 context headers + (var-int, states, payload)
struct Ans1EncWs { u32* hist; uint2* encTab; u8* hdr; u8* pay; u64 payStride; };
void launch_ans1_encode(hipStream_t s, BlockView view, int nBlocks, int chunksPerBlock, ChunkDesc* desc, const Ans1EncWs& ws);
size_t ans1_hist_bytes(size_t nChunks);
size_t ans1_enctab_bytes(size_t nChunks);
struct Ans1DecWs { void* meta; u32* slotTab; };
void launch_ans1_decode(hipStream_t s, BitSrc src, DecBlock* blocks, int nBlocks, int chunksPerBlock, const Ans1DecWs& ws, u8* const* outPtr);
size_t ans1_meta_bytes(size_t nChunks);
size_t ans1_slottab_bytes(size_t nChunks);

// One transform stage over a batch (all arrays are device arrays indexed by block).
struct XfStage {
    const u8* const* src;
    u8* const* dst;
    const u32* len;          // active length (0 = block does not take part in this stage)
    const u32* cap;          // destination capacity seen by the transform
    u8* ok;                  // out: 1 when the reference's forward()/inverse() would return true
    u32* newLen;             // out: bytes produced
    int nBlocks;
    u32 maxLen;              // upper bound of len[] (grid sizing)
    u32* scratchU32;         // stage scratch (8-byte aligned)
    int entropyType;         // stream entropy id (RLT escape choice)
    int bsVersion = 6;       // bitstream version the blocks come from (inverse only: BWT block header of versions below 6)
    u32 maxCap = 0;          // upper bound of cap[] when the host knows one (inverse stages that size scratch by their output)
};

// zrlt_mtft.hip
void launch_zrlt_forward(hipStream_t s, const XfStage& st);
void launch_zrlt_inverse(hipStream_t s, const XfStage& st);
void launch_mtft_forward(hipStream_t s, const XfStage& st);
void launch_mtft_inverse(hipStream_t s, const XfStage& st);
size_t zrlt_scratch_u32(int nBlocks, u32 maxLen);
size_t mtft_scratch_u32(int nBlocks, u32 maxLen);
int mtft_tune_chain(int on);                                 // 1 = forward ranks by the byte-serial chain kernel (knz_hip_tune "mtf_chain")
int mtft_tune(int tileBytes);                                // 0 = tile size by batch size, 1024 / 4096 force it (knz_hip_tune "mtf_tile")

// bwt.hip (these synchronise the stream: active-set sizes are read back through h_pinned)
int launch_bwt_forward(hipStream_t s, const XfStage& st, void* scratch, size_t scratchBytes, u32* h_pinned);
int launch_bwt_inverse(hipStream_t s, const XfStage& st, void* scratch, size_t scratchBytes, u32* h_pinned);
size_t bwt_forward_scratch_bytes(int nBlocks, u32 VS, size_t total);
size_t bwt_inverse_scratch_bytes(int nBlocks, u32 VS, size_t total);
int bwt_forward_tune(const char* key, int value);          // developer knobs of the suffix sort (knz_hip_tune)

// fpaq.hip (probs: fpaq_probs_bytes(nBlocks, S) bytes of scratch, S = upper bound of the block lengths)
void launch_fpaq_encode(hipStream_t s, BlockView view, const u32* origLen, u32 copyThreshold, int nBlocks, int maxChunks, ChunkDesc* desc, u8* tmp, u64 tmpStride,
                        u16* probs, u64 S);
size_t fpaq_probs_bytes(int nBlocks, u64 S);
void launch_fpaq_decode(hipStream_t s, BitSrc src, DecBlock* blocks, int nBlocks, u8* const* outPtr);
void launch_srt_forward(hipStream_t s, const XfStage& st);          // scratch: srt_scratch_u32(nBlocks, maxLen) words
size_t srt_scratch_u32(int nBlocks, u32 maxLen);
void launch_srt_inverse(hipStream_t s, const XfStage& st);          // scratch: srt_inverse_scratch_u32(nBlocks, maxLen) words
size_t srt_inverse_scratch_u32(int nBlocks, u32 maxLen);
void launch_rlt_forward(hipStream_t s, const XfStage& st);
void launch_rlt_inverse(hipStream_t s, const XfStage& st);

// sbrt.hip (mode 2 = RANK, 3 = TIMESTAMP; 1 = MTF for cross-checks)
void launch_sbrt_forward(hipStream_t s, const XfStage& st, int mode);
void launch_sbrt_inverse(hipStream_t s, const XfStage& st, int mode);

// lz.hip (ttype = KNZ_T_LZ or KNZ_T_LZX)
int launch_lz_forward(hipStream_t s, const XfStage& st, int ttype, void* scratch, size_t scratchBytes);
void launch_lz_inverse(hipStream_t s, const XfStage& st, void* scratch, size_t scratchBytes, u32 maxCap);   // no scratch: serial decoder only
size_t lz_inverse_scratch_bytes(int nBlocks, u32 maxCap);
int lz_serial_decode(int set);                               // knob "lz_serial_decode": >= 0 sets it, returns the value (1 = one wave per block only)
size_t lz_forward_scratch_bytes(int ttype, int nBlocks, u32 maxLen);

// xxhash.hip
void launch_xxhash(hipStream_t s, const u8* const* ptr, const u32* lens, int nBlocks, int bits, u64* out);
void launch_block_ptrs(hipStream_t s, const u8* base, u64 stride, int nBlocks, const u8** ptr);
void launch_verify_checksums(hipStream_t s, DecBlock* blocks, int nBlocks, int bits, const u8* out, u64 outStride, const u8** ptrScratch,
                             u32* lenScratch, u64* sumScratch);

// sequence.hip : TransformSequence bookkeeping on the device
struct SeqArrays {
    u8* where;               // 0 = caller buffer, 1 = workspace A, 2 = workspace B
    u8* swaps;
    u8* active;
    u32* len;                // current data length per block
    u32* alen;               // active length for the current stage
    u8* skip;                // skip flags (bit 7-i set = stage i not applied)
    const u8** src;
    u8** dst;
    u32* cap;
    u8* ok;
    u32* newLen;
    const u32* origLen;
    const u32* dataCap;      // reference buffer capacities (forward only)
    const u32* bufCap;
};
void launch_seq_fwd_direct(hipStream_t s, const SeqArrays& a, u32* origLen, u64 n, u32 blockSize, int nBlocks, int nStages, const u8* in, const u8** viewPtr);
void launch_seq_fwd_prepare(hipStream_t s, const SeqArrays& a, int nBlocks, int stage, const u8* in, u64 inStride, u8* A, u8* B, u64 S);
void launch_seq_fwd_null(hipStream_t s, const SeqArrays& a, int nBlocks, int stage);
void launch_seq_fwd_hosted(hipStream_t s, const SeqArrays& a, int nBlocks, int stage, int applied);
void launch_seq_fwd_commit(hipStream_t s, const SeqArrays& a, int nBlocks, int stage);
void launch_seq_fwd_finish(hipStream_t s, const SeqArrays& a, int nBlocks, const u8* in, u64 inStride, u8* A, u8* B, u64 S, const u8** viewPtr);
void launch_seq_inv_entropy_dst(hipStream_t s, const SeqArrays& a, DecBlock* blocks, int nBlocks, u8* out, u64 outStride, u8* A, u64 S, u8** entDst, u32 realMask, u32 unit, u64 outCap);
void launch_seq_inv_prepare(hipStream_t s, const SeqArrays& a, DecBlock* blocks, int nBlocks, int stage, u8* out, u64 outStride, u8* A, u8* B, u64 S, u32 capMid, u32 capFinal, u32 realMask, u64 outCap);
void launch_seq_inv_commit(hipStream_t s, const SeqArrays& a, DecBlock* blocks, int nBlocks, int stage, int ttype);

// huffman.hip
void launch_huffman_encode(hipStream_t s, BlockView view, int nBlocks, int maxChunks, ChunkDesc* desc, u8* tmp);
void launch_huffman_decode(hipStream_t s, BitSrc src, DecBlock* blocks, int nBlocks, int maxChunks, void* chunkMeta, u8* const* outPtr, int bsVersion = 6);
size_t huffman_dec_chunk_bytes();

}  // namespace knz
