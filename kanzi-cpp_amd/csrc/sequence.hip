// TransformSequence<byte>::forward / inverse bookkeeping (transform/TransformSequence.hpp:88-162,
// :165-247) for a batch of blocks, kept on the device so that a batch needs no host round trip
// between stages: which physical buffer holds each block, its current length, its skip flags and
// the destination capacity the reference would present to the next transform (capacities change
// results for ZRLT/RLT).
#include "common.hpp"
#include "stages.hpp"

namespace knz {

__device__ __forceinline__ const u8* fwd_ptr(u8 where, int b, const u8* in, u64 inStride, u8* A, u8* B, u64 S)
{
    return where == 0 ? in + (size_t)b * inStride : (where == 1 ? A + (size_t)b * S : B + (size_t)b * S);
}

// stage 0 additionally initialises the per-block state
__global__ void k_seq_fwd_prepare(SeqArrays a, int nBlocks, int stage, const u8* in, u64 inStride, u8* A, u8* B, u64 S)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    if (stage == 0) {
        a.where[b] = 0;
        a.swaps[b] = 0;
        a.len[b] = a.origLen[b];
        // blocks <= 15 bytes: tType forced to NONE -> one NullTransform that succeeds -> 0x7F
        // (io/CompressedOutputStream.cpp:691-695)
        const bool copy = a.origLen[b] <= 15;
        a.skip[b] = copy ? 0x7F : 0xFF;
        a.active[b] = copy ? 0 : 1;
    }
    const u8 w = a.where[b];
    a.src[b] = fwd_ptr(w, b, in, inStride, A, B, S);
    a.dst[b] = (w == 1) ? B + (size_t)b * S : A + (size_t)b * S;
    a.alen[b] = a.active[b] ? a.len[b] : 0;
    // TransformSequence.hpp:104-115: out is the task's "buffer" after an even number of swaps, its "data"
    // otherwise; either is replaced by a scratch of requiredSize when shorter than that.
    // (bufCap/dataCap arrive from the host already raised to the block's requiredSize)
    u32 cap = ((a.swaps[b] & 1) == 0) ? a.bufCap[b] : a.dataCap[b];
    if (cap > S) cap = (u32)S;
    a.cap[b] = cap;
    a.ok[b] = 0;
    a.newLen[b] = 0;
}

__global__ void k_seq_fwd_commit(SeqArrays a, int nBlocks, int stage)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    if (!a.active[b] || !a.ok[b]) return;
    a.len[b] = a.newLen[b];
    a.where[b] = (a.where[b] == 1) ? 2 : 1;
    a.swaps[b]++;
    a.skip[b] &= (u8)~(1u << (7 - stage));
}

// NullTransform stage: always succeeds, data stays where it is (transform/NullTransform.hpp:49-66)
__global__ void k_seq_fwd_null(SeqArrays a, int nBlocks, int stage)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    if (!a.active[b]) return;
    a.swaps[b]++;
    a.skip[b] &= (u8)~(1u << (7 - stage));
}

// A stage the HOST has run on the block already (knz_hip_encode_block_hosted): the data is where it was, in its new length; what is left
// to do is what TransformSequence::forward does around a stage -- a swap of the buffers and a cleared skip flag when it succeeded
__global__ void k_seq_fwd_hosted(SeqArrays a, int nBlocks, int stage, int applied)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    if (!a.active[b] || !applied) return;
    a.swaps[b]++;
    a.skip[b] &= (u8)~(1u << (7 - stage));
}

__global__ void k_seq_fwd_finish(SeqArrays a, int nBlocks, const u8* in, u64 inStride, u8* A, u8* B, u64 S, const u8** viewPtr)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    viewPtr[b] = fwd_ptr(a.where[b], b, in, inStride, A, B, S);
}

// Chains made of NullTransforms only (e.g. "-t NONE"): everything k_init_blocks + prepare + null + finish would
// compute, in one launch.  Every stage "succeeds", the data never leaves the caller's buffer.
__global__ void k_seq_fwd_direct(SeqArrays a, u32* origLen, u64 n, u32 blockSize, int nBlocks, int nStages, const u8* in, const u8** viewPtr)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    const u64 off = (u64)b * blockSize;
    const u32 len = (n - off < blockSize) ? (u32)(n - off) : blockSize;
    origLen[b] = len;
    a.len[b] = len;
    a.where[b] = 0;
    const bool copy = len <= 15;                    // io/CompressedOutputStream.cpp:691-695
    u8 skip = copy ? 0x7F : 0xFF;
    if (!copy) for (int i = 0; i < nStages; i++) skip &= (u8)~(1u << (7 - i));
    a.skip[b] = skip;
    viewPtr[b] = in + off;
}

// ---- inverse
__global__ void k_seq_inv_entropy_dst(SeqArrays a, DecBlock* blocks, int nBlocks, u8* out, u64 outStride, u8* A, u64 S, u8** entDst, u32 realMask,
                                      u32 unit, u64 outCap)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    DecBlock& db = blocks[b];
    // realMask: bit (7-i) set for every non-NONE stage i of the sequence
    const bool any = !db.copyBlock && ((~db.skipFlags) & realMask & 0xFF) != 0;
    // A block no inverse stage will touch is entropy-decoded straight into the caller's buffer: it has to fit its
    // own slot there (TransformSequence::inverse refuses count > output room, TransformSequence.hpp:165-247).
    if (!any && !db.error) {
        const u64 at = (u64)b * outStride;
        const u64 room = at < outCap ? outCap - at : 0;
        if (db.preLen > unit || (u64)db.preLen > room) db.error = KNZ_ERR_PROCESS_BLOCK;
    }
    a.skip[b] = db.copyBlock ? 0xFF : (u8)db.skipFlags;
    a.where[b] = any ? 1 : 0;
    a.len[b] = db.preLen;
    entDst[b] = any ? A + (size_t)b * S : out + (size_t)b * outStride;
}

__global__ void k_seq_inv_prepare(SeqArrays a, DecBlock* blocks, int nBlocks, int stage, u8* out, u64 outStride, u8* A, u8* B, u64 S,
                                  u32 capMid, u32 capFinal, u32 realMask, u64 outCap)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    const u8 skip = a.skip[b];
    const bool applied = !blocks[b].error && !((skip >> (7 - stage)) & 1);
    a.active[b] = applied ? 1 : 0;
    a.alen[b] = applied ? a.len[b] : 0;
    bool lower = false;
    for (int j = 0; j < stage; j++) if (!((skip >> (7 - j)) & 1) && ((realMask >> (7 - j)) & 1)) lower = true;
    const u8 w = a.where[b];
    a.src[b] = (w == 1) ? A + (size_t)b * S : B + (size_t)b * S;
    if (lower) { a.dst[b] = (w == 1) ? B + (size_t)b * S : A + (size_t)b * S; a.cap[b] = capMid; }
    else {
        // the last inverse to run writes into the caller's buffer, never past its end (a short last block may
        // leave less than a block of room)
        const u64 off = (u64)b * outStride;
        const u64 room = (outCap > off) ? outCap - off : 0;
        a.dst[b] = out + off;
        a.cap[b] = (room < capFinal) ? (u32)room : capFinal;
    }
    a.swaps[b] = lower ? 1 : 0;      // 1 = result stays in a workspace
    a.ok[b] = 0;
    a.newLen[b] = 0;
}

__global__ void k_seq_inv_commit(SeqArrays a, DecBlock* blocks, int nBlocks, int stage, int ttype)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    if (!a.active[b]) return;
    if (!a.ok[b]) { blocks[b].error = KNZ_ERR_PROCESS_BLOCK; return; }   // all inverse transforms must succeed
    a.len[b] = a.newLen[b];
    blocks[b].preLen = a.newLen[b];
    a.where[b] = a.swaps[b] ? ((a.where[b] == 1) ? 2 : 1) : 0;
}

#define L1D(k, ...) hipLaunchKernelGGL(k, dim3((nBlocks + 255) / 256), dim3(256), 0, s, __VA_ARGS__)

void launch_seq_fwd_direct(hipStream_t s, const SeqArrays& a, u32* origLen, u64 n, u32 blockSize, int nBlocks, int nStages, const u8* in, const u8** viewPtr)
{ KScope ks_("k_seq_fwd_direct"); L1D(k_seq_fwd_direct, a, origLen, n, blockSize, nBlocks, nStages, in, viewPtr); }
void launch_seq_fwd_prepare(hipStream_t s, const SeqArrays& a, int nBlocks, int stage, const u8* in, u64 inStride, u8* A, u8* B, u64 S)
{ KScope ks_("k_seq_fwd_prepare"); L1D(k_seq_fwd_prepare, a, nBlocks, stage, in, inStride, A, B, S); }
void launch_seq_fwd_commit(hipStream_t s, const SeqArrays& a, int nBlocks, int stage)
{ KScope ks_("k_seq_fwd_commit"); L1D(k_seq_fwd_commit, a, nBlocks, stage); }
void launch_seq_fwd_null(hipStream_t s, const SeqArrays& a, int nBlocks, int stage)
{ KScope ks_("k_seq_fwd_null"); L1D(k_seq_fwd_null, a, nBlocks, stage); }
void launch_seq_fwd_hosted(hipStream_t s, const SeqArrays& a, int nBlocks, int stage, int applied)
{ KScope ks_("k_seq_fwd_hosted"); L1D(k_seq_fwd_hosted, a, nBlocks, stage, applied); }
void launch_seq_fwd_finish(hipStream_t s, const SeqArrays& a, int nBlocks, const u8* in, u64 inStride, u8* A, u8* B, u64 S, const u8** viewPtr)
{ KScope ks_("k_seq_fwd_finish"); L1D(k_seq_fwd_finish, a, nBlocks, in, inStride, A, B, S, viewPtr); }
void launch_seq_inv_entropy_dst(hipStream_t s, const SeqArrays& a, DecBlock* blocks, int nBlocks, u8* out, u64 outStride, u8* A, u64 S, u8** entDst, u32 realMask,
                                u32 unit, u64 outCap)
{ KScope ks_("k_seq_inv_entropy_dst"); L1D(k_seq_inv_entropy_dst, a, blocks, nBlocks, out, outStride, A, S, entDst, realMask, unit, outCap); }
void launch_seq_inv_prepare(hipStream_t s, const SeqArrays& a, DecBlock* blocks, int nBlocks, int stage, u8* out, u64 outStride, u8* A, u8* B, u64 S, u32 capMid, u32 capFinal, u32 realMask, u64 outCap)
{ KScope ks_("k_seq_inv_prepare"); L1D(k_seq_inv_prepare, a, blocks, nBlocks, stage, out, outStride, A, B, S, capMid, capFinal, realMask, outCap); }
void launch_seq_inv_commit(hipStream_t s, const SeqArrays& a, DecBlock* blocks, int nBlocks, int stage, int ttype)
{ KScope ks_("k_seq_inv_commit"); L1D(k_seq_inv_commit, a, blocks, nBlocks, stage, ttype); }

}  // namespace knz
