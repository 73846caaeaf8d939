// rANS order-1 (kanzi "ANS1") on gfx950.
//
// Reference being replaced (results are bit-identical):
//   encoder  entropy/ANSRangeEncoder.cpp:59-67 (chunk 16384<<8 = 4 MiB, logRange 11, 256 contexts),
//            :158-192 (encode), :264-287 (rebuildStatistics), :83-116 (updateFrequencies, one table per
//            context), :119-155 (encodeHeader), :194-261 (encodeChunk, order-1 branch);
//            Global.cpp:223-271 (order-1 histogram with totals); entropy/EntropyUtils.cpp:57-89,131-245
//   decoder  entropy/ANSRangeDecoder.cpp:80-175 (decodeHeader), :177-216 (decode), :218-292 (decodeChunk)
//
// The format leaves little parallelism inside a chunk: 4 MiB share one set of 4 interleaved states, so the
// coding loop is a 1 Mi-step dependent chain per chunk.  What can be parallel is: the order-1 histogram
// (global atomics over the whole chunk), the 256 context tables of a chunk (one wave each: normalisation,
// header bits, encoder reciprocals / decoder slot tables), and the chunks themselves.  The coding loops run
// one wave per chunk: lanes 0-3 carry the states, all 64 lanes prefetch what the next steps will need.
#include "common.hpp"
#include "stages.hpp"
#include "ans_common.hpp"

namespace knz {

constexpr u32 A1_TOP = 1u << 15;
constexpr u32 A1_LR = 11;
constexpr u32 A1_SCALE = 1u << A1_LR;

// ------------------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------------------
// Global.cpp:223-271 as called from ANSRangeEncoder.cpp:264-287: the chunk is histogrammed as 4 quarters of
// end>>2 bytes; the first byte of every quarter is counted in context 0, the (end & 3) tail bytes are not
// counted at all.  A chunk shorter than 4 bytes is one "quarter".
__global__ __launch_bounds__(256) void k_ans1_hist(BlockView view, int chunksPerBlock, u32* __restrict__ hist)
{
    const int gc = blockIdx.y;
    const int b = gc / chunksPerBlock;
    const int ci = gc - b * chunksPerBlock;
    const u32 len = view.len[b];
    const u32 start = (u32)ci * ANS1_CHUNK;
    if (len <= 32 || start >= len) return;
    const u32 end = (len - start < ANS1_CHUNK) ? (len - start) : ANS1_CHUNK;
    const u8* blk = view.ptr[b] + start;
    const u32 quarter = end >> 2;
    const u32 counted = quarter ? 4 * quarter : end;
    u32* h = hist + (size_t)gc * 65536;
    // every thread walks its own contiguous slice and merges repeats of the same (context, symbol) pair in a
    // register, so that runs (the worst case for same-address atomics) cost one atomic per run and thread
    const u32 part = (counted + gridDim.x - 1) / gridDim.x;
    const u32 lo = blockIdx.x * part;
    const u32 hi = (lo + part < counted) ? lo + part : counted;
    const u32 per = (part + 255) / 256;
    u32 i = lo + threadIdx.x * per;
    const u32 iend = (i + per < hi) ? i + per : hi;
    u32 curKey = 0xFFFFFFFFu, curCnt = 0;
    for (; i < iend; i++) {
        const bool first = quarter ? (i % quarter == 0) : (i == 0);
        const u32 ctx = first ? 0u : (u32)blk[i - 1];
        const u32 key = ctx * 256 + blk[i];
        if (key == curKey) { curCnt++; continue; }
        if (curCnt) atomicAdd(&h[curKey], curCnt);
        curKey = key; curCnt = 1;
    }
    if (curCnt) atomicAdd(&h[curKey], curCnt);
}

// one wave per (chunk, context): normalise the row, emit its header bits, build its encoder table
__global__ __launch_bounds__(64) void k_ans1_ctx(BlockView view, int chunksPerBlock, const u32* __restrict__ hist,
                                                 ChunkDesc* __restrict__ desc, uint2* __restrict__ encTab, u8* __restrict__ hdrBuf)
{
    const int gc = blockIdx.x >> 8;
    const u32 ctx = blockIdx.x & 255;
    const int b = gc / chunksPerBlock;
    const int ci = gc - b * chunksPerBlock;
    const u32 len = view.len[b];
    const u32 start = (u32)ci * ANS1_CHUNK;
    if (len <= 32 || start >= len) return;
    const int lane = lane_id();
    const size_t slot = (size_t)gc * ANS1_SLOTS + ctx;

    __shared__ u32 hdrw[HDR_WORDS];
    __shared__ u32 grpMax[64];
    __shared__ u32 grpOff[64];
    for (int i = lane; i < (int)HDR_WORDS; i += 64) hdrw[i] = 0;
    grpMax[lane] = 0;
    __syncthreads();

    const uint4 row = reinterpret_cast<const uint4*>(hist + (size_t)gc * 65536 + ctx * 256)[lane];
    u32 f[4] = { row.x, row.y, row.z, row.w };
    const u32 n = wave_sum(f[0] + f[1] + f[2] + f[3]);
    u32 present = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) present |= (f[k] != 0 ? 1u : 0u) << k;
    const u32 myCount = __popc(present);
    const u32 inclCount = wave_incl_scan(myCount);
    const u32 asz = (u32)__shfl((int)inclCount, 63, 64);
    const u32 rankBase = inclCount - myCount;
    if (asz) ans_normalize<A1_LR>(lane, f, n, asz);
    const u32 pos = ans_header_bits<A1_LR>(lane, f, present, asz, rankBase, ctx == 0, hdrw, grpMax, grpOff, 0);
    __syncthreads();
    u32* hdrOut = reinterpret_cast<u32*>(hdrBuf + slot * HDR_BYTES);
    const u32 hdrWordsUsed = (pos + 31) >> 5;
    for (u32 i = lane; i < hdrWordsUsed; i += 64) hdrOut[i] = bswap32(hdrw[i]);
    if (asz) ans_enc_table<A1_LR>(lane, f, encTab + ((size_t)gc * 256 + ctx) * 256);
    if (lane == 0) { desc[slot].hdrBits = pos; desc[slot].aux = asz; }
}

// ANSRangeEncoder.cpp:194-261 (order-1 branch): state j walks quarter j backwards; the symbol at position p is
// coded in the context of the byte before it, the first byte of a quarter in context 0.
//
// One wave per chunk, lanes 0-3 are the four states. What the states need -- the (reciprocal, frequency, shift, bias) record of
// every (context, symbol) pair they will meet -- is known up front, so all 64 lanes fetch it a stretch of 64 steps ahead (bytes
// two stretches ahead, records one) into LDS, unpacked into the operands the recurrence uses, and the serial part is the state
// recurrence alone: compare, shift, multiply-high, multiply, add, and one LDS write of (low 16 bits | flag). Where the 16-bit
// emissions go is settled after the stretch by the whole wave (lane = step: a scan over the flags), which also writes them out:
// a store inside the step loop shares the memory counter with the record loads, so every wait for a load was a wait for the store
// before it to reach memory (round 4: 2.2 us per 16 steps, 142 ms per 4 MiB chunk); the text is read through the global address
// space for the same reason (a flat load counts as an LDS operation, and the wait for the next record then waited for memory).
// every load issued so far has arrived (used before a burst of stores, so that the wait for a load that follows in program order is not
// a wait for the stores: one counter for both on this hardware)
// (KNZ_LOADS_DONE: common.hpp)
constexpr u32 A1E_ST = 64;                     // steps per stretch (= lanes: one step per lane when the emissions are placed)

__global__ __launch_bounds__(64) void k_ans1_encode(BlockView view, int chunksPerBlock, ChunkDesc* __restrict__ desc,
                                                    const uint2* __restrict__ encTab, u8* __restrict__ payBase, u64 payStride)
{
    const int gc = blockIdx.x;
    const int b = gc / chunksPerBlock;
    const int ci = gc - b * chunksPerBlock;
    const u32 len = view.len[b];
    const u32 start = (u32)ci * ANS1_CHUNK;
    const int lane = lane_id();
    if (len <= 32) {
        // ANSRangeEncoder.cpp:160-163 : tiny block stored raw (also the <= 15 byte copy-block mode)
        if (ci == 0 && lane == 0) {
            ChunkDesc* c0 = desc + (size_t)gc * ANS1_SLOTS;
            c0->nPieces = 1; c0->pieceBits[0] = 8 * len; c0->piecePtr[0] = view.ptr[b];
        }
        return;
    }
    if (start >= len) return;
    const u32 end = (len - start < ANS1_CHUNK) ? (len - start) : ANS1_CHUNK;
    const u8* blk = view.ptr[b] + start;
    const u32 end4 = end & ~3u;
    const u32 quarter = end4 >> 2;
    const u32 tail = end - end4;
    u8* pay = payBase + (size_t)gc * payStride;
    // exclusive end of the payload area; shifted by one when the raw tail is odd so that every 16-bit
    // emission lands on an even address
    const u32 top = (u32)payStride - (tail & 1);
    if (lane == 0) for (u32 t = 0; t < tail; t++) pay[top - tail + t] = blk[end4 + t];   // ANSRangeEncoder.cpp:204-205
    const u32 itemsTop = top - tail;                 // item i (in emission order) lies at itemsTop - 2 (i + 1)
    u32 nOut = 0;                                    // items emitted so far (the same in all lanes)
    u32 st = A1_TOP;
    const uint2* tab = encTab + (size_t)gc * 65536;

    __shared__ uint4 ebuf[2][A1E_ST][4];             // per (step, state): reciprocal, xMax, 2^11 - freq, bias | shift << 16
    __shared__ u32 obuf[A1E_ST][4];                  // per (step, state): low 16 bits of the state before the step | emitted << 16
    const u32 j = (u32)lane & 3, t4 = ((u32)lane >> 2) * 4;
    const u32 total = quarter;                       // steps per state
    const u32 qs = j * quarter;
    // table index of step s for state j (clamped so that idle lanes still load inside the chunk)
    auto step_index = [&](u32 s) -> u32 {
        const u32 sc = (s < total) ? s : total - 1;
        const u32 pos = qs + quarter - 1 - sc;
        const u32 sym = ldg<u8>(blk + pos);
        const u32 cb = ldg<u8>(blk + (pos > qs ? pos - 1 : pos));
        return ((pos > qs) ? cb : 0u) * 256 + sym;
    };
    auto operands = [](uint2 e) -> uint4 {
        const u32 fr = e.y & 0x1FFF;
        // xMax = ((TOP >> 11) << 16) * freq
        return make_uint4(e.x, fr << 20, A1_SCALE - fr, (e.y >> 17) | (((e.y >> 13) & 0xF) << 16));
    };
    if (total) {
        const u32 nStretch = (total + A1E_ST - 1) / A1E_ST;
        u32 idx1[4];
        uint2 e1[4];
#pragma unroll
        for (int k = 0; k < 4; k++) ebuf[0][t4 + k][j] = operands(tab[step_index(t4 + k)]);
#pragma unroll
        for (int k = 0; k < 4; k++) idx1[k] = step_index(A1E_ST + t4 + k);
        __syncthreads();
        for (u32 r = 0; r < nStretch; r++) {
            const u32 buf = r & 1;
            const u32 sBase = A1E_ST * r;
            const u32 cnt = (total - sBase < A1E_ST) ? (total - sBase) : A1E_ST;
            // in flight during the serial part: the records of stretch r + 1, the bytes of stretch r + 2
            u32 idx2[4];
#pragma unroll
            for (int k = 0; k < 4; k++) e1[k] = tab[idx1[k]];
#pragma unroll
            for (int k = 0; k < 4; k++) idx2[k] = step_index(A1E_ST * (r + 2) + t4 + k);
            if (lane < 4) {
                uint4 e = ebuf[buf][0][lane];
                for (u32 tt = 0; tt < cnt; tt++) {
                    const uint4 eN = ebuf[buf][(tt + 1) & (A1E_ST - 1)][lane];       // (the record of the next step: its LDS latency is off the chain)
                    const bool flag = st >= e.y;
                    obuf[tt][lane] = (st & 0xFFFFu) | (flag ? 0x10000u : 0u);
                    st = flag ? (st >> 16) : st;
                    const u32 qd = __umulhi(st, e.x) >> (e.w >> 16);
                    st = st + (e.w & 0xFFFFu) + qd * e.z;
                    e = eN;
                }
            }
            __syncthreads();
            // the records of the next stretch go to LDS BEFORE this stretch's emissions are stored: the wait for the loads then does
            // not include those stores (they have the whole next stretch to reach memory)
            KNZ_LOADS_DONE();
#pragma unroll
            for (int k = 0; k < 4; k++) ebuf[buf ^ 1][t4 + k][j] = operands(e1[k]);
#pragma unroll
            for (int k = 0; k < 4; k++) idx1[k] = idx2[k];
            // lane = step: lower states write first, i.e. at the higher addresses of the payload that grows downwards
            {
                const uint4 o = ((u32)lane < cnt) ? *reinterpret_cast<const uint4*>(&obuf[lane][0]) : make_uint4(0, 0, 0, 0);
                const u32 ov[4] = { o.x, o.y, o.z, o.w };
                const u32 mine = (o.x >> 16) + (o.y >> 16) + (o.z >> 16) + (o.w >> 16);
                const u32 incl = wave_incl_scan(mine);
                u32 i = nOut + incl - mine;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (ov[k] >> 16) {
                        stg<u16>(pay + (itemsTop - 2 * (i + 1)), (u16)(((ov[k] >> 8) & 0xFF) | ((ov[k] & 0xFF) << 8)));
                        i++;
                    }
                }
                nOut += (u32)__shfl((int)incl, 63, 64);
            }
            __syncthreads();
        }
    }
    const u32 q = itemsTop - 2 * nOut;
    // varint(size) + 4 states -> mid ; payload piece
    const u32 s0 = (u32)__shfl((int)st, 0, 64);
    const u32 s1v = (u32)__shfl((int)st, 1, 64);
    const u32 s2v = (u32)__shfl((int)st, 2, 64);
    const u32 s3v = (u32)__shfl((int)st, 3, 64);
    if (lane == 0) {
        ChunkDesc* cd = desc + (size_t)gc * ANS1_SLOTS + 256;
        const u32 sz = top - q;
        u8 mid[24];
        u32 ml = 0;
        u32 v = sz;
        while (v >= 128) { mid[ml++] = (u8)(0x80 | (v & 0x7F)); v >>= 7; }
        mid[ml++] = (u8)v;
        const u32 sts[4] = { s0, s1v, s2v, s3v };
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

size_t ans1_hist_bytes(size_t nChunks) { return nChunks * 65536 * sizeof(u32); }
size_t ans1_enctab_bytes(size_t nChunks) { return nChunks * 65536 * sizeof(uint2); }

void launch_ans1_encode(hipStream_t s, BlockView view, int nBlocks, int chunksPerBlock, ChunkDesc* desc, const Ans1EncWs& ws)
{
    const int nCh = nBlocks * chunksPerBlock;
    (void)hipMemsetAsync(ws.hist, 0, ans1_hist_bytes((size_t)nCh), s);
    (void)hipMemsetAsync(desc, 0, sizeof(ChunkDesc) * (size_t)nCh * ANS1_SLOTS, s);
    { KScope ks_("k_ans1_hist"); hipLaunchKernelGGL(k_ans1_hist, dim3(64, nCh), dim3(256), 0, s, view, chunksPerBlock, ws.hist); }
    { KScope ks_("k_ans1_ctx"); hipLaunchKernelGGL(k_ans1_ctx, dim3(nCh * 256), dim3(64), 0, s, view, chunksPerBlock, ws.hist, desc, ws.encTab, ws.hdr); }
    { KScope ks_("k_ans1_encode"); hipLaunchKernelGGL(k_ans1_encode, dim3(nCh), dim3(64), 0, s, view, chunksPerBlock, desc, ws.encTab, ws.pay, ws.payStride); }
}

}  // namespace knz
