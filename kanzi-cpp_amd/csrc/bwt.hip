// Inverse Burrows-Wheeler transform (block codec) on gfx950, all blocks of a batch at once (forward: bwt_fwd.hip).
//
// Reference being replaced (bit-identical results; the suffix array of a block is unique, so any
// correct construction matches divsufsort):
//   forward  transform/BWTBlockCodec.cpp:32-87 (header), transform/BWT.cpp:92-134,
//            transform/DivSufSort.cpp:171-295 (computeBWT / constructBWT): "proper prefix sorts first",
//            out[0] = in[n-1], then in[SA[r]-1] for ranks with SA[r] != 0, 8 primary indexes
//            rank(suffix k*ceil(n/8)) + 1 (BWT.hpp:40-61)
//   inverse  transform/BWTBlockCodec.cpp:89-168, transform/BWT.cpp:136-166, :169-292 (mergeTPSI),
//            :295-492 (biPSIv2): both walk the psi permutation from the primary indexes
//
// GPU formulation
//   inverse  psi by a stable counting sort of the BWT symbols (tile histograms + ballot ranks), then splitter-based list ranking (Helman-JaJa):
//            the 8 chains the reference walks serially become ~n/64 independent ones.
#include "common.hpp"
#include "stages.hpp"
#include "bwt_common.hpp"

#include <cstring>
#include <algorithm>
#include "prims.hpp"

namespace knz {

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, s

// ------------------------------------------------------------------------------------------------
// inverse
// ------------------------------------------------------------------------------------------------
struct BwtHdr { u32 n; u32 hdr; u32 pIdx; u32 okFlag; };

struct InvInfo { u32 total; u32 nWords; u32 count; u32 dyn; u32 pad[4]; };      // device-resident sizes: no host read-back in this stage

template <bool OLD>
__global__ void k_bwt_i_header(BwtView v, BwtHdr* __restrict__ hd, u32* __restrict__ base, u8* __restrict__ ok, u32* __restrict__ newLen, InvInfo* __restrict__ info)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u32 sum = 0;
    for (int b = 0; b < v.nBlocks; b++) {
        base[b] = sum;
        BwtHdr h; h.n = 0; h.hdr = 0; h.pIdx = 0; h.okFlag = 0;
        const u32 blockSize = v.len[b];
        newLen[b] = 0;
        ok[b] = 0;
        if (blockSize == 0) { hd[b] = h; ok[b] = 1; continue; }     // nothing to do / inactive
        if (OLD) {
            // the header of bitstream versions below 6 (BWTBlockCodec.cpp:140-164): the chunk count follows from the length WITH the
            // header, every chunk has a mode byte (top two bits: bytes of its primary index - 1, low six bits: the index's top bits)
            // and the rest of the index, stored as it is (since version 6: index - 1, behind one mode byte for all chunks)
            const u8* s = v.src[b];
            const u32 chunks = (u32)bwt_chunks(blockSize);
            u32 prim[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
            u32 left = blockSize, idx = 0;
            bool good = chunks <= 8;
            for (u32 k = 0; k < chunks && good; k++) {
                if (idx >= blockSize) { good = false; break; }
                const u32 mode = s[idx++];
                const u32 sz = 1 + ((mode >> 6) & 3);
                if (left < sz || idx + (sz - 1) > blockSize) { good = false; break; }
                left -= sz;
                u32 shift = (sz - 1) << 3;
                u32 pi = (mode & 0x3F) << shift;
                for (u32 q = 1; q < sz; q++) { shift -= 8; pi |= (u32)s[idx++] << shift; }
                prim[k] = pi;
            }
            const u32 n = left;
            if (good && n > 1) {
                // BWT.cpp:176-177 (chunk 0), :230-245 / :310-318 (every chunk the data itself is cut into)
                if (prim[0] == 0 || prim[0] > n) good = false;
                if (bwt_chunks(n) == 8) for (u32 k = 1; k < 8; k++) if (prim[k] == 0 || prim[k] > n) good = false;
            }
            if (good && n > v.cap[b]) good = false;
            if (good) { h.n = n; h.hdr = idx; h.pIdx = prim[0]; h.okFlag = 1; ok[b] = 1; newLen[b] = n; if (n >= 2) sum += n; }
        } else if (blockSize >= 2) {
            const u8* s = v.src[b];
            const u32 mode = s[0];
            const u32 logNbChunks = (mode >> 2) & 7;
            const u32 pIndexSize = (mode & 3) + 1;
            const u32 chunks = 1u << logNbChunks;
            const u32 headerSize = 1 + chunks * pIndexSize;
            bool good = (blockSize >= headerSize);
            u32 n = 0;
            if (good) {
                n = blockSize - headerSize;
                good = (chunks == (u32)bwt_chunks(n)) && chunks <= 8;
            }
            u32 p0 = 0;
            if (good) {
                u32 idx = 1;
                for (u32 k = 0; k < chunks && good; k++) {
                    u32 pi = 0;
                    for (u32 q = 0; q < pIndexSize; q++) pi = (pi << 8) | s[idx++];
                    if (pi >= 0x7FFFFFFFu) good = false;
                    const u32 prim = pi + 1;
                    if (k == 0) p0 = prim;
                    // BWT.cpp:176-177 (mergeTPSI, chunk 0), :230-245 (8 chunks, n <= 2 MiB), :310-318 (biPSIv2)
                    if (n > 1) {
                        if (k == 0 && (prim == 0 || prim > n)) good = false;
                        if (k > 0 && (prim == 0 || prim > n)) good = false;
                    }
                }
            }
            if (good && n > v.cap[b]) good = false;
            if (good) { h.n = n; h.hdr = headerSize; h.pIdx = p0; h.okFlag = 1; ok[b] = 1; newLen[b] = n; if (n >= 2) sum += n; }
        }
        hd[b] = h;
    }
    base[v.nBlocks] = sum;
    info->total = sum; info->nWords = (sum + 31) / 32; info->count = 0; info->dyn = 0;
    u32 longest = 0;
    for (int b = 0; b < v.nBlocks; b++) if (hd[b].okFlag && hd[b].n > longest) longest = hd[b].n;
    info->pad[2] = longest;                          // (decides the width of the link records: inv_narrow)
}

// Link records (round 6). A record is the next node, "the next node is a splitter" and the node's symbol. With positions relative to the
// block's first node a block of up to 8 MiB needs 23 + 1 + 8 bits: the records are 4-byte words then (half the bytes the stable counting
// sort scatters, half the working set of the walk), 8-byte words otherwise. The stage has no host read-back, so both forms of the two
// kernels are launched and the one the batch's longest block (InvInfo::pad[2]) does not ask for returns at once.
constexpr u32 INV_NARROW_MAX = 1u << 23;
__device__ __forceinline__ bool inv_narrow(const InvInfo* info) { return info->pad[2] <= INV_NARROW_MAX; }
template <class REC> struct InvRec;
template <> struct InvRec<u64> {
    static __device__ __forceinline__ u64 make(u32 nx, u32 bb, bool split, u32 sym) { (void)bb; return (u64)nx | (split ? 0x80000000ull : 0ull) | ((u64)sym << 32); }
    static __device__ __forceinline__ u32 next(u64 r, u32 bb) { (void)bb; return (u32)r & 0x7FFFFFFFu; }
    static __device__ __forceinline__ bool split(u64 r) { return (r >> 31) & 1; }
    static __device__ __forceinline__ u32 sym(u64 r) { return (u32)(r >> 32) & 0xFFu; }
};
template <> struct InvRec<u32> {
    static __device__ __forceinline__ u32 make(u32 nx, u32 bb, bool split, u32 sym) { return (nx - bb) | (split ? 0x800000u : 0u) | (sym << 24); }
    static __device__ __forceinline__ u32 next(u32 r, u32 bb) { return bb + (r & 0x7FFFFFu); }
    static __device__ __forceinline__ bool split(u32 r) { return (r >> 23) & 1u; }
    static __device__ __forceinline__ u32 sym(u32 r) { return r >> 24; }
};

// ---- list ranking with random splitters (Helman-JaJa style) ------------------------------------
// Node j = F-position (global slot). rec[j] = next node (31 bits) | "next is a splitter" (bit 31) | symbol << 32.
// The node holding BWT index 0 is the end of the text (self loop). Splitters: a pseudo-random 1/64 of the
// nodes + every block's chain head + the terminals. Every splitter walks to the next splitter ONCE and keeps the symbols
// it passes in a row of ROW bytes (a sub-list longer than a row continues in a fresh row taken from a counter: the walk
// never repeats a hop). The rows are ranked by pointer jumping over (successor row, length), then a copy kernel moves every
// row to its place in the text. A hop is a random 8-byte read, i.e. one 128-byte line from the fabric (55 G/s on MI355X
// whatever the size of the working set, tools/membench.hip): one O(n) random-access pass -- round 2 made two.
//
// The F-position of BWT index i is C[symbol] + (number of equal symbols before i): a stable counting sort, done per tile
// of 4096 symbols (histogram, scan over tiles and symbols, rank by ballot matching inside a wave) and written straight
// into rec -- no sorted copy of (symbol, index) pairs is ever materialised.
constexpr int SPLIT_LOG = 6;
constexpr u32 IT = 4096;             // symbols per tile: 4 waves x 16 rows of 64

// Splitters come in clusters of SPLIT_CLUSTER adjacent nodes (1 node in 64 all the same). Adjacent nodes are suffixes that are
// neighbours in sorted order: they share a prefix, so the chains that start in one cluster -- walked by adjacent lanes -- move
// through adjacent records for as many hops as that prefix lasts, and the lanes' reads fall into one or two 128-byte lines
// instead of eight. On data with long repeats most hops of the walk are of that kind.
#ifndef KNZ_SPLIT_CLUSTER_LOG
#define KNZ_SPLIT_CLUSTER_LOG 3
#endif
__device__ __forceinline__ bool hash_split(u32 j) { return (((j >> KNZ_SPLIT_CLUSTER_LOG) * 2654435761u) >> (32 - SPLIT_LOG)) == 0; }

__global__ __launch_bounds__(256) void k_bwt_i_hist(BwtView v, const BwtHdr* __restrict__ hd, int perTiles, u32* __restrict__ tileHist)
{
    const int b = blockIdx.y;
    const BwtHdr h = hd[b];
    if (!h.okFlag || h.n < 2) return;
    const u32 t0 = blockIdx.x * IT;
    if (t0 >= h.n) return;
    __shared__ u32 cnt[256];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    cnt[tid] = 0;
    __syncthreads();
    const u8* s = v.src[b] + h.hdr;
    // a row of 64 symbols that shows one symbol (a run of the BWT) costs one update, other rows go through LDS atomics
    u32 sy[16];
#pragma unroll
    for (int r = 0; r < 16; r++) { const u32 i = t0 + (u32)wave * 1024u + (u32)r * 64u + (u32)lane; sy[r] = (i < h.n) ? (u32)s[i] : 0xFFFFFFFFu; }
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const bool valid = sy[r] != 0xFFFFFFFFu;
        const u32 s0 = (u32)__builtin_amdgcn_readfirstlane((int)sy[r]);
        const unsigned long long va = __ballot(valid);
        if (__ballot(valid && sy[r] != s0) == 0) { if (lane == 0 && va) atomicAdd(&cnt[s0], (u32)__popcll(va)); }
        else if (valid) atomicAdd(&cnt[sy[r]], 1u);
    }
    __syncthreads();
    tileHist[((size_t)b * perTiles + blockIdx.x) * 256 + tid] = cnt[tid];
}

// tileHist[t][sym] := number of sym in the tiles before t, in two levels (no thread walks more than ~sqrt(tiles) tiles): inside
// segments of segT tiles first (segSum = the segment's totals), then over the segments; C[b][sym] = number of smaller symbols in the
// block; term[b] = node of BWT index 0 (first occurrence of its symbol). A tile's count is tileHist + segSum of its segment.
__host__ __device__ inline u32 bwt_i_seg_tiles(u32 perTiles) { u32 g = 1; while (g * g < perTiles) g <<= 1; return g; }

__global__ __launch_bounds__(256) void k_bwt_i_scan(const BwtHdr* __restrict__ hd, int perTiles, u32 segT, u32 nSeg, u32* __restrict__ tileHist,
                                                    u32* __restrict__ segSum)
{
    const int b = blockIdx.y;
    const u32 seg = blockIdx.x;
    const BwtHdr h = hd[b];
    const int tid = (int)threadIdx.x;
    u32 run = 0;
    if (h.okFlag && h.n >= 2) {
        const u32 nT = (h.n + IT - 1) / IT;
        const u32 t0 = seg * segT;
        const u32 t1 = (t0 + segT < nT) ? t0 + segT : nT;
        u32* p = tileHist + (size_t)b * perTiles * 256 + tid;
        for (u32 t = t0; t < t1; t++) { const u32 x = p[(size_t)t * 256]; p[(size_t)t * 256] = run; run += x; }
    }
    segSum[((size_t)b * nSeg + seg) * 256 + tid] = run;
}

__global__ __launch_bounds__(256) void k_bwt_i_scan2(BwtView v, const BwtHdr* __restrict__ hd, const u32* __restrict__ base, u32 nSeg,
                                                     u32* __restrict__ segSum, u32* __restrict__ Cb, u32* __restrict__ term)
{
    const int b = blockIdx.x;
    const BwtHdr h = hd[b];
    const int tid = (int)threadIdx.x;
    __shared__ u32 tot[256];
    u32 run = 0;
    u32* p = segSum + (size_t)b * nSeg * 256 + tid;
    for (u32 g = 0; g < nSeg; g++) { const u32 x = p[(size_t)g * 256]; p[(size_t)g * 256] = run; run += x; }
    tot[tid] = run;
    __syncthreads();
    if (tid == 0) { u32 acc = 0; for (int c = 0; c < 256; c++) { const u32 x = tot[c]; tot[c] = acc; acc += x; } }
    __syncthreads();
    Cb[b * 256 + tid] = tot[tid];
    if (tid == 0) term[b] = (h.okFlag && h.n >= 2) ? base[b] + tot[v.src[b][h.hdr]] : 0xFFFFFFFFu;
}

__device__ __forceinline__ bool is_splitter(u32 node, u32 head, u32 term) { return hash_split(node) || node == head || node == term; }

template <class REC>
__global__ __launch_bounds__(256) void k_bwt_i_links(BwtView v, const BwtHdr* __restrict__ hd, const u32* __restrict__ base, int perTiles,
                                                     const u32* __restrict__ tileHist, u32 segT, u32 nSeg, const u32* __restrict__ segSum,
                                                     const u32* __restrict__ Cb, const u32* __restrict__ term, REC* __restrict__ rec, const InvInfo* __restrict__ info)
{
    if (inv_narrow(info) != (sizeof(REC) == 4)) return;
    const int b = blockIdx.y;
    const BwtHdr h = hd[b];
    if (!h.okFlag || h.n < 2) return;
    const u32 t0 = blockIdx.x * IT;
    if (t0 >= h.n) return;
    __shared__ u32 cntw[4][256];
    __shared__ u32 start[256];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int w = 0; w < 4; w++) cntw[w][tid] = 0;
    start[tid] = base[b] + Cb[b * 256 + tid] + segSum[((size_t)b * nSeg + blockIdx.x / segT) * 256 + tid]
                 + tileHist[((size_t)b * perTiles + blockIdx.x) * 256 + tid];
    __syncthreads();
    const u8* s = v.src[b] + h.hdr;
    u32 sym[16], pre[16];
    // rank among the equal symbols of my wave's 1024-symbol stretch, rows in order
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32 i = t0 + (u32)wave * 1024u + (u32)r * 64u + (u32)lane;
        const bool valid = i < h.n;
        sym[r] = valid ? s[i] : 0u;
        // (prims::rs_row_rank: a row that shows one symbol -- most rows behind a BWT -- costs two ballots, the others the cheap form of the
        // eight-ballot match)
        pre[r] = prims::rs_row_rank(valid, sym[r], cntw[wave], lane, ltMask);
    }
    __syncthreads();
    // symbols of earlier waves of the tile come first
    {
        const u32 c0 = cntw[0][tid], c1 = cntw[1][tid], c2 = cntw[2][tid];
        const u32 st = start[tid];
        cntw[0][tid] = st; cntw[1][tid] = st + c0; cntw[2][tid] = st + c0 + c1; cntw[3][tid] = st + c0 + c1 + c2;
    }
    __syncthreads();
    const u32 bb = base[b], pIdx = h.pIdx;
    const u32 head = bb + pIdx - 1, tm = term[b];
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const u32 i = t0 + (u32)wave * 1024u + (u32)r * 64u + (u32)lane;
        if (i >= h.n) continue;
        const u32 j = cntw[wave][sym[r]] + pre[r];
        u32 nx;
        if (i == 0) nx = j;                                       // end of text
        else nx = bb + ((i < pIdx) ? i - 1 : i);                  // BWT.cpp:203-215
        rec[j] = InvRec<REC>::make(nx, bb, is_splitter(nx, head, tm), sym[r]);
    }
}

// splitter flags from the node number alone, as a bit map: a thread makes the word of 32 nodes and its population count
__global__ __launch_bounds__(256) void k_bwt_i_flags(const BwtHdr* __restrict__ hd, const u32* __restrict__ base, const u32* __restrict__ term, int nBlocks,
                                                     const InvInfo* __restrict__ info, u32* __restrict__ bits, u32* __restrict__ wcount, u32* __restrict__ wblk)
{
    const u32 w = blockIdx.x * 256 + threadIdx.x;
    const u32 j0 = 32u * w;
    const u32 total = info->total;
    if (j0 >= total) return;
    int b = find_block(base, nBlocks, j0);
    wblk[w] = (u32)b;                                            // block of the word's first node (the walk starts its search there)
    u32 be = base[b + 1], head = base[b] + hd[b].pIdx - 1, tm = term[b];
    u32 word = 0;
    for (u32 k = 0; k < 32; k++) {
        const u32 j = j0 + k;
        if (j >= total) break;
        if (j >= be) { do { b++; } while (j >= base[b + 1]); be = base[b + 1]; head = base[b] + hd[b].pIdx - 1; tm = term[b]; }
        word |= (is_splitter(j, head, tm) ? 1u : 0u) << k;
    }
    bits[w] = word;
    wcount[w] = (u32)__popc(word);
}

// the set bits of every word, in order: splitNode[rank] = node
__global__ __launch_bounds__(256) void k_bwt_i_compact(const u32* __restrict__ bits, const u32* __restrict__ wprefix, const InvInfo* __restrict__ info, u32 maxRows,
                                                       u32* __restrict__ splitNode)
{
    const u32 w = blockIdx.x * 256 + threadIdx.x;
    if (w >= info->nWords) return;
    u32 word = bits[w];
    u32 at = wprefix[w];
    while (word) {
        const u32 k = (u32)__ffs((int)word) - 1;
        if (at < maxRows) splitNode[at] = 32u * w + k;
        at++;
        word &= word - 1;
    }
}

// rank of a splitter node among the splitters
__device__ __forceinline__ u32 split_rank(const u32* __restrict__ bits, const u32* __restrict__ wprefix, u32 node)
{
    return wprefix[node >> 5] + (u32)__popc(bits[node >> 5] & ((1u << (node & 31)) - 1u));
}

constexpr u32 ROW = 128;            // bytes of one row of symbols (sub-lists have 64 nodes on average, 13 % are longer than a row)

// the walk: sub-list length, successor row and the symbols of the sub-list, one thread per splitter
template <class REC>
__global__ __launch_bounds__(256) void k_bwt_i_walk(const REC* __restrict__ rec, const u32* __restrict__ rowNode, const u32* __restrict__ bits,
                                                    const u32* __restrict__ wprefix, InvInfo* __restrict__ info, u32 maxRows, u32* __restrict__ succ,
                                                    u32* __restrict__ dist, u8* __restrict__ rowLen, u8* __restrict__ rows, const u32* __restrict__ base,
                                                    int nBlocks, u32* __restrict__ rowBlk, const u32* __restrict__ wblk)
{
    if (inv_narrow(info) != (sizeof(REC) == 4)) return;
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    const u32 count = info->count;
    if (c >= count || c >= maxRows) return;
    u32 node = rowNode[c];
    REC r = rec[node];
    // a chain never leaves its block: the block of every row this thread fills, looked up once (the copy kernel needs it per row)
    u32 blk = wblk[node >> 5];
    while (node >= base[blk + 1]) blk++;
    (void)nBlocks;
    rowBlk[c] = blk;
    const u32 bb = base[blk];
    if (InvRec<REC>::next(r, bb) == node) {                      // terminal: the last byte of the text
        succ[c] = c; dist[c] = 0; rowLen[c] = 1; rows[(size_t)c * ROW] = (u8)InvRec<REC>::sym(r);
        return;
    }
    u32 cur = c;
    for (;;) {
        u64* row = reinterpret_cast<u64*>(rows + (size_t)cur * ROW);
        u32 len = 0, nx = 0;
        u64 acc = 0;
        bool done = false;
        for (;;) {
            acc |= (u64)InvRec<REC>::sym(r) << (8 * (len & 7));
            len++;
            if ((len & 7) == 0) { row[(len >> 3) - 1] = acc; acc = 0; }
            nx = InvRec<REC>::next(r, bb);
            if (InvRec<REC>::split(r)) { done = true; break; }
            if (len == ROW) break;
            node = nx;
            r = rec[node];
        }
        if (len & 7) row[len >> 3] = acc;
        dist[cur] = len;
        rowLen[cur] = (u8)len;
        if (done) { succ[cur] = split_rank(bits, wprefix, nx); break; }
        // the sub-list goes on: a fresh row (only malformed input -- a cycle without a splitter -- can exhaust them)
        const u32 id = count + atomicAdd(&info->dyn, 1u);
        if (id >= maxRows) { succ[cur] = cur; break; }
        succ[cur] = id;
        rowBlk[id] = blk;
        cur = id;
        node = nx;
        r = rec[node];
    }
}

__global__ __launch_bounds__(256) void k_bwt_i_jump(const u32* __restrict__ nextIn, const u32* __restrict__ distIn, const InvInfo* __restrict__ info, u32 maxRows,
                                                    u32* __restrict__ nextOut, u32* __restrict__ distOut)
{
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    u32 rowsTotal = info->count + info->dyn;
    if (rowsTotal > maxRows) rowsTotal = maxRows;
    if (j >= rowsTotal) return;
    const u32 nx = nextIn[j];
    distOut[j] = distIn[j] + distIn[nx];
    nextOut[j] = nextIn[nx];
}

// ---- ranking the rows through rulers (round 5) ----------------------------------------------------------------------------------
// Pointer jumping over all rows is 17-18 rounds of two random gathers per row: 119 M gathers for the 3.3 M rows of the headline
// workload, 0.9 ms. One row in 8 (by a hash of its number, rlog = 3; chain heads and terminals on top) is a RULER: a ruler walks to the next
// one, ONCE, and tells every row it passes which ruler owns it and how far behind that ruler it lies; only the rulers (1/8 of the rows,
// in dense arrays) are ranked by pointer jumping; a row's distance to the end is its ruler's minus its offset. 3.3 M walked hops
// instead of 119 M gathers. Damaged input (a cycle without a ruler, a row that points at itself) ends a walk after RULER_LIMIT steps
// or at the self-loop: such rows get distances without meaning, never an access outside a buffer -- as with pointer jumping. A walk
// that runs out of steps marks its BLOCK as failed (the stage's ok flag -> ERR_PROCESS_BLOCK, what the reference reports for a block
// its inverse rejects): rulers are one row in eight by a hash of the row number (rlog = 3), so RULER_LIMIT rows in a row without one do
// not happen by chance (7/8 to the 65,536th), but a block built on purpose to lead its chain through such rows must be refused, not
// decoded to wrong bytes. KNZ_BWT_I_JUMP_ALL=1 (pointer jumping over all rows, rounds 3-4) decodes such a block.
constexpr u32 RULER_LIMIT = 1u << 16;
__device__ __forceinline__ bool ruler_hash(u32 c, u32 rlog) { return ((c * 2654435761u) >> (32 - rlog)) == 0; }
__device__ __forceinline__ bool ruler_bit(const u32* __restrict__ rbits, u32 c) { return (rbits[c >> 5] >> (c & 31)) & 1u; }

__global__ __launch_bounds__(256) void k_bwt_i_ruler_flags(const BwtHdr* __restrict__ hd, const u32* __restrict__ base, const u32* __restrict__ rowNode,
                                                           const u32* __restrict__ rowBlk, const u32* __restrict__ succ, const InvInfo* __restrict__ info,
                                                           u32 maxRows, u32* __restrict__ rbits, u32* __restrict__ rcount, u32 rlog)
{
    const u32 w = blockIdx.x * 256 + threadIdx.x;
    u32 rowsTotal = info->count + info->dyn;
    if (rowsTotal > maxRows) rowsTotal = maxRows;
    if (32u * w >= ((maxRows + 31u) & ~31u)) return;
    u32 count = info->count;
    if (count > maxRows) count = maxRows;
    u32 word = 0;
    for (u32 k = 0; k < 32; k++) {
        const u32 c = 32u * w + k;
        if (c >= count) break;                                   // (rows made on the way -- continuations of long sub-lists -- are never rulers)
        bool r = ruler_hash(c, rlog) || succ[c] == c;
        if (!r) { const u32 b = rowBlk[c]; r = rowNode[c] == base[b] + hd[b].pIdx - 1; }     // the first row of the block's chain
        word |= (r ? 1u : 0u) << k;
    }
    rbits[w] = word;
    rcount[w] = (u32)__popc(word);
}

__device__ __forceinline__ u32 ruler_rank(const u32* __restrict__ rbits, const u32* __restrict__ rprefix, u32 c)
{
    return rprefix[c >> 5] + (u32)__popc(rbits[c >> 5] & ((1u << (c & 31)) - 1u));
}

__global__ __launch_bounds__(256) void k_bwt_i_ruler_walk(const u32* __restrict__ succ, const u32* __restrict__ dist, const u32* __restrict__ rbits,
                                                          const u32* __restrict__ rprefix, const InvInfo* __restrict__ info, u32 maxRows, u32 maxRulers,
                                                          u32* __restrict__ owner, u32* __restrict__ pre, u32* __restrict__ rS, u32* __restrict__ rD,
                                                          const u32* __restrict__ rowBlk, u8* __restrict__ ok)
{
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    u32 count = info->count;
    if (count > maxRows) count = maxRows;
    if (c >= count || !ruler_bit(rbits, c)) return;
    u32 rowsTotal = info->count + info->dyn;
    if (rowsTotal > maxRows) rowsTotal = maxRows;
    const u32 idx = ruler_rank(rbits, rprefix, c);
    if (idx >= maxRulers) return;
    u32 x = succ[c];
    u32 acc = dist[c];
    if (x == c) { rS[idx] = idx; rD[idx] = acc; return; }       // terminal (dist 0)
    u32 to = idx;                                                // where the walk ends when it does not meet a ruler
    u32 steps = 0;
    for (; steps < RULER_LIMIT; steps++) {
        if (x >= rowsTotal) break;                               // (damaged input)
        if (x < count && ruler_bit(rbits, x)) { const u32 r = ruler_rank(rbits, rprefix, x); to = r < maxRulers ? r : idx; break; }
        owner[x] = idx;
        pre[x] = acc;
        acc += dist[x];
        const u32 nx = succ[x];
        if (nx == x) break;
        x = nx;
    }
    if (steps == RULER_LIMIT) ok[rowBlk[c]] = 0;                 // out of steps: the block is not decoded (see above)
    rS[idx] = to;
    rD[idx] = (to == idx) ? 0u : acc;                            // a walk that found no ruler ends the chain there
}

__global__ __launch_bounds__(256) void k_bwt_i_ruler_jump(const u32* __restrict__ nextIn, const u32* __restrict__ distIn, const u32* __restrict__ nPtr, u32 cap,
                                                          u32* __restrict__ nextOut, u32* __restrict__ distOut)
{
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    u32 n = *nPtr;
    if (n > cap) n = cap;
    if (j >= n) return;
    const u32 nx = nextIn[j];
    distOut[j] = distIn[j] + distIn[nx];
    nextOut[j] = nextIn[nx];
}

__global__ __launch_bounds__(256) void k_bwt_i_ruler_finish(const u32* __restrict__ rbits, const u32* __restrict__ rprefix, const u32* __restrict__ owner,
                                                            const u32* __restrict__ pre, const u32* __restrict__ rD, const InvInfo* __restrict__ info,
                                                            u32 maxRows, u32 maxRulers, u32* __restrict__ dEnd)
{
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    u32 rowsTotal = info->count + info->dyn;
    if (rowsTotal > maxRows) rowsTotal = maxRows;
    if (c >= rowsTotal) return;
    u32 count = info->count;
    if (count > maxRows) count = maxRows;
    u32 d = 0xFFFFFFFFu;                                         // (a row no walk came by: damaged input; the copy kernel skips it)
    if (c < count && ruler_bit(rbits, c)) { const u32 r = ruler_rank(rbits, rprefix, c); if (r < maxRulers) d = rD[r]; }
    else { const u32 o = owner[c]; if (o < maxRulers) { const u32 t = rD[o], p = pre[c]; d = t >= p ? t - p : 0xFFFFFFFFu; } }
    dEnd[c] = d;
}

// every row to its place: 32 lanes per row, a lane makes one aligned dword of the output from two aligned dwords of the row.
// Round 5: a half-wave takes FOUR rows at a time and keeps going (grid-stride) -- with one row per half-wave the kernel was three
// dependent loads and one store per wave, 750 K workgroups of which half had nothing to do: 0.95 ms for a copy of 212 MB. The loads of
// the four rows are issued side by side (what a row needs first -- block, end, length -- then the block's size and buffer, then the two
// dwords), so a wave has twelve loads in flight instead of three.
constexpr int PLACE_U = 4;

__global__ __launch_bounds__(256) void k_bwt_i_place(BwtView v, const BwtHdr* __restrict__ hd, const u32* __restrict__ rowBlk,
                                                     const u32* __restrict__ dEnd, const u8* __restrict__ rowLen, const u8* __restrict__ rows,
                                                     const InvInfo* __restrict__ info, u32 maxRows)
{
    const u32 half = blockIdx.x * 8 + (threadIdx.x >> 5);
    const u32 nHalf = gridDim.x * 8;
    const u32 l = threadIdx.x & 31;
    u32 rowsTotal = info->count + info->dyn;
    if (rowsTotal > maxRows) rowsTotal = maxRows;
    for (u32 c0 = half; c0 < rowsTotal; c0 += PLACE_U * nHalf) {
        bool ok[PLACE_U];
        u32 cc[PLACE_U], dd[PLACE_U], ll[PLACE_U], nn[PLACE_U];
        int bb[PLACE_U];
        u8* dp[PLACE_U];
#pragma unroll
        for (int t = 0; t < PLACE_U; t++) {
            const u32 c = c0 + (u32)t * nHalf;
            ok[t] = c < rowsTotal;
            cc[t] = ok[t] ? c : c0;
            bb[t] = (int)rowBlk[cc[t]]; dd[t] = dEnd[cc[t]]; ll[t] = rowLen[cc[t]];
        }
#pragma unroll
        for (int t = 0; t < PLACE_U; t++) { nn[t] = hd[bb[t]].n; dp[t] = v.dst[bb[t]]; }
        u32 aa[PLACE_U], lo[PLACE_U], hi[PLACE_U];
        int oo[PLACE_U], lead[PLACE_U];
        uintptr_t W0[PLACE_U];
#pragma unroll
        for (int t = 0; t < PLACE_U; t++) {
            if (dd[t] >= nn[t]) ok[t] = false;                       // (malformed input)
            aa[t] = ok[t] ? nn[t] - 1 - dd[t] : 0u;                  // text position of the row's first symbol
            if (ll[t] > nn[t] - aa[t]) ll[t] = nn[t] - aa[t];
            if (!ok[t]) ll[t] = 0;
            const uintptr_t A = reinterpret_cast<uintptr_t>(dp[t]) + aa[t];
            W0[t] = A & ~(uintptr_t)3;
            lead[t] = (int)(A - W0[t]);                              // bytes of the first dword that lie in front of the row
            // dword k of the output range covers row bytes [4k - lead, 4k - lead + 4); this lane: k = l
            oo[t] = 4 * (int)l - lead[t];
            const u32* row32 = reinterpret_cast<const u32*>(rows + (size_t)cc[t] * ROW);
            const u32 q = oo[t] > 0 ? (u32)oo[t] >> 2 : 0u;
            lo[t] = row32[q < ROW / 4 ? q : ROW / 4 - 1];
            hi[t] = row32[q + 1 < ROW / 4 ? q + 1 : ROW / 4 - 1];
        }
#pragma unroll
        for (int t = 0; t < PLACE_U; t++) {
            const u32 len = ll[t];
            const u8* rb = rows + (size_t)cc[t] * ROW;
            u8* dst = dp[t];
            if (4 * l < (u32)lead[t] + len) {
                const int o = oo[t];
                if (o >= 0 && (u32)o + 4 <= len) {
                    const u32 sh = ((u32)o & 3) * 8;
                    const u32 h2 = sh ? hi[t] : 0u;
                    *reinterpret_cast<u32*>(W0[t] + 4 * (uintptr_t)l) = (u32)(((((u64)h2) << 32) | lo[t]) >> sh);
                } else {
                    for (int j = 0; j < 4; j++) { const int ro = o + j; if (ro >= 0 && (u32)ro < len) dst[aa[t] + (u32)ro] = rb[ro]; }
                }
            }
            // a row that starts in the middle of a dword reaches into a 33rd one: its last (at most three) bytes
            if (l == 0 && (u32)lead[t] + len > 128u) {
                const int o = 128 - lead[t];
                for (int j = 0; j < 4; j++) { const int ro = o + j; if ((u32)ro < len) dst[aa[t] + (u32)ro] = rb[ro]; }
            }
        }
    }
}

__global__ void k_bwt_i_tiny(BwtView v, const BwtHdr* __restrict__ hd)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= v.nBlocks) return;
    const BwtHdr h = hd[b];
    if (h.okFlag && h.n == 1) v.dst[b][0] = v.src[b][h.hdr];     // BWT.cpp:152-158
}

// scratch layout, shared by the size query and the launch
struct InvScratch {
    u32* tileHist; u32* segSum; u32* Cb; u32* term; u64* rec; u32* bits; u32* wcount; u32* wprefix; u32* wblk;
    u32* rowNode; u32* rowBlk; u32* nA; u32* nB; u32* dA; u32* dB; u8* rowLen; u8* rows;
    BwtHdr* hd; u32* base; InvInfo* info; void* scanTmp;
    u32* rS[2]; u32* rD[2];      // rulers, dense: successor ruler and distance to it (two copies for the pointer jumping)
    u32* rbits; u32* rprefix;    // which rows are rulers (bit map over the rows), rulers in the words before
    u32 maxRows, maxRulers; int perTiles; u32 segT, nSeg;
};

static size_t inv_carve(u8* p, int nBlocks, u32 VS, size_t maxTotal, InvScratch* w)
{
    u8* q = p;
    auto take = [&](size_t sz) { u8* r = q; q += align256(sz); return r; };
    // splitters: a multiplicative hash of consecutive node numbers picks 1 in 64 with low discrepancy; heads and terminals on top
    const size_t maxRows = maxTotal / 48 + maxTotal / ROW + 4096 + 3 * (size_t)nBlocks;
    w->maxRows = (u32)maxRows;
    w->perTiles = (int)(((size_t)VS + IT - 1) / IT);
    w->segT = bwt_i_seg_tiles((u32)w->perTiles);
    w->nSeg = ((u32)w->perTiles + w->segT - 1) / w->segT;
    const size_t nWordsMax = maxTotal / 32 + 2;
    w->tileHist = (u32*)take(4ull * 256 * (size_t)w->perTiles * nBlocks);
    w->segSum = (u32*)take(4ull * 256 * (size_t)w->nSeg * nBlocks);
    w->Cb = (u32*)take(4ull * 256 * nBlocks);
    w->term = (u32*)take(4ull * (nBlocks + 1));
    w->rec = (u64*)take(8 * maxTotal);
    w->bits = (u32*)take(4 * nWordsMax); w->wcount = (u32*)take(4 * nWordsMax); w->wprefix = (u32*)take(4 * nWordsMax); w->wblk = (u32*)take(4 * nWordsMax);
    w->rowNode = (u32*)take(4 * maxRows);
    w->rowBlk = (u32*)take(4 * maxRows);
    w->nA = (u32*)take(4 * maxRows); w->nB = (u32*)take(4 * maxRows);
    w->dA = (u32*)take(4 * maxRows); w->dB = (u32*)take(4 * maxRows);
    w->rowLen = (u8*)take(maxRows);
    w->rows = (u8*)take(maxRows * (size_t)ROW + 64);
    w->hd = (BwtHdr*)take(sizeof(BwtHdr) * (size_t)nBlocks);
    w->base = (u32*)take(4ull * (nBlocks + 2));
    w->info = (InvInfo*)take(sizeof(InvInfo));
    w->scanTmp = take(prims::scan_tmp_bytes(std::max(nWordsMax, maxRows / 32 + 2)));
    // one row in 32 is a ruler, two more per block; four times that is room no input reaches (the hash picks by row NUMBER)
    const size_t maxRulers = maxRows / 2 + 64 + 2 * (size_t)nBlocks;
    w->maxRulers = (u32)maxRulers;
    for (int k = 0; k < 2; k++) { w->rS[k] = (u32*)take(4 * maxRulers); w->rD[k] = (u32*)take(4 * maxRulers); }
    w->rbits = (u32*)take(4 * (maxRows / 32 + 2)); w->rprefix = (u32*)take(4 * (maxRows / 32 + 2));
    return (size_t)(q - p);
}

size_t bwt_inverse_scratch_bytes(int nBlocks, u32 VS, size_t total)
{
    InvScratch w;
    return inv_carve(nullptr, nBlocks, VS, total, &w) + 4096;
}

int launch_bwt_inverse(hipStream_t s, const XfStage& st, void* scratch, size_t scratchBytes, u32* h_pinned)
{
    (void)h_pinned;
    BwtView v; v.src = st.src; v.dst = st.dst; v.len = st.len; v.cap = st.cap; v.VS = st.maxLen; v.nBlocks = st.nBlocks;
    const size_t maxTotal = (size_t)st.nBlocks * v.VS;
    if (maxTotal >= (1ull << 31)) return -2;                 // node ids share a word with the splitter flag
    InvScratch w;
    if (inv_carve(reinterpret_cast<u8*>(scratch), st.nBlocks, v.VS, maxTotal, &w) > scratchBytes) return -2;
    const int perTiles = w.perTiles;
    const u32 segT = w.segT, nSeg = w.nSeg;
    // no size is read back: every grid is sized by the upper bound (blocks x longest block), the kernels take the sizes from `info`
    {
        KScope ks_("k_bwt_i_header");
        if (st.bsVersion < 6) hipLaunchKernelGGL(k_bwt_i_header<true>, dim3(1), dim3(64), 0, s, v, w.hd, w.base, st.ok, st.newLen, w.info);
        else hipLaunchKernelGGL(k_bwt_i_header<false>, dim3(1), dim3(64), 0, s, v, w.hd, w.base, st.ok, st.newLen, w.info);
    }
    { KScope ks_("k_bwt_i_tiny"); hipLaunchKernelGGL(k_bwt_i_tiny, dim3((st.nBlocks + 63) / 64), dim3(64), 0, s, v, w.hd); }
    const dim3 gridT((unsigned)perTiles, st.nBlocks);
    { KScope ks_("k_bwt_i_hist"); hipLaunchKernelGGL(k_bwt_i_hist, gridT, dim3(256), 0, s, v, w.hd, perTiles, w.tileHist); }
    { KScope ks_("k_bwt_i_scan"); hipLaunchKernelGGL(k_bwt_i_scan, dim3(nSeg, st.nBlocks), dim3(256), 0, s, w.hd, perTiles, segT, nSeg, w.tileHist, w.segSum);
      hipLaunchKernelGGL(k_bwt_i_scan2, dim3(st.nBlocks), dim3(256), 0, s, v, w.hd, w.base, nSeg, w.segSum, w.Cb, w.term); }
    { KScope ks_("k_bwt_i_links");
      hipLaunchKernelGGL(k_bwt_i_links<u32>, gridT, dim3(256), 0, s, v, w.hd, w.base, perTiles, w.tileHist, segT, nSeg, w.segSum, w.Cb, w.term, reinterpret_cast<u32*>(w.rec), w.info);
      hipLaunchKernelGGL(k_bwt_i_links<u64>, gridT, dim3(256), 0, s, v, w.hd, w.base, perTiles, w.tileHist, segT, nSeg, w.segSum, w.Cb, w.term, w.rec, w.info); }
    const u32 nWordsMax = (u32)(maxTotal / 32 + 1);
    { KScope ks_("k_bwt_i_flags"); hipLaunchKernelGGL(k_bwt_i_flags, GRID1(nWordsMax), w.hd, w.base, w.term, st.nBlocks, w.info, w.bits, w.wcount, w.wblk); }
    { KScope ks_("k_bwt_i_scan_words"); prims::launch_scan<prims::SCAN_SUM_EXCL>(s, w.wcount, w.wprefix, nWordsMax, &w.info->nWords, w.scanTmp, &w.info->count); }
    { KScope ks_("k_bwt_i_compact"); hipLaunchKernelGGL(k_bwt_i_compact, GRID1(nWordsMax), w.bits, w.wprefix, w.info, w.maxRows, w.rowNode); }
    const u32 maxCount = (u32)(maxTotal / 48 + 4096 + 3 * (size_t)st.nBlocks);
    { KScope ks_("k_bwt_i_walk");
      hipLaunchKernelGGL(k_bwt_i_walk<u32>, GRID1(maxCount), reinterpret_cast<const u32*>(w.rec), w.rowNode, w.bits, w.wprefix, w.info, w.maxRows, w.nA, w.dA, w.rowLen, w.rows, w.base, st.nBlocks, w.rowBlk, w.wblk);
      hipLaunchKernelGGL(k_bwt_i_walk<u64>, GRID1(maxCount), w.rec, w.rowNode, w.bits, w.wprefix, w.info, w.maxRows, w.nA, w.dA, w.rowLen, w.rows, w.base, st.nBlocks, w.rowBlk, w.wblk); }
    u32* nA = w.nA; u32* nB = w.nB; u32* dA = w.dA; u32* dB = w.dB;
    // chains never leave a block: a chain has at most rows-per-block rows
    const u64 chainRows = (u64)v.VS / 48 + (u64)v.VS / ROW + 4096 + 3;
    static const bool allRowsJump = getenv("KNZ_BWT_I_JUMP_ALL") != nullptr && atoi(getenv("KNZ_BWT_I_JUMP_ALL")) != 0;   // the ranking of rounds 3-4
    if (allRowsJump) {
        for (u64 span = 1; span < chainRows; span <<= 1) {
            { KScope ks_("k_bwt_i_jump"); hipLaunchKernelGGL(k_bwt_i_jump, GRID1(w.maxRows), nA, dA, w.info, w.maxRows, nB, dB); }
            std::swap(nA, nB); std::swap(dA, dB);
        }
    } else {
        // rulers (see k_bwt_i_ruler_walk): nB / dB hold every row's owner and offset, the rows' distances to the end come out where their
        // successors were
        KScope ks_("k_bwt_i_jump");
        u32* rbits = w.rbits; u32* rprefix = w.rprefix;
        const u32 rWords = (w.maxRows + 31) / 32;
        u32* nRulers = &w.info->pad[0];
        static const u32 rlog = []() { const char* e = getenv("KNZ_BWT_I_RULER_LOG"); const int v = e ? atoi(e) : 0; return (v >= 2 && v <= 8) ? (u32)v : 3u; }();
        hipLaunchKernelGGL(k_bwt_i_ruler_flags, GRID1(rWords), w.hd, w.base, w.rowNode, w.rowBlk, nA, w.info, w.maxRows, rbits, rprefix, rlog);
        prims::launch_scan<prims::SCAN_SUM_EXCL>(s, rprefix, rprefix, rWords, nullptr, w.scanTmp, nRulers);
        hipMemsetAsync(nB, 0xFF, 4 * (size_t)w.maxRows, s);                      // owner: none
        hipLaunchKernelGGL(k_bwt_i_ruler_walk, GRID1(w.maxRows), nA, dA, rbits, rprefix, w.info, w.maxRows, w.maxRulers, nB, dB, w.rS[0], w.rD[0], w.rowBlk, st.ok);
        int cur = 0;
        const u64 chainRulers = (chainRows >> (rlog - 1)) + 64;                     // (twice the expected rulers of the longest chain)
        for (u64 span = 1; span < chainRulers; span <<= 1) {
            hipLaunchKernelGGL(k_bwt_i_ruler_jump, GRID1(w.maxRulers), w.rS[cur], w.rD[cur], nRulers, w.maxRulers, w.rS[cur ^ 1], w.rD[cur ^ 1]);
            cur ^= 1;
        }
        hipLaunchKernelGGL(k_bwt_i_ruler_finish, GRID1(w.maxRows), rbits, rprefix, nB, dB, w.rD[cur], w.info, w.maxRows, w.maxRulers, nA);
        dA = nA;                                                                    // (the successor array is not needed any more: distances to the end)
    }
    { KScope ks_("k_bwt_i_place");
      const u32 wantWg = (w.maxRows + 8 * PLACE_U - 1) / (8 * PLACE_U);
      hipLaunchKernelGGL(k_bwt_i_place, dim3(std::max(1u, std::min(wantWg, 16384u))), dim3(256), 0, s, v, w.hd, w.rowBlk, dA, w.rowLen, w.rows, w.info, w.maxRows); }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace knz
