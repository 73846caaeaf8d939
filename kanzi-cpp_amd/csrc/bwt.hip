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
#include <rocprim/rocprim.hpp>

namespace knz {

// compaction: act[out] = value[a] for kept elements, given the exclusive scan of keep
__global__ __launch_bounds__(256) void k_compact(const u32* __restrict__ keep, const u32* __restrict__ scan, const u32* __restrict__ value, u32 n,
                                                 u32* __restrict__ out)
{
    const u32 a = blockIdx.x * 256 + threadIdx.x;
    if (a >= n) return;
    if (keep[a]) out[scan[a]] = value ? value[a] : a;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, s

// ------------------------------------------------------------------------------------------------
// inverse
// ------------------------------------------------------------------------------------------------
struct BwtHdr { u32 n; u32 hdr; u32 pIdx; u32 okFlag; };

__global__ void k_bwt_i_header(BwtView v, BwtHdr* __restrict__ hd, u32* __restrict__ base, u8* __restrict__ ok, u32* __restrict__ newLen)
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
        if (blockSize >= 2) {
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
}

// ---- list ranking with random splitters (Helman-JaJa style) ------------------------------------
// Node j = F-position (global slot). rec[j] = next node (31 bits) | "next is a splitter" (bit 31) | symbol << 32.
// The node holding BWT index 0 is the end of the text (self loop). Splitters: a pseudo-random 1/64 of the
// nodes + every block's chain head + the terminals. Each splitter walks to the next splitter (sub-list
// length), the short splitter list is ranked by pointer jumping, then each splitter walks its sub-list
// again and writes the text. Two O(n) random-access passes instead of log2(n) of them.
//
// The F-position of BWT index i is C[symbol] + (number of equal symbols before i): a stable counting sort, done per tile
// of 4096 symbols (histogram, scan over tiles and symbols, rank by ballot matching inside a wave) and written straight
// into rec -- no sorted copy of (symbol, index) pairs is ever materialised.
constexpr int SPLIT_LOG = 6;
constexpr u32 IT = 4096;             // symbols per tile: 4 waves x 16 rows of 64

__device__ __forceinline__ bool hash_split(u32 j) { return ((j * 2654435761u) >> (32 - SPLIT_LOG)) == 0; }

// lanes of the wave whose (valid) symbol equals mine
__device__ __forceinline__ unsigned long long sym_peers(bool valid, u32 sym)
{
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
        const bool one = (sym >> bit) & 1u;
        const unsigned long long bal = __ballot(valid && one);
        peers &= one ? bal : ~bal;
    }
    return peers;
}

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
#pragma unroll 4
    for (int r = 0; r < 16; r++) {
        const u32 i = t0 + (u32)wave * 1024u + (u32)r * 64u + (u32)lane;
        const bool valid = i < h.n;
        const u32 sym = valid ? s[i] : 0u;
        const unsigned long long peers = sym_peers(valid, sym);
        if (valid && lane == __ffsll((long long)peers) - 1) atomicAdd(&cnt[sym], (u32)__popcll(peers));
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

__global__ __launch_bounds__(256) void k_bwt_i_links(BwtView v, const BwtHdr* __restrict__ hd, const u32* __restrict__ base, int perTiles,
                                                     const u32* __restrict__ tileHist, u32 segT, u32 nSeg, const u32* __restrict__ segSum,
                                                     const u32* __restrict__ Cb, const u32* __restrict__ term, u64* __restrict__ rec)
{
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
        const unsigned long long peers = sym_peers(valid, sym[r]);
        const int leader = __ffsll((long long)peers) - 1;
        u32 old = 0;
        if (valid && lane == leader) { old = cntw[wave][sym[r]]; cntw[wave][sym[r]] = old + (u32)__popcll(peers); }
        old = (u32)__shfl((int)old, leader < 0 ? 0 : leader, 64);
        pre[r] = old + (u32)__popcll(peers & ltMask);
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
        rec[j] = (u64)nx | (is_splitter(nx, head, tm) ? 0x80000000ull : 0ull) | ((u64)sym[r] << 32);
    }
}

// splitter flags from the node number alone, as a bit map: a thread makes the word of 32 nodes and its population count
__global__ __launch_bounds__(256) void k_bwt_i_flags(const BwtHdr* __restrict__ hd, const u32* __restrict__ base, const u32* __restrict__ term, int nBlocks,
                                                     u32 total, u32* __restrict__ bits, u32* __restrict__ wcount)
{
    const u32 w = blockIdx.x * 256 + threadIdx.x;
    const u32 j0 = 32u * w;
    if (j0 >= total) return;
    int b = find_block(base, nBlocks, j0);
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
__global__ __launch_bounds__(256) void k_bwt_i_compact(const u32* __restrict__ bits, const u32* __restrict__ wprefix, u32 nWords, u32* __restrict__ splitNode)
{
    const u32 w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nWords) return;
    u32 word = bits[w];
    u32 at = wprefix[w];
    while (word) {
        const u32 k = (u32)__ffs((int)word) - 1;
        splitNode[at++] = 32u * w + k;
        word &= word - 1;
    }
}

// rank of a splitter node among the splitters
__device__ __forceinline__ u32 split_rank(const u32* __restrict__ bits, const u32* __restrict__ wprefix, u32 node)
{
    return wprefix[node >> 5] + (u32)__popc(bits[node >> 5] & ((1u << (node & 31)) - 1u));
}

// walk 1: sub-list length and successor splitter (compact indices)
__global__ __launch_bounds__(256) void k_bwt_i_walk1(const u64* __restrict__ rec, const u32* __restrict__ splitNode, const u32* __restrict__ bits,
                                                     const u32* __restrict__ wprefix, u32 count, u32 limit, u32* __restrict__ succ, u32* __restrict__ dist)
{
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    if (c >= count) return;
    u32 node = splitNode[c];
    u64 r = rec[node];
    if (((u32)r & 0x7FFFFFFFu) == node) { succ[c] = c; dist[c] = 0; return; }    // terminal
    u32 len = 0;
    while (true) {
        len++;
        const u32 nx = (u32)r & 0x7FFFFFFFu;
        if ((r >> 31) & 1) { succ[c] = split_rank(bits, wprefix, nx); break; }
        if (len > limit) { succ[c] = c; break; }              // malformed input (cycle without splitter)
        node = nx;
        r = rec[node];
    }
    dist[c] = len;
}

__global__ __launch_bounds__(256) void k_bwt_i_jump(const u32* __restrict__ nextIn, const u32* __restrict__ distIn, u32 total,
                                                    u32* __restrict__ nextOut, u32* __restrict__ distOut)
{
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= total) return;
    const u32 nx = nextIn[j];
    distOut[j] = distIn[j] + distIn[nx];
    nextOut[j] = nextIn[nx];
}

// walk 2: every splitter writes the text bytes of its sub-list (ascending addresses: four bytes are gathered per store)
__global__ __launch_bounds__(256) void k_bwt_i_walk2(BwtView v, const BwtHdr* __restrict__ hd, const u32* __restrict__ base, const u64* __restrict__ rec,
                                                     const u32* __restrict__ splitNode, const u32* __restrict__ dEnd, u32 count, u32 limit)
{
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    if (c >= count) return;
    u32 node = splitNode[c];
    const int b = find_block(base, v.nBlocks, node);
    const u32 n = hd[b].n;
    u8* dst = v.dst[b];
    const bool al = (reinterpret_cast<uintptr_t>(dst) & 3) == 0;
    u32 d = dEnd[c];
    u32 steps = 0;
    u32 curW = 0xFFFFFFFFu, acc = 0, mask = 0;
    auto flush = [&]() {
        if (mask == 0xF && al) { reinterpret_cast<u32*>(dst)[curW] = acc; }
        else { for (u32 k = 0; k < 4; k++) if ((mask >> k) & 1) dst[4 * curW + k] = (u8)(acc >> (8 * k)); }
    };
    while (true) {
        const u64 r = rec[node];
        if (d < n) {
            const u32 pos = n - 1 - d;
            if ((pos >> 2) != curW) { if (mask) flush(); curW = pos >> 2; acc = 0; mask = 0; }
            acc |= (u32)((r >> 32) & 0xFF) << (8 * (pos & 3));
            mask |= 1u << (pos & 3);
        }
        if (((r >> 31) & 1) || d == 0 || ++steps > limit) break;
        node = (u32)r & 0x7FFFFFFFu;
        d--;
    }
    if (mask) flush();
}

__global__ void k_bwt_i_tiny(BwtView v, const BwtHdr* __restrict__ hd)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= v.nBlocks) return;
    const BwtHdr h = hd[b];
    if (h.okFlag && h.n == 1) v.dst[b][0] = v.src[b][h.hdr];     // BWT.cpp:152-158
}

size_t bwt_inverse_scratch_bytes(int nBlocks, u32 VS, size_t total)
{
    size_t primSort = 0, primScan = 0;
    rocprim::radix_sort_pairs(nullptr, primSort, (u32*)nullptr, (u32*)nullptr, (u32*)nullptr, (u32*)nullptr, total, 0u, 32u, (hipStream_t)0);
    rocprim::exclusive_scan(nullptr, primScan, (u32*)nullptr, (u32*)nullptr, 0u, total, rocprim::plus<u32>(), (hipStream_t)0);
    const size_t prim = primSort > primScan ? primSort : primScan;
    const size_t perTiles = ((size_t)VS + IT - 1) / IT;
    const u32 segT = bwt_i_seg_tiles((u32)perTiles);
    return align256(1024 * perTiles * (size_t)nBlocks) + align256(1024 * ((perTiles + segT - 1) / segT) * (size_t)nBlocks) + align256(1024 * (size_t)nBlocks) + align256(8 * total) + align256(8 * (total / 32 + 2)) + align256(4 * (total / 32 + 2)) + 6 * align256(4 * (total / 8 + 4096 + 3 * (size_t)nBlocks)) +
           align256(sizeof(BwtHdr) * (size_t)nBlocks) + align256(4ull * (nBlocks + 2)) + align256(prim) + 16384;
}

int launch_bwt_inverse(hipStream_t s, const XfStage& st, void* scratch, size_t scratchBytes, u32* h_pinned)
{
    BwtView v; v.src = st.src; v.dst = st.dst; v.len = st.len; v.cap = st.cap; v.VS = st.maxLen; v.nBlocks = st.nBlocks;
    const size_t maxTotal = (size_t)st.nBlocks * v.VS;
    if (maxTotal >= (1ull << 31)) return -2;                 // node ids share a word with the splitter flag
    const size_t maxSplit = maxTotal / 8 + 4096 + 3 * (size_t)st.nBlocks;
    u8* q = reinterpret_cast<u8*>(scratch);
    auto take = [&](size_t sz) { u8* r = q; q += align256(sz); return r; };
    const int perTiles = (int)((v.VS + IT - 1) / IT);
    u32* tileHist = (u32*)take(4ull * 256 * (size_t)perTiles * st.nBlocks);
    const u32 segT = bwt_i_seg_tiles((u32)perTiles), nSeg = ((u32)perTiles + segT - 1) / segT;
    u32* segSum = (u32*)take(4ull * 256 * (size_t)nSeg * st.nBlocks);
    u32* Cb = (u32*)take(4ull * 256 * st.nBlocks);
    u32* term = (u32*)take(4ull * (st.nBlocks + 1));
    u64* rec = (u64*)take(8 * maxTotal);
    u32* flags = (u32*)take(8 * (maxTotal / 32 + 2)); u32* scanIdx = (u32*)take(4 * (maxTotal / 32 + 2));   // splitter bit map + word counts, word prefix
    u32* splitNode = (u32*)take(4 * maxSplit);
    u32* nA = (u32*)take(4 * maxSplit); u32* nB = (u32*)take(4 * maxSplit);
    u32* dA = (u32*)take(4 * maxSplit); u32* dB = (u32*)take(4 * maxSplit);
    u32* spare = (u32*)take(4 * maxSplit); (void)spare;
    BwtHdr* hd = (BwtHdr*)take(sizeof(BwtHdr) * (size_t)st.nBlocks);
    u32* base = (u32*)take(4ull * (st.nBlocks + 2));
    void* prim = q;
    const size_t primBytes = scratchBytes - (size_t)(q - reinterpret_cast<u8*>(scratch));
    { KScope ks_("k_bwt_i_header"); hipLaunchKernelGGL(k_bwt_i_header, dim3(1), dim3(64), 0, s, v, hd, base, st.ok, st.newLen); }
    if (hipMemcpyAsync(h_pinned, base + st.nBlocks, 4, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    const u32 total = h_pinned[0];
    { KScope ks_("k_bwt_i_tiny"); hipLaunchKernelGGL(k_bwt_i_tiny, dim3((st.nBlocks + 63) / 64), dim3(64), 0, s, v, hd); }
    if (total == 0) return 0;
    const dim3 gridT((unsigned)perTiles, st.nBlocks);
    { KScope ks_("k_bwt_i_hist"); hipLaunchKernelGGL(k_bwt_i_hist, gridT, dim3(256), 0, s, v, hd, perTiles, tileHist); }
    { KScope ks_("k_bwt_i_scan"); hipLaunchKernelGGL(k_bwt_i_scan, dim3(nSeg, st.nBlocks), dim3(256), 0, s, hd, perTiles, segT, nSeg, tileHist, segSum);
      hipLaunchKernelGGL(k_bwt_i_scan2, dim3(st.nBlocks), dim3(256), 0, s, v, hd, base, nSeg, segSum, Cb, term); }
    { KScope ks_("k_bwt_i_links"); hipLaunchKernelGGL(k_bwt_i_links, gridT, dim3(256), 0, s, v, hd, base, perTiles, tileHist, segT, nSeg, segSum, Cb, term, rec); }
    const u32 nWords = (total + 31) / 32;
    u32* bits = flags; u32* wcount = flags + nWords; u32* wprefix = scanIdx;           // (the two node-sized arrays of the first version)
    { KScope ks_("k_bwt_i_flags"); hipLaunchKernelGGL(k_bwt_i_flags, GRID1(nWords), hd, base, term, st.nBlocks, total, bits, wcount); }
    size_t pb;
    pb = primBytes;
    { KScope ks_("bwt_i_scan_sum"); if (rocprim::exclusive_scan(prim, pb, wcount, wprefix, 0u, (size_t)nWords, rocprim::plus<u32>(), s) != hipSuccess) return -1; }
    { KScope ks_("k_bwt_i_compact"); hipLaunchKernelGGL(k_bwt_i_compact, GRID1(nWords), bits, wprefix, nWords, splitNode); }
    hipMemcpyAsync(h_pinned, wprefix + (nWords - 1), 4, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(h_pinned + 1, wcount + (nWords - 1), 4, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    const u32 count = h_pinned[0] + h_pinned[1];
    if (count == 0 || count > maxSplit) return -3;
    const u32 limit = v.VS + 1;
    { KScope ks_("k_bwt_i_walk1"); hipLaunchKernelGGL(k_bwt_i_walk1, GRID1(count), rec, splitNode, bits, wprefix, count, limit, nA, dA); }
    for (u32 span = 1; span < count; span <<= 1) {
        { KScope ks_("k_bwt_i_jump"); hipLaunchKernelGGL(k_bwt_i_jump, GRID1(count), nA, dA, count, nB, dB); }
        std::swap(nA, nB); std::swap(dA, dB);
    }
    { KScope ks_("k_bwt_i_walk2"); hipLaunchKernelGGL(k_bwt_i_walk2, GRID1(count), v, hd, base, rec, splitNode, dA, count, limit); }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace knz
