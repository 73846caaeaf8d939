// Burrows-Wheeler transform (block codec) on gfx950, all blocks of a batch at once.
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
//   forward  one suffix sort for the whole batch: the sort key carries the block id, suffix
//            comparisons stop at the block end (rank 0 past the end = "shorter sorts first").
//            Prefix doubling with group discarding (Larsson-Sadakane order refinement): round 0
//            sorts 9-bit symbols packed in a 64-bit key, later rounds sort only the members of still
//            unsorted groups on (group head, rank[i+h]). The two device-wide primitives (LSD radix
//            sort of 64-bit keys, scans) come from rocPRIM; everything else is hand-written.
//            (Measured alternative, rejected: rocprim::segmented_radix_sort_pairs on 32-bit rank[i+h] keys
//            per group is 2.4x slower per round here -- tens of millions of tiny segments.)
//   inverse  psi by a stable 8-bit counting sort of the BWT symbols, then list ranking by pointer
//            jumping (log2 n rounds of next[next[j]]) gives every F-position its text offset; the
//            8 chains the reference walks serially become one data-parallel scatter.
#include "common.hpp"
#include "stages.hpp"

#include <cstring>
#include <algorithm>
#include <rocprim/rocprim.hpp>

namespace knz {

struct BwtView {
    const u8* const* src;
    u8* const* dst;
    const u32* len;
    const u32* cap;
    u32 VS;                // virtual stride: position id = b * VS + offset
    int nBlocks;
};

__device__ __forceinline__ int bwt_chunks(u32 n) { return n < 256 ? 1 : 8; }

__device__ __forceinline__ bool bwt_fwd_applies(u32 n, u32 cap, u32* pIdxSizeOut)
{
    if (n == 0) return false;
    if (cap < n + 33) return false;                       // getMaxEncodedLength, BWTBlockCodec.hpp:47-50
    u32 logBlockSize = (u32)ilog2_u32(n);
    if ((n & (n - 1)) != 0) logBlockSize++;
    const u32 pIndexSize = (logBlockSize + 7) >> 3;
    if (pIndexSize == 0 || pIndexSize >= 5) return false; // n == 1 -> 0 bytes -> refused (BWTBlockCodec.cpp:53-56)
    *pIdxSizeOut = pIndexSize;
    return true;
}

// base[b] = sum of active lengths of the blocks before b ; total in base[nBlocks]
__global__ void k_bwt_bases(BwtView v, u32* __restrict__ base, u8* __restrict__ ok)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u32 sum = 0;
    for (int b = 0; b < v.nBlocks; b++) {
        base[b] = sum;
        u32 ps;
        const bool a = bwt_fwd_applies(v.len[b], v.cap[b], &ps);
        ok[b] = a ? 1 : 0;
        if (a) sum += v.len[b];
    }
    base[v.nBlocks] = sum;
}

// round 0 keys: [block id | symbols as 9-bit values (byte + 1, 0 past the block end)]
__global__ __launch_bounds__(256) void k_bwt_f_init(BwtView v, const u32* __restrict__ base, const u8* __restrict__ ok, int bbits, int nsym,
                                                    u64* __restrict__ keys, u32* __restrict__ vals)
{
    const int b = blockIdx.y;
    if (!ok[b]) return;
    const u32 n = v.len[b];
    const u8* s = v.src[b];
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        u64 k = (u64)b;
        for (int q = 0; q < nsym; q++) {
            const u32 j = i + (u32)q;
            k = (k << 9) | (j < n ? (u64)s[j] + 1 : 0ull);
        }
        (void)bbits;
        keys[base[b] + i] = k;
        vals[base[b] + i] = (u32)b * v.VS + i;
    }
}

// flags[a] = 1 when element a starts a new group (differs from its predecessor)
__global__ __launch_bounds__(256) void k_bwt_flags(const u64* __restrict__ keys, u32 n, u32* __restrict__ headIdx)
{
    const u32 a = blockIdx.x * 256 + threadIdx.x;
    if (a >= n) return;
    const bool f = (a == 0) || (keys[a] != keys[a - 1]);
    headIdx[a] = f ? a : 0;
}

// first round: SA = sorted values; rank[pos] = group head slot; keep[a] = member of a group of size > 1
__global__ __launch_bounds__(256) void k_bwt_round0(const u64* __restrict__ keys, const u32* __restrict__ vals, const u32* __restrict__ head, u32 n,
                                                    u32* __restrict__ SA, u32* __restrict__ rank, u32* __restrict__ keep)
{
    const u32 a = blockIdx.x * 256 + threadIdx.x;
    if (a >= n) return;
    const u32 p = vals[a];
    SA[a] = p;
    rank[p] = head[a];
    const bool startsHere = (a == 0) || (keys[a] != keys[a - 1]);
    const bool nextStarts = (a + 1 >= n) || (keys[a + 1] != keys[a]);
    keep[a] = (startsHere && nextStarts) ? 0u : 1u;
}

// compaction: act[out] = value[a] for kept elements, given the exclusive scan of keep
__global__ __launch_bounds__(256) void k_compact(const u32* __restrict__ keep, const u32* __restrict__ scan, const u32* __restrict__ value, u32 n,
                                                 u32* __restrict__ out)
{
    const u32 a = blockIdx.x * 256 + threadIdx.x;
    if (a >= n) return;
    if (keep[a]) out[scan[a]] = value ? value[a] : a;
}

// keys of the active elements for offset h: (head slot of the group, rank[pos + h] + 1 or 0 past the block end)
__global__ __launch_bounds__(256) void k_bwt_keys(BwtView v, const u32* __restrict__ act, u32 nAct, const u32* __restrict__ SA,
                                                  const u32* __restrict__ rank, u32 h, int rbits, u64* __restrict__ keys, u32* __restrict__ vals)
{
    const u32 a = blockIdx.x * 256 + threadIdx.x;
    if (a >= nAct) return;
    const u32 slot = act[a];
    const u32 p = SA[slot];
    const u32 b = p / v.VS;
    const u32 off = p - b * v.VS;
    const u32 n = v.len[b];
    const u64 r2 = (off + h < n && off + h >= off) ? (u64)rank[p + h] + 1 : 0ull;
    keys[a] = ((u64)rank[p] << rbits) | r2;
    vals[a] = p;
}

// after sorting the active elements: write them back to their slots, new group heads, who stays active
__global__ __launch_bounds__(256) void k_bwt_place(const u32* __restrict__ act, u32 nAct, const u64* __restrict__ keys, const u32* __restrict__ vals,
                                                   const u32* __restrict__ headIdx, u32* __restrict__ SA, u32* __restrict__ rank, u32* __restrict__ keep)
{
    const u32 a = blockIdx.x * 256 + threadIdx.x;
    if (a >= nAct) return;
    const u32 p = vals[a];
    SA[act[a]] = p;
    rank[p] = act[headIdx[a]];
    const bool startsHere = (a == 0) || (keys[a] != keys[a - 1]);
    const bool nextStarts = (a + 1 >= nAct) || (keys[a + 1] != keys[a]);
    keep[a] = (startsHere && nextStarts) ? 0u : 1u;
}

// final: BWT bytes + header (BWTBlockCodec.cpp:58-86)
__global__ __launch_bounds__(256) void k_bwt_f_emit(BwtView v, const u32* __restrict__ base, const u8* __restrict__ ok, const u32* __restrict__ SA,
                                                    const u32* __restrict__ rank, u32* __restrict__ newLen)
{
    const int b = blockIdx.y;
    if (!ok[b]) return;
    const u32 n = v.len[b];
    u32 pIndexSize = 0;
    bwt_fwd_applies(n, v.cap[b], &pIndexSize);
    const int chunks = bwt_chunks(n);
    const u32 hdr = 1 + (u32)chunks * pIndexSize;
    const u8* s = v.src[b];
    u8* d = v.dst[b];
    const u32 vb = (u32)b * v.VS;
    const u32 r0 = rank[vb] - base[b];                     // rank of suffix 0
    for (u32 r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
        const u32 p = SA[base[b] + r] - vb;
        if (p != 0) d[hdr + 1 + r - (r > r0 ? 1u : 0u)] = s[p - 1];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d[hdr] = s[n - 1];
        const u32 st = n / (u32)chunks;
        const u32 step = ((u32)chunks * st == n) ? st : st + 1;
        const u32 logNbChunks = (u32)ilog2_u32((u32)chunks);
        d[0] = (u8)((logNbChunks << 2) | (pIndexSize - 1));
        u32 idx = 1;
        for (int k = 0; k < chunks; k++) {
            const u32 prim = rank[vb + (u32)k * step] - base[b];   // primaryIndex - 1
            for (int sh = (int)(pIndexSize - 1) * 8; sh >= 0; sh -= 8) d[idx++] = (u8)(prim >> sh);
        }
        newLen[b] = hdr + n;
    }
}

// ------------------------------------------------------------------------------------------------
struct BwtScratch {
    u64* keysA; u64* keysB;
    u32* valsA; u32* valsB;
    u32* SA; u32* rank;
    u32* t0; u32* t1; u32* act; u32* act2;
    u32* base;
    void* prim; size_t primBytes;
    u32* d_count;
};

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t bwt_forward_scratch_bytes(int nBlocks, u32 VS, size_t total)
{
    size_t primSort = 0, primScan = 0;
    rocprim::radix_sort_pairs(nullptr, primSort, (u64*)nullptr, (u64*)nullptr, (u32*)nullptr, (u32*)nullptr, total, 0u, 64u, (hipStream_t)0);
    rocprim::inclusive_scan(nullptr, primScan, (u32*)nullptr, (u32*)nullptr, total, rocprim::maximum<u32>(), (hipStream_t)0);
    const size_t prim = align256(primSort > primScan ? primSort : primScan) + 4096;
    return 2 * align256(8 * total) + 6 * align256(4 * total) + 2 * align256(4 * total) + align256(4ull * nBlocks * VS) +
           align256(4ull * (nBlocks + 2)) + prim + 4096;
}

static void carve(u8* p, int nBlocks, u32 VS, size_t total, size_t bytes, BwtScratch* w)
{
    u8* q = p;
    auto take = [&](size_t sz) { u8* r = q; q += align256(sz); return r; };
    w->keysA = (u64*)take(8 * total); w->keysB = (u64*)take(8 * total);
    w->valsA = (u32*)take(4 * total); w->valsB = (u32*)take(4 * total);
    w->SA = (u32*)take(4 * total);
    w->t0 = (u32*)take(4 * total); w->t1 = (u32*)take(4 * total);
    w->act = (u32*)take(4 * total); w->act2 = (u32*)take(4 * total);
    w->rank = (u32*)take(4ull * nBlocks * VS);
    w->base = (u32*)take(4ull * (nBlocks + 2));
    w->d_count = (u32*)take(256);
    w->prim = q;
    w->primBytes = bytes - (size_t)(q - p);
}

#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, s

// Returns 0 or a negative HIP error. Synchronises the stream (active-set sizes are read back per round).
int launch_bwt_forward(hipStream_t s, const XfStage& st, void* scratch, size_t scratchBytes, u32* h_pinned)
{
    BwtView v; v.src = st.src; v.dst = st.dst; v.len = st.len; v.cap = st.cap; v.VS = st.maxLen; v.nBlocks = st.nBlocks;
    // worst-case total = nBlocks * VS ; carve for that
    const size_t maxTotal = (size_t)st.nBlocks * v.VS;
    BwtScratch w;
    carve(reinterpret_cast<u8*>(scratch), st.nBlocks, v.VS, maxTotal, scratchBytes, &w);
    { KScope ks_("k_bwt_f_bases"); hipLaunchKernelGGL(k_bwt_bases, dim3(1), dim3(64), 0, s, v, w.base, st.ok); }
    hipMemsetAsync(st.newLen, 0, sizeof(u32) * st.nBlocks, s);
    if (hipMemcpyAsync(h_pinned, w.base + st.nBlocks, 4, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    const u32 total = h_pinned[0];
    if (total == 0) return 0;
    int bbits = 0;
    while ((1 << bbits) < st.nBlocks) bbits++;
    const int nsym = (64 - bbits) / 9;
    const dim3 gridB((unsigned)std::min<size_t>(((size_t)v.VS + 255) / 256, 4096), st.nBlocks);
    { KScope ks_("k_bwt_f_init"); hipLaunchKernelGGL(k_bwt_f_init, gridB, dim3(256), 0, s, v, w.base, st.ok, bbits, nsym, w.keysA, w.valsA); }
    size_t pb = w.primBytes;
    { KScope ks_("bwt_f_sort_round0");
      if (rocprim::radix_sort_pairs(w.prim, pb, w.keysA, w.keysB, w.valsA, w.valsB, (size_t)total, 0u, (unsigned)(bbits + 9 * nsym), s) != hipSuccess) return -1; }
    { KScope ks_("k_bwt_f_flags"); hipLaunchKernelGGL(k_bwt_flags, GRID1(total), w.keysB, total, w.t0); }
    pb = w.primBytes;
    { KScope ks_("bwt_f_scan_max"); if (rocprim::inclusive_scan(w.prim, pb, w.t0, w.t1, (size_t)total, rocprim::maximum<u32>(), s) != hipSuccess) return -1; }
    { KScope ks_("k_bwt_f_round0"); hipLaunchKernelGGL(k_bwt_round0, GRID1(total), w.keysB, w.valsB, w.t1, total, w.SA, w.rank, w.t0); }
    // active slots = members of groups with more than one element
    u32 nAct = total;
    auto compact = [&](u32 n, const u32* value, u32* out) -> int {
        size_t pbs = w.primBytes;
        { KScope ks_("bwt_f_scan_sum"); if (rocprim::exclusive_scan(w.prim, pbs, w.t0, w.t1, 0u, (size_t)n, rocprim::plus<u32>(), s) != hipSuccess) return -1; }
        { KScope ks_("k_bwt_f_compact"); hipLaunchKernelGGL(k_compact, GRID1(n), w.t0, w.t1, value, n, out); }
        // count = scan[n-1] + keep[n-1]
        hipMemcpyAsync(h_pinned, w.t1 + (n - 1), 4, hipMemcpyDeviceToHost, s);
        hipMemcpyAsync(h_pinned + 1, w.t0 + (n - 1), 4, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) return -1;
        return (int)(h_pinned[0] + h_pinned[1]);
    };
    int cnt = compact(total, nullptr, w.act);
    if (cnt < 0) return -1;
    nAct = (u32)cnt;
    int rbits = 1;
    while ((1ull << rbits) < (u64)total + 2) rbits++;
    u32 h = (u32)nsym;
    u32* act = w.act; u32* act2 = w.act2;
    while (nAct > 0) {
        { KScope ks_("k_bwt_f_keys"); hipLaunchKernelGGL(k_bwt_keys, GRID1(nAct), v, act, nAct, w.SA, w.rank, h, rbits, w.keysA, w.valsA); }
        pb = w.primBytes;
        { KScope ks_("bwt_f_sort_round"); if (rocprim::radix_sort_pairs(w.prim, pb, w.keysA, w.keysB, w.valsA, w.valsB, (size_t)nAct, 0u, (unsigned)(2 * rbits), s) != hipSuccess) return -1; }
        { KScope ks_("k_bwt_f_flags"); hipLaunchKernelGGL(k_bwt_flags, GRID1(nAct), w.keysB, nAct, w.t0); }
        pb = w.primBytes;
        { KScope ks_("bwt_f_scan_max"); if (rocprim::inclusive_scan(w.prim, pb, w.t0, w.t1, (size_t)nAct, rocprim::maximum<u32>(), s) != hipSuccess) return -1; }
        { KScope ks_("k_bwt_f_place"); hipLaunchKernelGGL(k_bwt_place, GRID1(nAct), act, nAct, w.keysB, w.valsB, w.t1, w.SA, w.rank, w.t0); }
        cnt = compact(nAct, act, act2);
        if (cnt < 0) return -1;
        nAct = (u32)cnt;
        std::swap(act, act2);
        if (h >= 0x80000000u) break;
        h <<= 1;
    }
    { KScope ks_("k_bwt_f_emit"); hipLaunchKernelGGL(k_bwt_f_emit, gridB, dim3(256), 0, s, v, w.base, st.ok, w.SA, w.rank, st.newLen); }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

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

__global__ __launch_bounds__(256) void k_bwt_i_keys(BwtView v, const BwtHdr* __restrict__ hd, const u32* __restrict__ base,
                                                    u32* __restrict__ keys, u32* __restrict__ vals)
{
    const int b = blockIdx.y;
    const BwtHdr h = hd[b];
    if (!h.okFlag || h.n < 2) return;
    const u8* s = v.src[b] + h.hdr;
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < h.n; i += gridDim.x * 256) {
        keys[base[b] + i] = ((u32)b << 8) | s[i];              // stable sort by (block, symbol)
        vals[base[b] + i] = i;
    }
}

// ---- list ranking with random splitters (Helman-JaJa style) ------------------------------------
// Node j = F-position (global slot). rec[j] = next node (31 bits) | "next is a splitter" (bit 31) | symbol << 32.
// The node holding BWT index 0 is the end of the text (self loop). Splitters: a pseudo-random 1/64 of the
// nodes + every block's chain head + the terminals. Each splitter walks to the next splitter (sub-list
// length), the short splitter list is ranked by pointer jumping, then each splitter walks its sub-list
// again and writes the text. Two O(n) random-access passes instead of log2(n) of them.
constexpr int SPLIT_LOG = 6;

__device__ __forceinline__ bool hash_split(u32 j) { return ((j * 2654435761u) >> (32 - SPLIT_LOG)) == 0; }

__global__ __launch_bounds__(256) void k_bwt_i_links(const BwtHdr* __restrict__ hd, const u32* __restrict__ base, u32 total,
                                                     const u32* __restrict__ keysSorted, const u32* __restrict__ valsSorted,
                                                     u64* __restrict__ rec, u32* __restrict__ flags)
{
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= total) return;
    const u32 key = keysSorted[j];
    const u32 b = key >> 8;
    const u32 i = valsSorted[j];
    const u32 pIdx = hd[b].pIdx;
    u32 nx;
    if (i == 0) nx = j;                                       // end of text
    else nx = base[b] + ((i < pIdx) ? i - 1 : i);            // BWT.cpp:203-215
    rec[j] = (u64)nx | ((u64)(key & 0xFF) << 32);
    u32 f = (hash_split(j) || i == 0) ? 1u : 0u;
    if (j == base[b] + pIdx - 1) f = 1;                      // chain head of the block
    flags[j] = f;
}

__global__ __launch_bounds__(256) void k_bwt_i_fold(u64* __restrict__ rec, const u32* __restrict__ flags, u32 total)
{
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= total) return;
    const u64 r = rec[j];
    const u32 nx = (u32)r & 0x7FFFFFFFu;
    if (flags[nx]) rec[j] = r | 0x80000000ull;
}

// walk 1: sub-list length and successor splitter (compact indices)
__global__ __launch_bounds__(256) void k_bwt_i_walk1(const u64* __restrict__ rec, const u32* __restrict__ splitNode, const u32* __restrict__ scanIdx,
                                                     u32 count, u32 limit, u32* __restrict__ succ, u32* __restrict__ dist)
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
        if ((r >> 31) & 1) { succ[c] = scanIdx[nx]; break; }
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

// walk 2: every splitter writes the text bytes of its sub-list
__global__ __launch_bounds__(256) void k_bwt_i_walk2(BwtView v, const BwtHdr* __restrict__ hd, const u64* __restrict__ rec,
                                                     const u32* __restrict__ splitNode, const u32* __restrict__ keysSorted,
                                                     const u32* __restrict__ dEnd, u32 count, u32 limit)
{
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    if (c >= count) return;
    u32 node = splitNode[c];
    const u32 b = keysSorted[node] >> 8;
    const u32 n = hd[b].n;
    u8* dst = v.dst[b];
    u32 d = dEnd[c];
    u32 steps = 0;
    while (true) {
        const u64 r = rec[node];
        if (d < n) dst[n - 1 - d] = (u8)(r >> 32);
        if (((r >> 31) & 1) || d == 0 || ++steps > limit) break;
        node = (u32)r & 0x7FFFFFFFu;
        d--;
    }
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
    return 4 * align256(4 * total) + align256(8 * total) + 2 * align256(4 * total) + 6 * align256(4 * (total / 8 + 4096 + 3 * (size_t)nBlocks)) +
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
    u32* keysA = (u32*)take(4 * maxTotal); u32* keysB = (u32*)take(4 * maxTotal);
    u32* valsA = (u32*)take(4 * maxTotal); u32* valsB = (u32*)take(4 * maxTotal);
    u64* rec = (u64*)take(8 * maxTotal);
    u32* flags = (u32*)take(4 * maxTotal); u32* scanIdx = (u32*)take(4 * maxTotal);
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
    const dim3 gridB((unsigned)std::min<size_t>(((size_t)v.VS + 255) / 256, 4096), st.nBlocks);
    { KScope ks_("k_bwt_i_keys"); hipLaunchKernelGGL(k_bwt_i_keys, gridB, dim3(256), 0, s, v, hd, base, keysA, valsA); }
    size_t pb = primBytes;
    int bbits = 0;
    while ((1 << bbits) < st.nBlocks) bbits++;
    { KScope ks_("bwt_i_sort_symbols"); if (rocprim::radix_sort_pairs(prim, pb, keysA, keysB, valsA, valsB, (size_t)total, 0u, (unsigned)(8 + bbits), s) != hipSuccess) return -1; }
    { KScope ks_("k_bwt_i_links"); hipLaunchKernelGGL(k_bwt_i_links, GRID1(total), hd, base, total, keysB, valsB, rec, flags); }
    { KScope ks_("k_bwt_i_fold"); hipLaunchKernelGGL(k_bwt_i_fold, GRID1(total), rec, flags, total); }
    pb = primBytes;
    { KScope ks_("bwt_i_scan_sum"); if (rocprim::exclusive_scan(prim, pb, flags, scanIdx, 0u, (size_t)total, rocprim::plus<u32>(), s) != hipSuccess) return -1; }
    { KScope ks_("k_bwt_i_compact"); hipLaunchKernelGGL(k_compact, GRID1(total), flags, scanIdx, (const u32*)nullptr, total, splitNode); }
    hipMemcpyAsync(h_pinned, scanIdx + (total - 1), 4, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(h_pinned + 1, flags + (total - 1), 4, hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    const u32 count = h_pinned[0] + h_pinned[1];
    if (count == 0 || count > maxSplit) return -3;
    const u32 limit = v.VS + 1;
    { KScope ks_("k_bwt_i_walk1"); hipLaunchKernelGGL(k_bwt_i_walk1, GRID1(count), rec, splitNode, scanIdx, count, limit, nA, dA); }
    for (u32 span = 1; span < count; span <<= 1) {
        { KScope ks_("k_bwt_i_jump"); hipLaunchKernelGGL(k_bwt_i_jump, GRID1(count), nA, dA, count, nB, dB); }
        std::swap(nA, nB); std::swap(dA, dB);
    }
    { KScope ks_("k_bwt_i_walk2"); hipLaunchKernelGGL(k_bwt_i_walk2, GRID1(count), v, hd, rec, splitNode, keysB, dA, count, limit); }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace knz
