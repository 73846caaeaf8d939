// SRT (sorted rank transform) on gfx950: forward one wave per 4 KiB tile, inverse one wave per block.
//
// Reference being replaced: transform/SRT.cpp:22-109 (forward), :111-204 (inverse), :206-244 (preprocess:
// symbols by descending frequency, ties by ascending value), :246-308 (header: 256 var-ints).
//
// What the transform is: a move-to-front over the RUNS of the input (initial list = symbols in order of first
// appearance), whose ranks are not written in place but appended to one bucket per symbol (buckets laid out in
// "preprocess" order); the bytes of a run after its head are zeros in the same bucket.
//
//   forward  tile parallel (round 3; one wave per block before): the ranks are move-to-front ranks from an initial list in
//            order of first appearance, so the block is cut into 4 KiB tiles whose start lists and bucket offsets come from
//            scans over per-tile tables (see "forward, tile parallel" below); inside a tile the 256-entry list is held in
//            registers (4 entries per lane, SWAR zero-byte test + ballot to find a symbol, one DPP shift to rotate).
//   inverse  inherently serial over runs (which bucket is read next depends on the list head), so one wave
//            runs the chain: the next 64 bytes of each of the 256 buckets are cached in LDS (16 KiB), the
//            length of a run is a ballot over the cached bucket bytes, the run is written by the whole wave,
//            and the list update (remove head, insert at rank r) is one cross-lane shift.
#include "common.hpp"
#include "stages.hpp"

namespace knz {

// rotate list positions [0, rank] right by one and put `front` at position 0 (rank = 4*lane0 + byteIdx)
__device__ __forceinline__ u32 srt_to_front(u32 w, int lane, int lane0, int byteIdx, u32 front)
{
    const u32 prev = (u32)__builtin_amdgcn_update_dpp(0, (int)w, 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane i <- lane i-1)
    const u32 carry = (lane == 0) ? front : (prev >> 24);
    const u32 shifted = (w << 8) | carry;
    if (lane < lane0) return shifted;
    if (lane == lane0) {
        const u32 mask = (byteIdx == 3) ? 0xFFFFFFFFu : ((1u << (8 * (byteIdx + 1))) - 1u);
        return (shifted & mask) | (w & ~mask);
    }
    return w;
}

// positions [1, rank] move down by one; position rank receives `ins` when insert is set, else keeps its value
// (memmove(&r2s[0], &r2s[1], rank) [+ r2s[rank] = ins], SRT.cpp:189-199)
__device__ __forceinline__ u32 srt_drop_front(u32 w, int lane, u32 rank, bool insert, u32 ins)
{
    const int lane0 = (int)(rank >> 2);
    const u32 byteIdx = rank & 3;
    const u32 next = (u32)__builtin_amdgcn_update_dpp(0, (int)w, 0x130, 0xF, 0xF, false);   // wave_shl:1 (lane i <- lane i+1)
    const u32 shifted = (w >> 8) | (next << 24);
    if (lane < lane0) return shifted;
    if (lane == lane0) {
        const u32 low = (1u << (8 * byteIdx)) - 1u;                 // bytes below byteIdx
        u32 r = (shifted & low) | (w & ~low);
        if (insert) r = (r & ~(0xFFu << (8 * byteIdx))) | (ins << (8 * byteIdx));
        return r;
    }
    return w;
}

// order index of every present symbol: number of present symbols with a larger frequency, or an equal one and
// a smaller value (SRT.cpp:206-244 is a shell sort under exactly this total order).  Returns nbSymbols.
__device__ u32 srt_order(int lane, const u32* freqs, u8* symbols)
{
    u32 present = 0;
    for (int k = 0; k < 4; k++) {
        const int c = 4 * lane + k;
        const u32 fc = freqs[c];
        if (fc == 0) continue;
        present++;
        u32 r = 0;
        for (int q = 0; q < 256; q++) {
            const u32 fq = freqs[q];
            r += (fq > fc || (fq == fc && q < c && fq != 0)) ? 1u : 0u;
        }
        symbols[r] = (u8)c;
    }
    return wave_sum(present);
}

__global__ __launch_bounds__(256) void k_srt_zero(XfStage st)
{
    const int b = blockIdx.y;
    const u32 length = st.len[b];
    if (length == 0 || st.cap[b] < length + 1024) return;
    const u32 total = length + 1024;
    uint4* d = reinterpret_cast<uint4*>(st.dst[b]);
    const bool al = (reinterpret_cast<uintptr_t>(st.dst[b]) & 15) == 0;
    const u32 n16 = (total + 15) >> 4;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) {
        if (al && 16 * i + 16 <= st.cap[b]) d[i] = z;
        else for (u32 k = 16 * i; k < 16 * i + 16 && k < st.cap[b] && k < total; k++) st.dst[b][k] = 0;
    }
}

__global__ __launch_bounds__(64) void k_srt_inverse(XfStage st)
{
    const int b = blockIdx.x;
    const int lane = lane_id();
    const int total = (int)st.len[b];
    if (lane == 0) { st.ok[b] = 0; st.newLen[b] = 0; }
    if (total == 0) { if (lane == 0) st.ok[b] = 1; return; }
    if (total < 256) return;                                   // SRT.cpp:122
    __shared__ u32 freqs[256];
    __shared__ int buckets[256], bucketEnds[256];
    __shared__ u8 symbols[256];
    __shared__ u32 listw[64];
    __shared__ u8 hdrBytes[1280];
    __shared__ int shHdr;
    __shared__ u8 cache[256][64];
    __shared__ u32 cbase[256];
    const u8* in = st.src[b];
    for (int i = lane; i < 1280; i += 64) hdrBytes[i] = (i < total) ? in[i] : 0;
    for (int i = lane; i < 256; i += 64) { buckets[i] = 0; bucketEnds[i] = 0; symbols[i] = 0; cbase[i] = 0xFFFFFFFFu; }
    listw[lane] = 0;
    __syncthreads();
    if (lane == 0) {
        // decodeHeader, SRT.cpp:279-308
        int srcIdx = 0;
        bool okh = true;
        for (int i = 0; i < 256 && okh; i++) {
            u32 res = 0;
            int shift = 0;
            for (int j = 0; j < 5; j++) {
                if (srcIdx >= total) { okh = false; break; }
                const u32 val = hdrBytes[srcIdx++];
                res |= ((val & 0x7F) << shift);
                if ((val & 0x80) == 0) break;
                if (j == 4) { okh = false; break; }
                shift += 7;
            }
            freqs[i] = res;
        }
        shHdr = okh ? srcIdx : -1;
    }
    __syncthreads();
    const int hdr = shHdr;
    if (hdr < 0) return;
    const int length = total - hdr;
    if (length < 0 || (u32)length > st.cap[b]) return;
    const u8* src = in + hdr;
    int nbSymbols = (int)srt_order(lane, freqs, symbols);
    __syncthreads();
    __shared__ int shOk;
    if (lane == 0) {
        u8* listb = reinterpret_cast<u8*>(listw);
        int okb = 1;
        int bucketPos = 0;
        for (int i = 0; i < nbSymbols; i++) {
            const u8 c = symbols[i];
            if ((bucketPos < 0) || (bucketPos >= length)) { okb = 0; break; }
            listb[src[bucketPos]] = c;
            buckets[c] = bucketPos + 1;
            bucketPos += (int)freqs[c];
            bucketEnds[c] = bucketPos;
        }
        shOk = okb;
    }
    __syncthreads();
    if (!shOk) return;
    u32 w = listw[lane];
    u8* dst = st.dst[b];
    u32 c = (u32)__builtin_amdgcn_readfirstlane((int)w) & 0xFF;
    int i = 0;
    while (i < length) {
        int p = buckets[c];
        const int e = bucketEnds[c];
        // zeros that follow in c's bucket, then its terminator (next rank of c, or exhaustion)
        int z = 0;
        u32 term = 0;
        while (p < e) {
            if ((u32)p < cbase[c] || (u32)p >= cbase[c] + 64u) {
                const int q = p + lane;
                __syncthreads();
                cache[c][lane] = (q < length) ? src[q] : (u8)0;        // reads past the body are zeros (the reference would read out of bounds)
                cbase[c] = (u32)p;                                      // every lane stores the same value
                __syncthreads();
            }
            const u32 off = (u32)p - cbase[c];
            const int avail = ((int)(64u - off) < e - p) ? (int)(64u - off) : (e - p);
            const u32 byte = ((int)lane < avail) ? (u32)cache[c][off + (u32)lane] : 0u;
            const u64 nz = __ballot(byte != 0);
            if (nz) {
                const int k = __ffsll((long long)nz) - 1;
                z += k;
                term = (u32)__builtin_amdgcn_readlane((int)byte, k);
                p += k + 1;
                break;
            }
            z += avail;
            p += avail;
        }
        int emit = z + 1;
        const bool cut = emit >= length - i;
        if (cut) emit = length - i;
        for (int k = lane; k < emit; k += 64) dst[i + k] = (u8)c;
        i += emit;
        if (cut) break;
        buckets[c] = p;                                         // every lane stores the same value
        if (term != 0) {
            w = srt_drop_front(w, lane, term, true, c);
        } else {
            if (nbSymbols == 1) {                               // SRT.cpp:194-195: the last symbol fills the rest
                for (int k = i + lane; k < length; k += 64) dst[k] = (u8)c;
                break;
            }
            nbSymbols--;
            w = srt_drop_front(w, lane, (u32)nbSymbols, false, 0);
        }
        c = (u32)__builtin_amdgcn_readfirstlane((int)w) & 0xFF;
    }
    if (lane == 0) { st.ok[b] = 1; st.newLen[b] = (u32)length; }
}

// ------------------------------------------------------------------------------------------------
// forward, tile parallel
// ------------------------------------------------------------------------------------------------
// The ranks SRT writes are plain move-to-front ranks of the byte sequence (a byte inside a run finds its symbol at the front:
// rank 0, list unchanged) from an initial list in order of first appearance; what is special is where they go: the k-th
// occurrence of symbol c lands at bucket(c) + k. So the block is cut into tiles of 4 KiB like MTFT: the list at a tile start
// is the symbols seen so far by their last occurrence (exclusive prefix maximum over per-tile tables), then the ones still to
// come by their first occurrence in the block; the output offset of a tile inside every bucket is an exclusive prefix sum over
// per-tile histograms. Both scans run in two levels (groups of ~sqrt(tiles) tiles, then over the groups). The body of the output
// is zero-filled first, only non-zero ranks are stored.
constexpr u32 SRT_T = 4096;

__host__ __device__ inline u32 srt_seg_tiles(u32 perTiles) { u32 g = 1; while (g * g < perTiles) g <<= 1; return g; }

struct SrtWs {
    u32* tileCnt;      // [nBlocks][perTiles][256] counts, then exclusive inside the tile's segment
    u32* tileLast;     // [nBlocks][perTiles][256] last occurrence (position + 1), then exclusive maximum inside the segment
    u32* segSum;       // [nBlocks][nSeg][256]
    u32* segMax;       // [nBlocks][nSeg][256]
    u32* firstOcc;     // [nBlocks][256] first occurrence of the symbol in the block (0xFFFFFFFF: absent)
    u32* bstart;       // [nBlocks][256] start of the symbol's bucket in the body
    u32* hdrLen;       // [nBlocks] (0: the transform does not apply)
    int perTiles; u32 segT, nSeg;
};

__global__ __launch_bounds__(64) void k_srt_f_tiles(XfStage st, SrtWs w)
{
    const int b = blockIdx.y;
    const u32 n = st.len[b];
    const u32 tbase = blockIdx.x * SRT_T;
    if (tbase >= n || st.cap[b] < n + 1024) return;
    const u8* s = st.src[b];
    __shared__ u32 cnt[256], last[256], first[256];
    const int lane = lane_id();
    for (int i = lane; i < 256; i += 64) { cnt[i] = 0; last[i] = 0; first[i] = 0xFFFFFFFFu; }
    __syncthreads();
    const u32 end = (tbase + SRT_T < n) ? tbase + SRT_T : n;
    for (u32 i = tbase + lane; i < end; i += 64) { const u32 c = s[i]; atomicAdd(&cnt[c], 1u); atomicMax(&last[c], i + 1); atomicMin(&first[c], i); }
    __syncthreads();
    const size_t o = ((size_t)b * w.perTiles + blockIdx.x) * 256;
    for (int i = lane; i < 256; i += 64) {
        w.tileCnt[o + i] = cnt[i];
        w.tileLast[o + i] = last[i];
        if (first[i] != 0xFFFFFFFFu) atomicMin(&w.firstOcc[b * 256 + i], first[i]);
    }
}

__global__ __launch_bounds__(256) void k_srt_f_scan1(XfStage st, SrtWs w)
{
    const int b = blockIdx.y;
    const u32 seg = blockIdx.x;
    const u32 n = st.len[b];
    const u32 nT = (st.cap[b] < n + 1024) ? 0u : (n + SRT_T - 1) / SRT_T;
    const u32 t0 = seg * w.segT, t1 = (t0 + w.segT < nT) ? t0 + w.segT : nT;
    u32* pc = w.tileCnt + (size_t)b * w.perTiles * 256 + threadIdx.x;
    u32* pl = w.tileLast + (size_t)b * w.perTiles * 256 + threadIdx.x;
    u32 sum = 0, mx = 0;
    for (u32 t = t0; t < t1; t++) {
        const u32 x = pc[(size_t)t * 256]; pc[(size_t)t * 256] = sum; sum += x;
        const u32 y = pl[(size_t)t * 256]; pl[(size_t)t * 256] = mx; mx = y > mx ? y : mx;
    }
    w.segSum[((size_t)b * w.nSeg + seg) * 256 + threadIdx.x] = sum;
    w.segMax[((size_t)b * w.nSeg + seg) * 256 + threadIdx.x] = mx;
}

// per block: scans over the segments, frequencies, bucket order (SRT.cpp:206-244), bucket starts, header (:246-277)
__global__ __launch_bounds__(256) void k_srt_f_scan2(XfStage st, SrtWs w)
{
    const int b = blockIdx.x;
    const int tid = (int)threadIdx.x;
    const u32 n = st.len[b];
    __shared__ u32 freqs[256];
    __shared__ u32 order[256];
    if (tid == 0) { st.ok[b] = 0; st.newLen[b] = 0; w.hdrLen[b] = 0; }
    if (n == 0) { if (tid == 0) st.ok[b] = 1; return; }
    if (st.cap[b] < n + 1024) return;                          // SRT.hpp:38
    u32* ps = w.segSum + (size_t)b * w.nSeg * 256 + tid;
    u32* pm = w.segMax + (size_t)b * w.nSeg * 256 + tid;
    u32 sum = 0, mx = 0;
    for (u32 g = 0; g < w.nSeg; g++) {
        const u32 x = ps[(size_t)g * 256]; ps[(size_t)g * 256] = sum; sum += x;
        const u32 y = pm[(size_t)g * 256]; pm[(size_t)g * 256] = mx; mx = y > mx ? y : mx;
    }
    freqs[tid] = sum;
    __syncthreads();
    // order index: number of present symbols with a larger frequency, or an equal one and a smaller value
    u32 r = 0;
    const u32 fc = freqs[tid];
    for (int q = 0; q < 256; q++) { const u32 fq = freqs[q]; r += (fq != 0 && (fq > fc || (fq == fc && q < tid))) ? 1u : 0u; }
    order[tid] = fc ? r : 0xFFFFFFFFu;
    __syncthreads();
    u32 start = 0;
    if (fc) for (int q = 0; q < 256; q++) start += (order[q] < r) ? freqs[q] : 0u;
    w.bstart[b * 256 + tid] = start;
    if (tid == 0) {
        u8* out = st.dst[b];
        u32 hdr = 0;
        for (int i = 0; i < 256; i++) {
            u32 f = freqs[i];
            for (int k = 0; k < 4 && f >= 128; k++) { out[hdr++] = (u8)(0x80 | f); f >>= 7; }
            out[hdr++] = (u8)f;
        }
        w.hdrLen[b] = hdr;
        st.ok[b] = 1;
        st.newLen[b] = hdr + n;
    }
}

__global__ __launch_bounds__(64) void k_srt_f_rank(XfStage st, SrtWs w)
{
    const int b = blockIdx.y;
    const u32 n = st.len[b];
    const u32 tbase = blockIdx.x * SRT_T;
    if (tbase >= n) return;
    const u32 hdr = w.hdrLen[b];
    if (hdr == 0) return;
    const int lane = lane_id();
    __shared__ u32 keys[256];
    __shared__ u32 listw[64];
    __shared__ u32 base[256];
    __shared__ u32 cntL[256];
    __shared__ u32 tile[SRT_T / 4];
    const u8* src = st.src[b] + tbase;
    const u32 cnt = (n - tbase < SRT_T) ? n - tbase : SRT_T;
    const size_t ot = ((size_t)b * w.perTiles + blockIdx.x) * 256, og = ((size_t)b * w.nSeg + blockIdx.x / w.segT) * 256;
    for (int c = lane; c < 256; c += 64) {
        const u32 lastBefore = max(w.tileLast[ot + c], w.segMax[og + c]);
        const u32 fo = w.firstOcc[b * 256 + c];
        // seen symbols by last occurrence (most recent first), then the ones still to come by first occurrence, absent ones last
        keys[c] = lastBefore ? (0x80000000u | lastBefore) : (fo == 0xFFFFFFFFu ? 0u : 0x7FFFFFFFu - fo);
        base[c] = hdr + w.bstart[b * 256 + c] + w.segSum[og + c] + w.tileCnt[ot + c];
        cntL[c] = 0;
    }
    {
        u8* tb = reinterpret_cast<u8*>(tile);
        for (u32 i = (u32)lane; i < cnt; i += 64) tb[i] = src[i];
    }
    __syncthreads();
    u8* listb = reinterpret_cast<u8*>(listw);
    for (int c = lane; c < 256; c += 64) {
        const u32 kc = keys[c];
        u32 r = 0;
        for (int q = 0; q < 256; q++) { const u32 kq = keys[q]; r += (kq > kc || (kq == kc && q < c)) ? 1u : 0u; }
        listb[r] = (u8)c;
    }
    __syncthreads();
    u32 lw = listw[lane];
    u32 front = (u32)__builtin_amdgcn_readfirstlane((int)lw) & 0xFF;
    u8* dst = st.dst[b];
    const u8* tb = reinterpret_cast<const u8*>(tile);
    u32 cur = 256, curCnt = 0;
    for (u32 k = 0; k < cnt; k++) {
        const u32 c = tb[k];                                   // (uniform)
        if (c != cur) {                                        // the bucket offset of the symbol at hand lives in a register
            if (cur < 256 && lane == 0) cntL[cur] = curCnt;
            KNZ_WAVE_ORDER();
            cur = c;
            curCnt = cntL[c];
        }
        if (c != front) {
            front = c;
            const u32 x = lw ^ (c * 0x01010101u);
            const u32 hz = (x - 0x01010101u) & ~x & 0x80808080u;
            const u64 m = __ballot(hz != 0);
            const int lane0 = __ffsll((long long)m) - 1;
            const u32 hz0 = (u32)__builtin_amdgcn_readlane((int)hz, lane0);
            const int byteIdx = (__ffs((int)hz0) - 1) >> 3;
            if (lane == 0) dst[base[c] + curCnt] = (u8)(4 * lane0 + byteIdx);
            lw = srt_to_front(lw, lane, lane0, byteIdx, c);
        }
        curCnt++;
    }
}

size_t srt_scratch_u32(int nBlocks, u32 maxLen)
{
    const size_t perTiles = ((size_t)maxLen + SRT_T - 1) / SRT_T;
    const u32 segT = srt_seg_tiles((u32)perTiles);
    const size_t nSeg = (perTiles + segT - 1) / segT;
    return (size_t)nBlocks * (2 * perTiles + 2 * nSeg + 2) * 256 + (size_t)nBlocks + 64;
}

void launch_srt_forward(hipStream_t s, const XfStage& st)
{
    SrtWs w;
    w.perTiles = (int)(((size_t)st.maxLen + SRT_T - 1) / SRT_T);
    if (w.perTiles < 1) w.perTiles = 1;
    w.segT = srt_seg_tiles((u32)w.perTiles);
    w.nSeg = ((u32)w.perTiles + w.segT - 1) / w.segT;
    u32* p = st.scratchU32;
    w.tileCnt = p; p += (size_t)st.nBlocks * w.perTiles * 256;
    w.tileLast = p; p += (size_t)st.nBlocks * w.perTiles * 256;
    w.segSum = p; p += (size_t)st.nBlocks * w.nSeg * 256;
    w.segMax = p; p += (size_t)st.nBlocks * w.nSeg * 256;
    w.firstOcc = p; p += (size_t)st.nBlocks * 256;
    w.bstart = p; p += (size_t)st.nBlocks * 256;
    w.hdrLen = p;
    const u32 per = (st.maxLen + 1024 + 4095) / 4096;
    { KScope ks_("k_srt_zero"); hipLaunchKernelGGL(k_srt_zero, dim3(per < 1024 ? per : 1024, st.nBlocks), dim3(256), 0, s, st); }
    hipMemsetAsync(w.firstOcc, 0xFF, sizeof(u32) * 256 * (size_t)st.nBlocks, s);
    const dim3 gridT((unsigned)w.perTiles, st.nBlocks);
    { KScope ks_("k_srt_f_tiles"); hipLaunchKernelGGL(k_srt_f_tiles, gridT, dim3(64), 0, s, st, w); }
    { KScope ks_("k_srt_f_scan"); hipLaunchKernelGGL(k_srt_f_scan1, dim3(w.nSeg, st.nBlocks), dim3(256), 0, s, st, w);
      hipLaunchKernelGGL(k_srt_f_scan2, dim3(st.nBlocks), dim3(256), 0, s, st, w); }
    { KScope ks_("k_srt_f_rank"); hipLaunchKernelGGL(k_srt_f_rank, gridT, dim3(64), 0, s, st, w); }
}

void launch_srt_inverse(hipStream_t s, const XfStage& st)
{
    KScope ks_("k_srt_inverse");
    hipLaunchKernelGGL(k_srt_inverse, dim3(st.nBlocks), dim3(64), 0, s, st);
}

}  // namespace knz
