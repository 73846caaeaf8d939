// SRT (sorted rank transform) on gfx950: one wave per block.
//
// Reference being replaced: transform/SRT.cpp:22-109 (forward), :111-204 (inverse), :206-244 (preprocess:
// symbols by descending frequency, ties by ascending value), :246-308 (header: 256 var-ints).
//
// What the transform is: a move-to-front over the RUNS of the input (initial list = symbols in order of first
// appearance), whose ranks are not written in place but appended to one bucket per symbol (buckets laid out in
// "preprocess" order); the bytes of a run after its head are zeros in the same bucket.
//
//   forward  histogram / first positions with LDS atomics; the body of the output is zero-filled by a separate
//            kernel and only non-zero ranks are scattered.  The wave walks the block 64 bytes per iteration,
//            finds run heads with a ballot, and for every head finds the symbol in the 256-entry list held in
//            registers (4 entries per lane, SWAR zero-byte test + ballot) and rotates it to the front.
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

__global__ __launch_bounds__(64) void k_srt_forward(XfStage st)
{
    const int b = blockIdx.x;
    const int lane = lane_id();
    const u32 length = st.len[b];
    if (lane == 0) { st.ok[b] = 0; st.newLen[b] = 0; }
    if (length == 0) { if (lane == 0) st.ok[b] = 1; return; }
    if (st.cap[b] < length + 1024) return;                     // SRT.hpp:38
    __shared__ u32 freqs[256];
    __shared__ u32 firstPos[256];
    __shared__ u32 bstart[256];
    __shared__ u32 cnt[256];
    __shared__ u8 symbols[256];
    __shared__ u32 listw[64];
    __shared__ u32 hdrLen;
    for (int i = lane; i < 256; i += 64) { freqs[i] = 0; firstPos[i] = 0xFFFFFFFFu; bstart[i] = 0; cnt[i] = 0; symbols[i] = 0; }
    listw[lane] = 0;
    __syncthreads();
    const u8* src = st.src[b];
    u8* out = st.dst[b];
    // ---- frequencies and first positions
    {
        const bool al = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
        const u32 n16 = al ? (length & ~15u) : 0;
        for (u32 i = 16u * (u32)lane; i < n16; i += 1024) {
            const uint4 v = *reinterpret_cast<const uint4*>(src + i);
            const u32 w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const u32 c = (w[k >> 2] >> (8 * (k & 3))) & 0xFF;
                atomicAdd(&freqs[c], 1u);
                atomicMin(&firstPos[c], i + (u32)k);
            }
        }
        for (u32 i = n16 + (u32)lane; i < length; i += 64) { const u32 c = src[i]; atomicAdd(&freqs[c], 1u); atomicMin(&firstPos[c], i); }
    }
    __syncthreads();
    // ---- bucket order, initial list (first-appearance order), header
    const u32 nbSymbols = srt_order(lane, freqs, symbols);
    {
        u8* listb = reinterpret_cast<u8*>(listw);
        for (int k = 0; k < 4; k++) {
            const int c = 4 * lane + k;
            if (freqs[c] == 0) continue;
            const u32 pc = firstPos[c];
            u32 r = 0;
            for (int q = 0; q < 256; q++) r += (firstPos[q] < pc) ? 1u : 0u;
            listb[r] = (u8)c;
        }
    }
    __syncthreads();
    if (lane == 0) {
        u32 pos = 0;
        for (u32 i = 0; i < nbSymbols; i++) { const u32 c = symbols[i]; bstart[c] = pos; pos += freqs[c]; }
        u32 hdr = 0;                                           // encodeHeader, SRT.cpp:246-277
        for (int i = 0; i < 256; i++) {
            u32 f = freqs[i];
            for (int k = 0; k < 4 && f >= 128; k++) { out[hdr++] = (u8)(0x80 | f); f >>= 7; }
            out[hdr++] = (u8)f;
        }
        hdrLen = hdr;
    }
    __syncthreads();
    u8* dst = out + hdrLen;
    u32 w = listw[lane];
    // ---- runs
    int curSym = -1;
    u32 curStart = 0;
    u32 carry = 0;
    u32 nextByte = ((u32)lane < length) ? (u32)src[lane] : 0u;
    for (u32 i0 = 0; i0 < length; i0 += 64) {
        const u32 byte = nextByte;
        const u32 in = i0 + 64 + (u32)lane;
        nextByte = (in < length) ? (u32)src[in] : 0u;          // next iteration's bytes are in flight while this one runs
        u32 prev = (u32)__shfl_up((int)byte, 1, 64);
        if (lane == 0) prev = carry;
        const bool head = (i0 + (u32)lane < length) && ((i0 + (u32)lane == 0) || byte != prev);
        u64 hm = __ballot(head);
        while (hm) {
            const int l = __ffsll((long long)hm) - 1;
            hm &= hm - 1;
            const u32 i = i0 + (u32)l;
            const u32 c = (u32)__builtin_amdgcn_readlane((int)byte, l);
            const u32 x = w ^ (c * 0x01010101u);
            const u32 hz = (x - 0x01010101u) & ~x & 0x80808080u;
            const u64 m = __ballot(hz != 0);
            const int lane0 = __ffsll((long long)m) - 1;
            const u32 hz0 = (u32)__builtin_amdgcn_readlane((int)hz, lane0);
            const int byteIdx = (__ffs((int)hz0) - 1) >> 3;
            const u32 r = (u32)(4 * lane0 + byteIdx);
            if (lane == 0) {
                if (curSym >= 0) cnt[curSym] += i - curStart;
                if (r) dst[bstart[c] + cnt[c]] = (u8)r;
            }
            if (r) w = srt_to_front(w, lane, lane0, byteIdx, c);
            curSym = (int)c;
            curStart = i;
        }
        carry = (u32)__builtin_amdgcn_readlane((int)byte, 63);
    }
    if (lane == 0) { st.ok[b] = 1; st.newLen[b] = hdrLen + length; }
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

void launch_srt_forward(hipStream_t s, const XfStage& st)
{
    const u32 per = (st.maxLen + 1024 + 4095) / 4096;
    { KScope ks_("k_srt_zero"); hipLaunchKernelGGL(k_srt_zero, dim3(per < 1024 ? per : 1024, st.nBlocks), dim3(256), 0, s, st); }
    { KScope ks_("k_srt_forward"); hipLaunchKernelGGL(k_srt_forward, dim3(st.nBlocks), dim3(64), 0, s, st); }
}

void launch_srt_inverse(hipStream_t s, const XfStage& st)
{
    KScope ks_("k_srt_inverse");
    hipLaunchKernelGGL(k_srt_inverse, dim3(st.nBlocks), dim3(64), 0, s, st);
}

}  // namespace knz
