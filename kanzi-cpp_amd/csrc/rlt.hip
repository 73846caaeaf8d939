// RLT (run length transform) on gfx950: one wave per block.
//
// Reference being replaced: transform/RLT.cpp:39-221 (forward), :223-245 (emitRunLength), :247-369 (inverse);
// Global.cpp:354-397 (data-type sniffing used for the escape choice).
//
// The reference is one serial loop whose exact behaviour (runs are extended 4 bytes at a time, cut at MAX_RUN - 4
// and 4 bytes before the end of the block, the first byte is emitted in the header) decides the bytes that come
// out.  The wave follows that loop token by token, but never byte by byte:
//   * the input is staged through a 2 KiB LDS window;
//   * a stretch of single literals (the common case in text) is found with one ballot over 64 positions and
//     copied by 64 lanes;
//   * a run is extended by emulating up to 64 of the reference's 4-byte compare steps at once (lane l looks at
//     srcIdx + 4l; the first lane whose step would not `continue` ends the emulation);
//   * only the token itself (<= 5 bytes) is written by one lane.
// The inverse scans 64 bytes per step for the next escape, copies literals and fills runs with the whole wave.
#include "common.hpp"
#include "stages.hpp"

namespace knz {

constexpr int RLT_ENC1 = 224;
constexpr int RLT_ENC2 = (255 - RLT_ENC1) << 8;
constexpr int RLT_THRESHOLD = 3;
constexpr int RLT_MAX_RUN = 0xFFFF + RLT_ENC2 + RLT_THRESHOLD - 1;
constexpr int RLT_MAX_RUN4 = RLT_MAX_RUN - 4;
constexpr int RLT_WIN = 2048;

__device__ int rlt_emit_run(u8* dst, int run, u8 escape, u8 val)
{
    dst[0] = val;
    dst[1] = 0;
    int dstIdx = (val == escape) ? 2 : 1;
    dst[dstIdx++] = escape;
    run -= RLT_THRESHOLD;
    if (run >= RLT_ENC1) {
        if (run < RLT_ENC2) { run -= RLT_ENC1; dst[dstIdx++] = (u8)(RLT_ENC1 + (run >> 8)); }
        else { run -= RLT_ENC2; dst[dstIdx++] = 0xFF; dst[dstIdx++] = (u8)(run >> 8); }
    }
    dst[dstIdx] = (u8)run;
    return dstIdx + 1;
}

// Global.cpp:354-397 (only the classes RLT cares about: DNA / BASE64 refuse the transform)
__device__ int rlt_refuses(int count, const u32* f)
{
    const char DNA[] = "acgntuACGNTU";
    const char NUM[] = "0123456789+-*/=,.:; ";
    const char B64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    int sum = 0;
    for (int i = 0; i < 12; i++) sum += (int)f[(u8)DNA[i]];
    if (sum > (count - count / 12)) return 1;                 // DNA
    sum = 0;
    for (int i = 0; i < 20; i++) sum += (int)f[(u8)NUM[i]];
    if (sum == count) return 0;                               // NUMERIC
    sum = (f[0x3D] == 1) ? 1 : 0;
    for (int i = 0; i < 64; i++) sum += (int)f[(u8)B64[i]];
    if (sum == count) return 1;                               // BASE64
    return 0;
}

// LDS window over a byte array: bytes [base, base + RLT_WIN) (zero past `count`)
struct RltWindow {
    u8* win;
    const u8* src;
    int count;
    int base;
    __device__ void refill(int from, int lane)
    {
        base = from < 0 ? 0 : (from & ~15);
        __syncthreads();
        const int o = 32 * lane;
        if (base + o + 32 <= count && ((reinterpret_cast<uintptr_t>(src + base) & 15) == 0)) {
            const uint4* p = reinterpret_cast<const uint4*>(src + base + o);
            reinterpret_cast<uint4*>(win + o)[0] = p[0];
            reinterpret_cast<uint4*>(win + o)[1] = p[1];
        } else {
            for (int k = 0; k < 32; k++) win[o + k] = (base + o + k < count) ? src[base + o + k] : (u8)0;
        }
        __syncthreads();
    }
    __device__ __forceinline__ bool covers(int lo, int hi) const { return lo >= base && hi <= base + RLT_WIN; }
    __device__ __forceinline__ u32 at(int pos) const { return win[pos - base]; }
    __device__ __forceinline__ u32 word(int pos) const
    {
        const u8* p = win + (pos - base);
        return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24);
    }
};

__global__ __launch_bounds__(64) void k_rlt_forward(XfStage st)
{
    const int b = blockIdx.x;
    const int lane = lane_id();
    const int count = (int)st.len[b];
    if (lane == 0) { st.ok[b] = 0; st.newLen[b] = 0; }
    if (count == 0) { if (lane == 0) st.ok[b] = 1; return; }
    if (count < 16) return;
    const int maxEnc = (count <= 512) ? count + 32 : count;
    if ((int)st.cap[b] < maxEnc) return;
    const u8* src = st.src[b];
    u8* dst = st.dst[b];
    const int etype = st.entropyType;
    const bool findBestEscape = !(etype == KNZ_E_NONE || etype == KNZ_E_ANS0 || etype == KNZ_E_HUFFMAN || etype == 4);
    __shared__ u32 freqs[256];
    __shared__ int shEscape;
    __shared__ __attribute__((aligned(16))) u8 winBuf[RLT_WIN + 32];
    u32 escape = 0xFB;
    if (findBestEscape) {
        for (int i = lane; i < 256; i += 64) freqs[i] = 0;
        __syncthreads();
        for (int i = lane; i < count; i += 64) atomicAdd(&freqs[src[i]], 1u);
        __syncthreads();
        if (lane == 0) {
            int e = -1;
            if (!rlt_refuses(count, freqs)) {
                int minIdx = 0;
                if (freqs[minIdx] > 0) {
                    for (int i = 1; i < 256; i++) {
                        if (freqs[i] < freqs[minIdx]) { minIdx = i; if (freqs[i] == 0) break; }
                    }
                }
                e = minIdx;
            }
            shEscape = e;
        }
        __syncthreads();
        if (shEscape < 0) return;
        escape = (u32)shEscape;
    }
    RltWindow W;
    W.win = winBuf; W.src = src; W.count = count; W.base = 0;
    W.refill(0, lane);
    const int srcEnd = count, srcEnd4 = srcEnd - 4, dstEnd = (int)st.cap[b];
    int srcIdx = 1, dstIdx = 0;
    int res = 1, run = 0;
    u32 prev = W.at(0);
    if (lane == 0) { dst[0] = (u8)escape; dst[1] = (u8)prev; if (prev == escape) dst[2] = 0; }
    dstIdx = (prev == escape) ? 3 : 2;
    while (true) {
        if (!W.covers(srcIdx - 1, srcIdx + 272)) W.refill(srcIdx - 1, lane);
        const u32 nb = W.at(srcIdx);
        if (run == 1 && prev != escape && nb != prev) {
            // ---- single literals: position p is emitted and p + 1 becomes `prev` while the loop would go on
            const int p = srcIdx - 1 + lane;
            const u32 cur = W.at(p), nxt = W.at(p + 1);
            const bool okl = (cur != escape) && (cur != nxt) && (p + 2 < srcEnd4);
            const u64 m = __ballot(!okl);
            const int k = m ? (__ffsll((long long)m) - 1) : 64;
            if (k > 0) {
                if (dstIdx + k >= dstEnd) { res = 0; break; }
                if (lane < k) dst[dstIdx + lane] = (u8)cur;
                dstIdx += k;
                srcIdx += k;
                prev = W.at(srcIdx - 1);
                continue;
            }
        }
        if (prev == nb) {
            // ---- run extension: lane l plays the reference's l-th 4-byte step from here
            const u32 v4 = 0x01010101u * prev;
            const int pos = srcIdx + 4 * lane;
            const u32 w = W.word(pos);
            const bool cont = (w == v4) && (run + 4 * (lane + 1) < RLT_MAX_RUN4) && (pos + 4 < srcEnd4);
            const u64 m = __ballot(!cont);
            if (m == 0) { srcIdx += 256; run += 256; continue; }
            const int l = __ffsll((long long)m) - 1;
            const u32 wl = (u32)__builtin_amdgcn_readlane((int)w, l);
            const u32 diff = wl ^ v4;
            const int n = (diff == 0) ? 4 : ((__ffs((int)diff) - 1) >> 3);
            srcIdx += 4 * l + n;
            run += 4 * l + n;
        }
        // ---- emit (RLT.cpp:150-175)
        if (run > RLT_THRESHOLD) {
            if (dstIdx + 6 >= dstEnd) { res = 0; break; }
            int len = 0;
            if (lane == 0) len = rlt_emit_run(&dst[dstIdx], run, (u8)escape, (u8)prev);
            dstIdx += __builtin_amdgcn_readfirstlane(len);
        } else if (prev != escape) {
            if (dstIdx + run >= dstEnd) { res = 0; break; }
            if (lane < run) dst[dstIdx + lane] = (u8)prev;
            dstIdx += run;
        } else {
            if (dstIdx + (2 * run) >= dstEnd) { res = 0; break; }
            if (lane < 2 * run) dst[dstIdx + lane] = (lane & 1) ? (u8)0 : (u8)escape;
            dstIdx += 2 * run;
        }
        if (!W.covers(srcIdx, srcIdx + 1)) W.refill(srcIdx - 1, lane);
        prev = W.at(srcIdx);
        srcIdx++;
        run = 1;
        if (srcIdx >= srcEnd4) break;
    }
    if (lane == 0) {
        // ---- tail (RLT.cpp:178-205): at most a few bytes, one lane
        if (res) {
            if (prev != escape) {
                if (dstIdx + run < dstEnd) while (run-- > 0) dst[dstIdx++] = (u8)prev;
            } else {
                if (dstIdx + (2 * run) < dstEnd) while (run-- > 0) { dst[dstIdx++] = (u8)escape; dst[dstIdx++] = 0; }
            }
            while ((srcIdx < srcEnd) && (dstIdx < dstEnd)) {
                if (src[srcIdx] == escape) {
                    if (dstIdx + 2 >= dstEnd) { res = 0; break; }
                    dst[dstIdx++] = (u8)escape;
                    dst[dstIdx++] = 0;
                    srcIdx++;
                    continue;
                }
                dst[dstIdx++] = src[srcIdx++];
            }
            res &= (srcIdx == srcEnd) ? 1 : 0;
        }
        st.ok[b] = (res && (dstIdx < srcIdx)) ? 1 : 0;
        st.newLen[b] = (u32)dstIdx;
    }
}

__global__ __launch_bounds__(64) void k_rlt_inverse(XfStage st)
{
    const int b = blockIdx.x;
    const int lane = lane_id();
    const int count = (int)st.len[b];
    if (lane == 0) { st.ok[b] = 0; st.newLen[b] = 0; }
    if (count == 0) { if (lane == 0) st.ok[b] = 1; return; }
    const u8* src = st.src[b];
    u8* dst = st.dst[b];
    __shared__ __attribute__((aligned(16))) u8 winBuf[RLT_WIN + 32];
    RltWindow W;
    W.win = winBuf; W.src = src; W.count = count; W.base = 0;
    W.refill(0, lane);
    int srcIdx = 0, dstIdx = 0;
    const int srcEnd = count, dstEnd = (int)st.cap[b];
    int res = 1;
    const u32 escape = W.at(srcIdx++);
    u32 last = 0;                                        // dst[dstIdx - 1]
    if ((srcIdx < srcEnd) && (W.at(srcIdx) == escape)) {
        srcIdx++;
        if ((srcIdx < srcEnd) && (W.at(srcIdx) != 0)) return;
        if (dstIdx >= dstEnd) return;
        if (lane == 0) dst[dstIdx] = (u8)escape;
        dstIdx++;
        last = escape;
        srcIdx++;
    }
    while (srcIdx < srcEnd) {
        if (!W.covers(srcIdx, srcIdx + 80)) W.refill(srcIdx, lane);
        // ---- literal span: up to 64 bytes towards the next escape
        const int pos = srcIdx + lane;
        const u32 cur = W.at(pos);
        const bool isEsc = (pos < srcEnd) && (cur == escape);
        const u64 m = __ballot(isEsc);
        const int room = srcEnd - srcIdx;
        int lit = m ? (__ffsll((long long)m) - 1) : (room < 64 ? room : 64);
        if (lit > 0) {
            if (lit > dstEnd - dstIdx) { res = 0; break; }
            if (lane < lit) dst[dstIdx + lane] = (u8)cur;
            last = W.at(srcIdx + lit - 1);
            srcIdx += lit;
            dstIdx += lit;
        }
        if (m == 0 || srcIdx >= srcEnd) continue;
        // ---- token (RLT.cpp:308-355)
        srcIdx++;
        if (srcIdx >= srcEnd) { res = 0; break; }
        int run = (int)W.at(srcIdx++);
        if (run == 0) {
            if (dstIdx >= dstEnd) { res = 0; break; }
            if (lane == 0) dst[dstIdx] = (u8)escape;
            dstIdx++;
            last = escape;
            continue;
        }
        if (run == 0xFF) {
            if (srcIdx + 1 >= srcEnd) { res = 0; break; }
            run = ((int)W.at(srcIdx) << 8) | (int)W.at(srcIdx + 1);
            srcIdx += 2;
            run += RLT_ENC2;
        } else if (run >= RLT_ENC1) {
            if (srcIdx >= srcEnd) { res = 0; break; }
            run = ((run - RLT_ENC1) << 8) | (int)W.at(srcIdx);
            srcIdx++;
            run += RLT_ENC1;
        }
        run += (RLT_THRESHOLD - 1);
        if ((dstIdx + run > dstEnd) || (run > RLT_MAX_RUN)) { res = 0; break; }
        if (dstIdx == 0) { res = 0; break; }
        for (int k = lane; k < run; k += 64) dst[dstIdx + k] = (u8)last;
        dstIdx += run;
    }
    if (lane == 0) {
        st.ok[b] = (res && (srcIdx == srcEnd)) ? 1 : 0;
        st.newLen[b] = (u32)dstIdx;
    }
}

void launch_rlt_forward(hipStream_t s, const XfStage& st) { KScope ks_("k_rlt_forward"); hipLaunchKernelGGL(k_rlt_forward, dim3(st.nBlocks), dim3(64), 0, s, st); }
void launch_rlt_inverse(hipStream_t s, const XfStage& st) { KScope ks_("k_rlt_inverse"); hipLaunchKernelGGL(k_rlt_inverse, dim3(st.nBlocks), dim3(64), 0, s, st); }

}  // namespace knz
