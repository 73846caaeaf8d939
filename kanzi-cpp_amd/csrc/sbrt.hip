// SBRT in its RANK and TIMESTAMP modes (and MTF, for cross-checks) on gfx950: one wave per block.
//
// Reference being replaced: transform/SBRT.cpp:46-97 (forward), :99-145 (inverse), mode masks :26-31.
// The 256 symbols are kept ordered by a key q[s], larger first, a newcomer ahead of equal keys:
//   MTF        q = i                       (always to the front; the dedicated kernels in mtft.hip are used for it)
//   RANK       q = (i + previous position of the symbol) >> 1
//   TIMESTAMP  q = previous position of the symbol
// Every byte moves one symbol up the list by an amount that depends on all earlier bytes: one dependent chain per
// block.  The wave keeps the list (symbol, key) and the last position of every symbol in registers, 4 entries per
// lane, so a step has no memory access at all: the rank of a symbol is a ballot over four compares, the new rank is
// the number of larger keys ahead of the old one (four ballots + popcounts; the list is ordered, so nothing else can
// be in the way), and the entries in between slide down with one DPP lane shift.  Input and output move 64 bytes
// at a time, coalesced.
#include "common.hpp"
#include "stages.hpp"

namespace knz {

__device__ __forceinline__ u32 pick4(const u32 (&a)[4], u32 k)     // k uniform
{
    return k == 0 ? a[0] : (k == 1 ? a[1] : (k == 2 ? a[2] : a[3]));
}

template <bool INV>
__global__ __launch_bounds__(64) void k_sbrt(XfStage st, int mode)
{
    const int b = blockIdx.x;
    const int lane = lane_id();
    const int n = (int)st.len[b];
    if (n == 0) return;
    if ((u32)n > st.cap[b]) { if (lane == 0) { st.ok[b] = 0; st.newLen[b] = 0; } return; }
    const u8* __restrict__ src = st.src[b];
    u8* __restrict__ dst = st.dst[b];
    const bool useTime = mode != 3, usePrev = mode != 1;
    const int shift = (mode == 2) ? 1 : 0;
    u32 sym[4], key[4], last[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { sym[k] = 4u * (u32)lane + (u32)k; key[k] = 0; last[k] = 0; }
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int m = (n - i0 < 64) ? n - i0 : 64;
        const u32 inb = (lane < m) ? (u32)src[i0 + lane] : 0u;
        u32 outb = 0;
        for (int t = 0; t < m; t++) {
            const u32 i = (u32)(i0 + t);
            const u32 v = (u32)__builtin_amdgcn_readlane((int)inb, t);
            u32 r, c;
            if (!INV) {
                c = v;
                const u32 hitk = (sym[0] == c) ? 1u : (sym[1] == c) ? 2u : (sym[2] == c) ? 3u : (sym[3] == c) ? 4u : 0u;
                const u64 bal = __ballot(hitk != 0);
                const int fl = __ffsll((long long)bal) - 1;
                r = 4u * (u32)fl + (u32)__builtin_amdgcn_readlane((int)hitk, fl) - 1u;
            } else {
                r = v;
                c = (u32)__builtin_amdgcn_readlane((int)pick4(sym, r & 3), (int)(r >> 2));
            }
            const u32 pc = (u32)__builtin_amdgcn_readlane((int)pick4(last, c & 3), (int)(c >> 2));
            const u32 qc = ((useTime ? i : 0u) + (usePrev ? pc : 0u)) >> shift;
            if ((u32)lane == (c >> 2)) {
#pragma unroll
                for (int k = 0; k < 4; k++) if ((c & 3) == (u32)k) last[k] = i;
            }
            // new rank = number of entries ahead of r whose key is larger than qc
            u32 rn = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) rn += (u32)__popcll(__ballot((4u * (u32)lane + (u32)k) < r && key[k] > qc));
            // entries rn .. r-1 move one position down, (c, qc) goes to rn
            const u32 upSym = (u32)__builtin_amdgcn_update_dpp(0, (int)sym[3], 0x138, 0xF, 0xF, false);   // wave_shr:1
            const u32 upKey = (u32)__builtin_amdgcn_update_dpp(0, (int)key[3], 0x138, 0xF, 0xF, false);
            u32 ns[4], nk[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const u32 j = 4u * (u32)lane + (u32)k;
                const u32 fromS = k ? sym[k ? k - 1 : 0] : upSym;
                const u32 fromK = k ? key[k ? k - 1 : 0] : upKey;
                const bool slide = j > rn && j <= r;
                ns[k] = (j == rn) ? c : (slide ? fromS : sym[k]);
                nk[k] = (j == rn) ? qc : (slide ? fromK : key[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) { sym[k] = ns[k]; key[k] = nk[k]; }
            if (lane == t) outb = INV ? c : r;
        }
        if (lane < m) dst[i0 + lane] = (u8)outb;
    }
    if (lane == 0) { st.ok[b] = 1; st.newLen[b] = (u32)n; }
}

void launch_sbrt_forward(hipStream_t s, const XfStage& st, int mode) { KScope ks_("k_sbrt_forward"); hipLaunchKernelGGL((k_sbrt<false>), dim3(st.nBlocks), dim3(64), 0, s, st, mode); }
void launch_sbrt_inverse(hipStream_t s, const XfStage& st, int mode) { KScope ks_("k_sbrt_inverse"); hipLaunchKernelGGL((k_sbrt<true>), dim3(st.nBlocks), dim3(64), 0, s, st, mode); }

}  // namespace knz
