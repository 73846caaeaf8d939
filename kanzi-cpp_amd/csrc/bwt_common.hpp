// Shared by the forward (bwt_fwd.hip) and inverse (bwt.hip) BWT block codec kernels.
#pragma once
#include "common.hpp"

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

// block that holds dense slot / position s: the last b with base[b] <= s (empty blocks repeat the base of their successor)
__device__ __forceinline__ int find_block(const u32* __restrict__ base, int nBlocks, u32 s)
{
    int lo = 0, hi = nBlocks;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (base[mid] <= s) lo = mid; else hi = mid; }
    return lo;
}

}  // namespace knz
