// Block checksums (-x32 / -x64): XXHash32 / XXHash64 as kanzi computes them (util/XXHash.hpp:61-115,
// :153-230; seed = 0x4B414E5A, io/CompressedOutputStream.cpp:104,109). kanzi's 64-bit variant merges
// the four accumulators with 32-bit style shifts ((v << 1) | (v >> 31) ...), reproduced literally.
//
// The hash is a serial recurrence per accumulator; the only parallelism inside one block is the four
// accumulators, so a block is hashed by 4 lanes (stripe 16 B / 32 B) and blocks run side by side.
// Off in every BASELINE config; latency-bound when enabled.
#include "common.hpp"
#include "stages.hpp"

namespace knz {

constexpr u32 P32_1 = 2654435761u, P32_2 = 2246822519u, P32_3 = 3266489917u, P32_4 = 668265263u, P32_5 = 374761393u;
constexpr u64 P64_1 = 0x9E3779B185EBCA87ull, P64_2 = 0xC2B2AE3D27D4EB4Full, P64_3 = 0x165667B19E3779F9ull,
              P64_4 = 0x85EBCA77C2B2AE63ull, P64_5 = 0x27D4EB2F165667C5ull;
constexpr u32 XX_SEED = 0x4B414E5Au;

__device__ __forceinline__ u32 ld32(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); }
__device__ __forceinline__ u64 ld64(const u8* p) { return (u64)ld32(p) | ((u64)ld32(p + 4) << 32); }
__device__ __forceinline__ u64 xx64_round(u64 acc, u64 val) { acc += val * P64_2; return ((acc << 31) | (acc >> 33)) * P64_1; }

// 16 blocks per wave, 4 lanes per block
__global__ __launch_bounds__(64) void k_xxhash(const u8* const* __restrict__ ptr, const u32* __restrict__ lens, int nBlocks, int bits,
                                               u64* __restrict__ out)
{
    const int lane = lane_id();
    const int b = blockIdx.x * 16 + (lane >> 2);
    const int j = lane & 3;
    const bool act = b < nBlocks;
    const u8* data = act ? ptr[b] : nullptr;
    const int length = act ? (int)lens[b] : 0;
    const bool al = act && ((reinterpret_cast<uintptr_t>(data) & 7) == 0);
    if (bits == 32) {
        u32 v = (j == 0) ? XX_SEED + P32_1 + P32_2 : (j == 1) ? XX_SEED + P32_2 : (j == 2) ? XX_SEED : XX_SEED - P32_1;
        int idx = 0;
        if (length >= 16) {
            const int end16 = length - 16;
            do {
                const u32 x = al ? *reinterpret_cast<const u32*>(data + idx + 4 * j) : ld32(data + idx + 4 * j);
                v += x * P32_2;
                v = ((v << 13) | (v >> 19)) * P32_1;
                idx += 16;
            } while (idx <= end16);
        }
        const int base = lane & ~3;
        const u32 v1 = (u32)__shfl((int)v, base, 64), v2 = (u32)__shfl((int)v, base + 1, 64);
        const u32 v3 = (u32)__shfl((int)v, base + 2, 64), v4 = (u32)__shfl((int)v, base + 3, 64);
        if (act && j == 0) {
            u32 h32;
            if (length >= 16) h32 = ((v1 << 1) | (v1 >> 31)) + ((v2 << 7) | (v2 >> 25)) + ((v3 << 12) | (v3 >> 20)) + ((v4 << 18) | (v4 >> 14));
            else h32 = XX_SEED + P32_5;
            h32 += (u32)length;
            while (idx <= length - 4) { h32 += ld32(data + idx) * P32_3; h32 = ((h32 << 17) | (h32 >> 15)) * P32_4; idx += 4; }
            while (idx < length) { h32 += (u32)data[idx] * P32_5; h32 = ((h32 << 11) | (h32 >> 21)) * P32_1; idx++; }
            h32 ^= h32 >> 15; h32 *= P32_2; h32 ^= h32 >> 13; h32 *= P32_3;
            out[b] = (u64)(h32 ^ (h32 >> 16));
        }
    } else {
        const u64 seed = (u64)(int64_t)(int32_t)XX_SEED;      // XXHash64(int64 seed) constructed from the int BITSTREAM_TYPE
        u64 v = (j == 0) ? seed + P64_1 + P64_2 : (j == 1) ? seed + P64_2 : (j == 2) ? seed : seed - P64_1;
        int idx = 0;
        if (length >= 32) {
            const int length32 = length - 32;
            do {
                const u64 x = al ? *reinterpret_cast<const u64*>(data + idx + 8 * j) : ld64(data + idx + 8 * j);
                v = xx64_round(v, x);
                idx += 32;
            } while (idx <= length32);
        }
        const int base = lane & ~3;
        const u64 v1 = (u64)__shfl((long long)v, base, 64), v2 = (u64)__shfl((long long)v, base + 1, 64);
        const u64 v3 = (u64)__shfl((long long)v, base + 2, 64), v4 = (u64)__shfl((long long)v, base + 3, 64);
        if (act && j == 0) {
            u64 h64;
            if (length >= 32) {
                h64 = ((v1 << 1) | (v1 >> 31)) + ((v2 << 7) | (v2 >> 25)) + ((v3 << 12) | (v3 >> 20)) + ((v4 << 18) | (v4 >> 14));
                h64 = (h64 ^ xx64_round(0, v1)) * P64_1 + P64_4;
                h64 = (h64 ^ xx64_round(0, v2)) * P64_1 + P64_4;
                h64 = (h64 ^ xx64_round(0, v3)) * P64_1 + P64_4;
                h64 = (h64 ^ xx64_round(0, v4)) * P64_1 + P64_4;
            } else h64 = seed + P64_5;
            h64 += (u64)(int64_t)length;
            while (idx + 8 <= length) { h64 ^= xx64_round(0, ld64(data + idx)); h64 = ((h64 << 27) | (h64 >> 37)) * P64_1 + P64_4; idx += 8; }
            while (idx + 4 <= length) { h64 ^= (u64)ld32(data + idx) * P64_1; h64 = ((h64 << 23) | (h64 >> 41)) * P64_2 + P64_3; idx += 4; }
            while (idx < length) { h64 ^= (u64)data[idx] * P64_5; h64 = ((h64 << 11) | (h64 >> 53)) * P64_1; idx++; }
            h64 ^= h64 >> 33; h64 *= P64_2; h64 ^= h64 >> 29; h64 *= P64_3;
            out[b] = h64 ^ (h64 >> 32);
        }
    }
}

__global__ void k_block_ptrs(const u8* base, u64 stride, int nBlocks, const u8** ptr)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nBlocks) ptr[b] = base + (size_t)b * stride;
}

// decode side: compare with the checksum stored in the stream (io/CompressedInputStream.cpp:1003-1022)
__global__ void k_verify_checksums(DecBlock* blocks, int nBlocks, int bits, const u64* sums)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nBlocks) return;
    DecBlock& db = blocks[b];
    if (db.error) return;
    const u64 a = bits == 32 ? (sums[b] & 0xFFFFFFFFull) : sums[b];
    const u64 c = bits == 32 ? (db.checksum & 0xFFFFFFFFull) : db.checksum;
    if (a != c) db.error = KNZ_ERR_CRC_CHECK;
}

__global__ void k_declen(const DecBlock* blocks, int nBlocks, u32* lens)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nBlocks) lens[b] = blocks[b].error ? 0u : blocks[b].preLen;
}

void launch_xxhash(hipStream_t s, const u8* const* ptr, const u32* lens, int nBlocks, int bits, u64* out)
{
    { KScope ks_("k_xxhash"); hipLaunchKernelGGL(k_xxhash, dim3((nBlocks + 15) / 16), dim3(64), 0, s, ptr, lens, nBlocks, bits, out); }
}

void launch_block_ptrs(hipStream_t s, const u8* base, u64 stride, int nBlocks, const u8** ptr)
{
    { KScope ks_("k_block_ptrs"); hipLaunchKernelGGL(k_block_ptrs, dim3((nBlocks + 255) / 256), dim3(256), 0, s, base, stride, nBlocks, ptr); }
}

void launch_verify_checksums(hipStream_t s, DecBlock* blocks, int nBlocks, int bits, const u8* out, u64 outStride, const u8** ptrScratch,
                             u32* lenScratch, u64* sumScratch)
{
    launch_block_ptrs(s, out, outStride, nBlocks, ptrScratch);
    { KScope ks_("k_declen"); hipLaunchKernelGGL(k_declen, dim3((nBlocks + 255) / 256), dim3(256), 0, s, blocks, nBlocks, lenScratch); }
    launch_xxhash(s, ptrScratch, lenScratch, nBlocks, bits, sumScratch);
    { KScope ks_("k_verify_checksums"); hipLaunchKernelGGL(k_verify_checksums, dim3((nBlocks + 255) / 256), dim3(256), 0, s, blocks, nBlocks, bits, sumScratch); }
}

}  // namespace knz
