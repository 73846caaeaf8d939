// Random-access rates of the MI355X memory system as a function of the working set (developer probe, not product code).
// What the BWT kernels are bound by: 4-byte gathers / scatters into a table (ISA), dependent 8-byte record walks (inverse BWT).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/membench tools/membench.hip ;  gpurun -- tools/bin/membench
// With an argument "pmc" only one launch of each shape is made (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE calibration).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef unsigned int u32;
typedef unsigned long long u64;

__device__ __forceinline__ u32 mix(u32 x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// out[i] = tbl[rnd(i) mod W]  (W = words in the working set), 4 per thread in flight
template <int PER>
__global__ __launch_bounds__(256) void k_gather4(const u32* __restrict__ tbl, u32 W, u32 n, u32* __restrict__ out)
{
    const u32 i0 = (blockIdx.x * 256 + threadIdx.x);
    u32 v[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) { const u32 i = i0 + (u32)k * gridDim.x * 256u; v[k] = (i < n) ? tbl[(u64)mix(i) * W >> 32] : 0; }
#pragma unroll
    for (int k = 0; k < PER; k++) { const u32 i = i0 + (u32)k * gridDim.x * 256u; if (i < n) out[i] = v[k]; }
}

template <int PER>
__global__ __launch_bounds__(256) void k_scatter4(u32* __restrict__ tbl, u32 W, u32 n, const u32* __restrict__ in)
{
    const u32 i0 = (blockIdx.x * 256 + threadIdx.x);
#pragma unroll
    for (int k = 0; k < PER; k++) { const u32 i = i0 + (u32)k * gridDim.x * 256u; if (i < n) tbl[(u64)mix(i) * W >> 32] = in[i]; }
}

// windowed: consecutive groups of `win` accesses fall into one region of R words (locality like a block-by-block sweep)
__global__ __launch_bounds__(256) void k_scatter4_win(u32* __restrict__ tbl, u32 W, u32 R, u32 win, u32 n)
{
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 region = (u32)(((u64)(i / win) * R) % (W - R + 1));
    tbl[region + (u32)((u64)mix(i) * R >> 32)] = i;
}

// dependent walk: node = tbl[node] (u64 records, low 32 bits = next), `steps` hops per thread
__global__ __launch_bounds__(256) void k_walk8(const u64* __restrict__ tbl, u32 W, u32 nThreads, u32 steps, u32* __restrict__ out)
{
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nThreads) return;
    u32 node = (u32)((u64)mix(t) * W >> 32);
    u32 acc = 0;
    for (u32 s = 0; s < steps; s++) { const u64 r = tbl[node]; node = (u32)r; acc += (u32)(r >> 32); }
    out[t] = acc + node;
}
__global__ __launch_bounds__(256) void k_walk4(const u32* __restrict__ tbl, u32 W, u32 nThreads, u32 steps, u32* __restrict__ out)
{
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nThreads) return;
    u32 node = (u32)((u64)mix(t) * W >> 32);
    u32 acc = 0;
    for (u32 s = 0; s < steps; s++) { const u32 r = tbl[node]; node = r & 0x00FFFFFFu; if (node >= W) node -= W; acc += r >> 24; }
    out[t] = acc + node;
}

__global__ __launch_bounds__(256) void k_fill8(u64* __restrict__ tbl, u32 W)
{
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < W) tbl[i] = (u64)((u64)mix(i * 2654435761u + 12345u) * W >> 32) | ((u64)(i & 255) << 32);
}
__global__ __launch_bounds__(256) void k_fill4(u32* __restrict__ tbl, u32 W, u32 mod)
{
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < W) tbl[i] = (u32)((u64)mix(i * 2654435761u + 12345u) * mod >> 32) | ((i & 255u) << 24);
}
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n16)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class F>
static float timeit(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int r = 0; r < reps; r++) f();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv)
{
    const bool pmc = argc > 1 && !strcmp(argv[1], "pmc");
    const int reps = pmc ? 1 : 3;
    const u32 n = 212u << 20;                       // accesses per launch
    const size_t maxW = (size_t)1 << 28;            // 1 GiB of words
    u32 *tbl, *io; u64* tbl8;
    CK(hipMalloc(&tbl, maxW * 4)); CK(hipMalloc(&io, (size_t)n * 4)); CK(hipMalloc(&tbl8, maxW * 8));
    CK(hipMemset(tbl, 1, maxW * 4)); CK(hipMemset(io, 2, (size_t)n * 4));
    const u32 grid4 = (n / 4 + 255) / 256;
    printf("accesses per launch: %u\n", n);
    {
        const float ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, 0, (const uint4*)tbl, (uint4*)tbl8, (size_t)n * 4 / 16); }, reps);
        printf("copy %zu MB: %.3f ms = %.2f TB/s (r+w)\n", (size_t)n * 4 >> 20, ms, 2.0 * n * 4 / ms / 1e9);
    }
    const size_t sets[] = { 1u << 20, 8u << 20, 32u << 20, 64u << 20, 128u << 20, 192u << 20, 256u << 20, 512u << 20, 1024u << 20 };   // bytes
    for (size_t ws : sets) {
        const u32 W = (u32)(ws / 4);
        const float g = timeit([&] { hipLaunchKernelGGL(k_gather4<4>, dim3(grid4), dim3(256), 0, 0, tbl, W, n, io); }, reps);
        const float sc = timeit([&] { hipLaunchKernelGGL(k_scatter4<4>, dim3(grid4), dim3(256), 0, 0, tbl, W, n, io); }, reps);
        printf("set %5zu MB: gather4 %.3f ms = %.1f G/s | scatter4 %.3f ms = %.1f G/s\n", ws >> 20, g, n / g / 1e6, sc, n / sc / 1e6);
    }
    // scatter with block-sweep locality: the whole table is 848 MB, but 2M consecutive accesses fall into one 32 MB region
    for (u32 win : { 1u << 19, 1u << 21, 1u << 23 }) {
        const u32 W = (u32)(848u << 18), R = 8u << 20;
        const float sc = timeit([&] { hipLaunchKernelGGL(k_scatter4_win, dim3((n + 255) / 256), dim3(256), 0, 0, tbl, W, R, win, n); }, reps);
        printf("scatter4, 848 MB table, %u consecutive accesses per 32 MB region: %.3f ms = %.1f G/s\n", win, sc, n / sc / 1e6);
    }
    // dependent walks: n/64 threads x 64 hops, 8-byte and 4-byte records
    for (size_t ws : { (size_t)32 << 20, (size_t)64 << 20, (size_t)128 << 20, (size_t)256 << 20, (size_t)512 << 20, (size_t)1700 << 20 }) {
        const u32 W8 = (u32)(ws / 8);
        hipLaunchKernelGGL(k_fill8, dim3((W8 + 255) / 256), dim3(256), 0, 0, tbl8, W8);
        const u32 nT = n / 64;
        const float w8 = timeit([&] { hipLaunchKernelGGL(k_walk8, dim3((nT + 255) / 256), dim3(256), 0, 0, tbl8, W8, nT, 64u, io); }, reps);
        printf("walk8 set %5zu MB (%u records): %.3f ms = %.1f G hops/s\n", ws >> 20, W8, w8, (double)nT * 64 / w8 / 1e6);
    }
    for (size_t ws : { (size_t)32 << 20, (size_t)64 << 20 }) {
        const u32 W4 = (u32)(ws / 4);                // <= 2^24 records: link in 24 bits
        hipLaunchKernelGGL(k_fill4, dim3((W4 + 255) / 256), dim3(256), 0, 0, tbl, W4, W4);
        const u32 nT = n / 64;
        const float w4 = timeit([&] { hipLaunchKernelGGL(k_walk4, dim3((nT + 255) / 256), dim3(256), 0, 0, tbl, W4, nT, 64u, io); }, reps);
        printf("walk4 set %5zu MB (%u records): %.3f ms = %.1f G hops/s\n", ws >> 20, W4, w4, (double)nT * 64 / w4 / 1e6);
    }
    // thread count sensitivity of the walk (latency vs throughput): 848 MB of 8-byte records... 1700 MB set above; here fewer threads
    {
        const u32 W8 = (u32)(((size_t)1700 << 20) / 8);
        hipLaunchKernelGGL(k_fill8, dim3((W8 + 255) / 256), dim3(256), 0, 0, tbl8, W8);
        for (u32 nT : { 1u << 16, 1u << 18, 1u << 20, 1u << 22 }) {
            const float w8 = timeit([&] { hipLaunchKernelGGL(k_walk8, dim3((nT + 255) / 256), dim3(256), 0, 0, tbl8, W8, nT, 256u, io); }, reps);
            printf("walk8 1700 MB, %u threads x 256 hops: %.3f ms = %.1f G hops/s, %.0f ns per hop\n", nT, w8, (double)nT * 256 / w8 / 1e6, w8 * 1e6 / 256);
        }
    }
    return 0;
}
