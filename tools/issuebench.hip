// Developer probe (not part of the product): what one lone wave pays per instruction on gfx950, for the block-serial chains
// (FPAQ, SRT inverse, LZ). Every pattern is REPT copies of a short instruction group inside a loop of ITER trips, timed with
// s_memtime (shader clock) and with the 100 MHz wall clock, so the shader frequency a mostly idle chip runs such a kernel at
// comes out as well.
//   hipcc --offload-arch=gfx950 -O3 tools/issuebench.hip -o tools/bin/issuebench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
typedef unsigned u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); fflush(stdout); exit(1); } } while (0)

constexpr int ITER = 1000;
#define STR2(x) #x
#define STR(x) STR2(x)
#define REPT 32

template <int MODE>
__global__ __launch_bounds__(64) void k_issue(u64* out, u32* sink)
{
    __shared__ u32 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (u32)((i * 4 + 64) & 4095);
    __syncthreads();
    u32 v = threadIdx.x, w = threadIdx.x * 3 + 1, s = 1, a = (threadIdx.x * 4) & 4095;
    u64 big = 0x123456789ull + threadIdx.x;
    const u64 t0 = __builtin_readcyclecounter();
    const u64 w0 = wall_clock64();
    for (int it = 0; it < ITER; it++) {
        if (MODE == 0) asm volatile(".rept " STR(REPT) "\n s_add_u32 %0, %0, 3\n .endr" : "+s"(s) : : "scc");
        if (MODE == 1) asm volatile(".rept " STR(REPT) "\n v_add_u32 %0, %0, %0\n .endr" : "+v"(v));
        if (MODE == 2) asm volatile(".rept " STR(REPT) "\n v_add_u32 %0, 1, %0\n v_add_u32 %1, 1, %1\n .endr" : "+v"(v), "+v"(w));
        if (MODE == 3) asm volatile(".rept " STR(REPT) "\n s_add_u32 %0, %0, 3\n v_add_u32 %1, 1, %1\n .endr" : "+s"(s), "+v"(v) : : "scc");
        if (MODE == 4) asm volatile(".rept " STR(REPT) "\n v_readfirstlane_b32 %0, %1\n s_add_u32 %0, %0, 1\n s_nop 0\n v_mov_b32 %1, %0\n .endr" : "+s"(s), "+v"(v) : : "scc");
        if (MODE == 5) asm volatile(".rept " STR(REPT) "\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n .endr" : "+v"(a));
        if (MODE == 6) asm volatile(".rept " STR(REPT) "\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n .endr" : "+s"(s) : : "scc");       // taken, skips one instruction
        if (MODE == 7) asm volatile(".rept " STR(REPT) "\n s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n 1:\n .endr" : "+s"(s) : : "scc");                  // not taken
        if (MODE == 8) asm volatile(".rept " STR(REPT) "\n v_cmp_lt_u64 vcc, %0, %1\n s_cbranch_vccnz 1f\n s_nop 0\n 1:\n .endr" : : "v"(big), "v"(big + 5) : "vcc");   // vector compare, then a taken branch on it
        if (MODE == 9) asm volatile(".rept " STR(REPT) "\n s_nop 1\n v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_alignbit_b32 %1, %0, %1, 8\n .endr" : "+v"(v), "+v"(w));
        if (MODE == 10) asm volatile(".rept " STR(REPT) "\n v_readlane_b32 %0, %1, 1\n s_add_u32 %0, %0, 8\n v_writelane_b32 %1, %0, 1\n .endr" : "+s"(s), "+v"(v) : : "scc");
        if (MODE == 11) asm volatile(".rept " STR(REPT) "\n s_mul_hi_u32 %0, %0, %0\n s_lshr_b32 %0, %0, 1\n .endr" : "+s"(s));
    }
    const u64 t1 = __builtin_readcyclecounter();
    const u64 w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; }
    sink[threadIdx.x] = v + w + s + a + (u32)big;
}

template <int MODE>
static void run(const char* what, int groupLen, u64* dOut, u32* dSink)
{
    u64 h[2] = { 0, 0 };
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(k_issue<MODE>, dim3(1), dim3(64), 0, 0, dOut, dSink);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, dOut, 16, hipMemcpyDeviceToHost));
    }
    const double groups = (double)ITER * REPT;
    printf("%-62s %7.2f shader cycles per group of %d = %5.2f per instruction; %6.2f ns per group (shader clock %.0f MHz)\n", what, h[0] / groups, groupLen,
           h[0] / groups / groupLen, h[1] * 10.0 / groups, h[0] / (h[1] * 10.0) * 1e3);
    fflush(stdout);
}

int main()
{
    u64* dOut; u32* dSink;
    CK(hipMalloc(&dOut, 64)); CK(hipMalloc(&dSink, 256));
    printf("one wave, %d x %d groups per pattern\n", ITER, REPT); fflush(stdout);
    run<0>("dependent s_add", 1, dOut, dSink);
    run<1>("dependent v_add", 1, dOut, dSink);
    run<2>("two independent v_add chains", 2, dOut, dSink);
    run<3>("s_add + v_add, independent", 2, dOut, dSink);
    run<4>("v_readfirstlane, s_add, s_nop, v_mov (scalar/vector round trip)", 4, dOut, dSink);
    run<5>("dependent ds_read_b32 + waitcnt", 2, dOut, dSink);
    run<6>("s_cmp + branch taken over one instruction", 2, dOut, dSink);
    run<7>("s_cmp + branch not taken", 2, dOut, dSink);
    run<8>("v_cmp_lt_u64 + s_cbranch_vccnz taken over one instruction", 2, dOut, dSink);
    run<9>("s_nop 1, DPP wave_shl, alignbit (list shift)", 3, dOut, dSink);
    run<10>("v_readlane, s_add, v_writelane (entry bump)", 3, dOut, dSink);
    run<11>("s_mul_hi_u32 + s_lshr dependent", 2, dOut, dSink);
    return 0;
}
