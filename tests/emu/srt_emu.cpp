// CPU-only check of the kernels of kanzi-cpp_amd/csrc/srt.hip (transform id 13) against the oracle, forward and inverse; see xf_harness.hpp.
#define KNZ_EMU 1
#include "hip/hip_runtime.h"
#include "../../kanzi-cpp_amd/csrc/srt.hip"
#define XF_TTYPE 13
#define XF_FWD(st) launch_srt_forward(nullptr,st)
#define XF_INV(st) launch_srt_inverse(nullptr,st)
#define XF_SCRATCH_U32(nb, ml) (knz::srt_scratch_u32(nb, ml) > knz::srt_inverse_scratch_u32(nb, ml) ? knz::srt_scratch_u32(nb, ml) : knz::srt_inverse_scratch_u32(nb, ml))
// damaged bodies (the header stays: 256 var-ints): a first bucket byte that makes the initial list no permutation, ranks beyond the
// live part of the list, ranks where zeros were and zeros where ranks were -- the fast chain has to hand these to the general one
static void srt_malform(uint8_t* p, int len, int variant, int block)
{
    int at = 0;
    for (int i = 0; i < 256 && at < len; i++) { while (at < len && (p[at] & 0x80)) at++; at++; }
    if (at >= len) return;
    uint8_t* body = p + at;
    const int n = len - at;
    uint32_t x = 12345u + 977u * (uint32_t)variant + 31u * (uint32_t)block;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 8; };
    switch (variant) {
    case 0: body[0] = (uint8_t)(body[0] + 1); break;                                     // two symbols claim one rank, or one out of range
    case 1: body[0] = 200; break;
    case 2: for (int k = 0; k < 3; k++) { const int i = 1 + (int)(rnd() % (uint32_t)(n > 1 ? n - 1 : 1)); if (i < n && body[i]) body[i] = 255; } break;
    case 3: for (int k = 0; k < 8; k++) { const int i = 1 + (int)(rnd() % (uint32_t)(n > 1 ? n - 1 : 1)); if (i < n) body[i] = (uint8_t)(rnd() % 7u); } break;
    case 4: for (int k = 0; k < 40; k++) { const int i = 1 + (int)(rnd() % (uint32_t)(n > 1 ? n - 1 : 1)); if (i < n) body[i] = (uint8_t)rnd(); } break;
    default: { const int i = n / 2; if (i < n && i > 0) body[i] = (uint8_t)(body[i] ? 0 : 3); } break;
    }
}
#define XF_MALFORM(p, len, variant, block) srt_malform(p, len, variant, block)
#define XF_MALFORM_VARIANTS 6
#include "xf_harness.hpp"
