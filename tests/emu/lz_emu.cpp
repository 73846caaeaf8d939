// CPU-only check of the LZ kernels of kanzi-cpp_amd/csrc/lz.hip (transform id 3: candidate phase with the library sort, the
// wave-per-block walk, the decoder) against the oracle, forward and inverse; see xf_harness.hpp.
#define KNZ_EMU 1
#include "hip/hip_runtime.h"
#include "../../kanzi-cpp_amd/csrc/lz.hip"
namespace {
std::vector<unsigned char> g_lzScratch;
int lz_fwd(const knz::XfStage& st)
{
    const size_t bytes = knz::lz_forward_scratch_bytes(3, st.nBlocks, st.maxLen);
    g_lzScratch.assign(bytes + 512, 0);
    void* sc = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(g_lzScratch.data()) + 255) & ~(uintptr_t)255);
    return knz::launch_lz_forward(nullptr, st, 3, sc, bytes);
}
// both decoders: the data-parallel one (parse, expand, pointer jumping, emit) and, with KNZ_LZ_SERIAL_DECODE=1, the one-wave-per-block one
std::vector<unsigned char> g_lzInvScratch;
void lz_inv(const knz::XfStage& st)
{
    knz::u32 maxCap = 1;
    for (int b = 0; b < st.nBlocks; b++) maxCap = std::max(maxCap, st.cap[b]);
    if (knz::lz_serial_decode(-1)) { knz::launch_lz_inverse(nullptr, st, nullptr, 0, 0); return; }
    const size_t bytes = knz::lz_inverse_scratch_bytes(st.nBlocks, maxCap);
    g_lzInvScratch.assign(bytes + 512, 0x7F);            // (stale entries must not look like finished ones)
    void* sc = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(g_lzInvScratch.data()) + 255) & ~(uintptr_t)255);
    knz::launch_lz_inverse(nullptr, st, sc, bytes, maxCap);
}
}
#define XF_TTYPE 3
#define XF_FWD(st) do { if (lz_fwd(st) != 0) { printf("launch_lz_forward failed\n"); return 1; } } while (0)
#define XF_INV(st) lz_inv(st)
#define XF_SCRATCH_U32(nb, ml) ((size_t)0)
#include "xf_harness.hpp"
