// CPU-only check of the forward BWT kernels' logic: kanzi-cpp_amd/csrc/bwt_fwd.hip compiled as plain C++ against the
// fiber emulation in tools/hipemu (no GPU involved; the product never runs this way), compared with the oracle's
// BWT block codec (oracle/transforms.c -> knzo_transform_forward). Built and run by tests/test_emu_kernels.py.
//   usage: bwt_fwd_emu <case file>    (binary: u32 nBlocks, then per block u32 len + bytes)
#include "hip/hip_runtime.h"
#include "../../kanzi-cpp_amd/csrc/bwt_fwd.hip"

#include <stdio.h>
#include <vector>

extern "C" int knzo_transform_forward(int ttype, const uint8_t* src, int n, uint8_t* dst, int dstCap, int etype, int* outLen);

namespace knz { thread_local ProfHook* g_prof = nullptr; }

int main(int argc, char** argv)
{
    using namespace knz;
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    u32 nBlocks = 0;
    if (fread(&nBlocks, 4, 1, f) != 1) return 2;
    std::vector<std::vector<u8>> in(nBlocks), out(nBlocks);
    u32 maxLen = 1;
    for (u32 b = 0; b < nBlocks; b++) {
        u32 n = 0;
        if (fread(&n, 4, 1, f) != 1) return 2;
        in[b].resize(n + 16);
        if (n && fread(in[b].data(), 1, n, f) != n) return 2;
        in[b].resize(n);
        out[b].assign(n + 64, 0xEE);
        maxLen = std::max(maxLen, n);
    }
    fclose(f);
    std::vector<const u8*> src(nBlocks); std::vector<u8*> dst(nBlocks);
    std::vector<u32> len(nBlocks), cap(nBlocks), newLen(nBlocks, 0);
    std::vector<u8> ok(nBlocks, 0);
    for (u32 b = 0; b < nBlocks; b++) { src[b] = in[b].data(); dst[b] = out[b].data(); len[b] = (u32)in[b].size(); cap[b] = len[b] + 33; }
    XfStage st;
    st.src = src.data(); st.dst = dst.data(); st.len = len.data(); st.cap = cap.data(); st.ok = ok.data(); st.newLen = newLen.data();
    st.nBlocks = (int)nBlocks; st.maxLen = maxLen; st.scratchU32 = nullptr; st.entropyType = -1;
    const size_t bytes = bwt_forward_scratch_bytes((int)nBlocks, maxLen, (size_t)nBlocks * maxLen);
    std::vector<u8> scratch(bytes + 256);
    u8* sc = reinterpret_cast<u8*>((reinterpret_cast<uintptr_t>(scratch.data()) + 255) & ~(uintptr_t)255);
    u32 pinned[512];                 // the host read-back area (counters, the 256-entry sample histogram)
    const int rc = launch_bwt_forward(nullptr, st, sc, bytes, pinned);
    if (rc != 0) { printf("FAIL launch rc=%d\n", rc); return 1; }
    int bad = 0;
    for (u32 b = 0; b < nBlocks; b++) {
        std::vector<u8> ref(len[b] + 64);
        int refLen = 0;
        const int rok = knzo_transform_forward(1, in[b].data(), (int)len[b], ref.data(), (int)len[b] + 33, -1, &refLen);
        if (rok != (int)ok[b]) { printf("FAIL block %u: ok %d vs oracle %d\n", b, ok[b], rok); bad++; continue; }
        if (!rok) continue;
        if ((u32)refLen != newLen[b] || memcmp(ref.data(), out[b].data(), (size_t)refLen) != 0) {
            u32 at = 0;
            while (at < (u32)refLen && ref[at] == out[b][at]) at++;
            printf("FAIL block %u (n=%u): len %u vs %d, first difference at %u\n", b, len[b], newLen[b], refLen, at);
            bad++;
        }
    }
    printf(bad ? "FAILED %d blocks\n" : "OK %u blocks\n", bad ? bad : nBlocks, nBlocks);
    return bad ? 1 : 0;
}
