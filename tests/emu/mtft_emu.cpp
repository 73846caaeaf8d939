// CPU-only check of the MTFT kernels' logic (forward: tile tables, two-level prefix max, in-register list; inverse: symbolic
// tiles, two-level composition of permutations, resolve): kanzi-cpp_amd/csrc/mtft.hip compiled as plain C++ against tools/hipemu,
// compared with the oracle's MTFT in both directions. Test infrastructure only.
//   usage: mtft_emu <case file>    (binary: u32 nBlocks, then per block u32 len + bytes)
#define KNZ_EMU 1
#include "hip/hip_runtime.h"
#include "../../kanzi-cpp_amd/csrc/mtft.hip"

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
    std::vector<std::vector<u8>> plain(nBlocks), want(nBlocks), fwd(nBlocks), back(nBlocks);
    u32 maxLen = 1;
    for (u32 b = 0; b < nBlocks; b++) {
        u32 n = 0;
        if (fread(&n, 4, 1, f) != 1) return 2;
        plain[b].resize(n + 8);
        if (n && fread(plain[b].data(), 1, n, f) != n) return 2;
        want[b].resize(n + 64);
        int el = 0;
        if (!knzo_transform_forward(7, plain[b].data(), (int)n, want[b].data(), (int)n + 64, -1, &el) || (u32)el != n) { printf("oracle MTFT refused block %u\n", b); return 2; }
        fwd[b].assign(n + 64, 0xEE);
        back[b].assign(n + 64, 0xEE);
        maxLen = std::max(maxLen, n);
    }
    fclose(f);
    std::vector<const u8*> src(nBlocks); std::vector<u8*> dst(nBlocks);
    std::vector<u32> len(nBlocks), cap(nBlocks), newLen(nBlocks, 0);
    std::vector<u8> ok(nBlocks, 0);
    std::vector<u32> scratch(mtft_scratch_u32((int)nBlocks, maxLen) + 64);
    XfStage st;
    st.src = src.data(); st.dst = dst.data(); st.len = len.data(); st.cap = cap.data(); st.ok = ok.data(); st.newLen = newLen.data();
    st.nBlocks = (int)nBlocks; st.maxLen = maxLen; st.scratchU32 = scratch.data(); st.entropyType = -1;
    int bad = 0;
    for (u32 b = 0; b < nBlocks; b++) { src[b] = plain[b].data(); dst[b] = fwd[b].data(); len[b] = (u32)plain[b].size() - 8; cap[b] = len[b] + 64; }
    launch_mtft_forward(nullptr, st);
    for (u32 b = 0; b < nBlocks; b++) {
        const u32 n = len[b];
        if (!ok[b] || newLen[b] != n || memcmp(fwd[b].data(), want[b].data(), n) != 0) {
            u32 at = 0;
            while (at < n && fwd[b][at] == want[b][at]) at++;
            printf("FAIL forward block %u (n=%u): ok %d len %u, first difference at %u\n", b, n, ok[b], newLen[b], at);
            bad++;
        }
    }
    for (u32 b = 0; b < nBlocks; b++) { src[b] = want[b].data(); dst[b] = back[b].data(); ok[b] = 0; newLen[b] = 0; }
    launch_mtft_inverse(nullptr, st);
    for (u32 b = 0; b < nBlocks; b++) {
        const u32 n = len[b];
        if (!ok[b] || newLen[b] != n || memcmp(back[b].data(), plain[b].data(), n) != 0) {
            u32 at = 0;
            while (at < n && back[b][at] == plain[b][at]) at++;
            printf("FAIL inverse block %u (n=%u): ok %d len %u, first difference at %u\n", b, n, ok[b], newLen[b], at);
            bad++;
        }
    }
    printf(bad ? "FAILED %d blocks\n" : "OK %u blocks\n", bad ? bad : nBlocks, nBlocks);
    return bad ? 1 : 0;
}
