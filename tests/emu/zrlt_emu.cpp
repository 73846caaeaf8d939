// CPU-only check of the ZRLT kernels' logic (forward: zero-run tokens with tile prefix sums; inverse: token classes, counts, emit):
// kanzi-cpp_amd/csrc/zrlt_mtft.hip compiled as plain C++ against tools/hipemu, compared with the oracle's ZRLT in both directions
// (the forward result's length differs from the input's; blocks the oracle's forward refuses are checked for refusal). Test
// infrastructure only.     usage: zrlt_emu <case file>    (binary: u32 nBlocks, then per block u32 len + bytes)
#define KNZ_EMU 1
#include "hip/hip_runtime.h"
#include "../../kanzi-cpp_amd/csrc/zrlt_mtft.hip"

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
    std::vector<int> wantOk(nBlocks), wantLen(nBlocks);
    u32 maxLen = 1;
    for (u32 b = 0; b < nBlocks; b++) {
        u32 n = 0;
        if (fread(&n, 4, 1, f) != 1) return 2;
        plain[b].resize(n + 8);
        if (n && fread(plain[b].data(), 1, n, f) != n) return 2;
        want[b].assign(n + 64, 0);
        int el = 0;
        wantOk[b] = knzo_transform_forward(6, plain[b].data(), (int)n, want[b].data(), (int)n, -1, &el);    // capacity n: an expanding block is refused
        wantLen[b] = wantOk[b] ? el : 0;
        fwd[b].assign(n + 64, 0xEE);
        back[b].assign(n + 64, 0xEE);
        maxLen = std::max(maxLen, n);
    }
    fclose(f);
    std::vector<const u8*> src(nBlocks); std::vector<u8*> dst(nBlocks);
    std::vector<u32> len(nBlocks), cap(nBlocks), newLen(nBlocks, 0);
    std::vector<u8> ok(nBlocks, 0);
    std::vector<u32> scratch(zrlt_scratch_u32((int)nBlocks, maxLen) + 64);
    XfStage st;
    st.src = src.data(); st.dst = dst.data(); st.len = len.data(); st.cap = cap.data(); st.ok = ok.data(); st.newLen = newLen.data();
    st.nBlocks = (int)nBlocks; st.maxLen = maxLen; st.scratchU32 = scratch.data(); st.entropyType = -1;
    int bad = 0;
    for (u32 b = 0; b < nBlocks; b++) { src[b] = plain[b].data(); dst[b] = fwd[b].data(); len[b] = (u32)plain[b].size() - 8; cap[b] = len[b]; }
    launch_zrlt_forward(nullptr, st);
    for (u32 b = 0; b < nBlocks; b++) {
        const u32 n = len[b];
        const bool same = (ok[b] != 0) == (wantOk[b] != 0) && (!wantOk[b] || ((int)newLen[b] == wantLen[b] && memcmp(fwd[b].data(), want[b].data(), (size_t)wantLen[b]) == 0));
        if (!same) { printf("FAIL forward block %u (n=%u): ok %d/%d len %u/%d\n", b, n, ok[b], wantOk[b], newLen[b], wantLen[b]); bad++; }
    }
    // inverse of the blocks the forward accepted
    std::vector<u32> origN(nBlocks);
    for (u32 b = 0; b < nBlocks; b++) {
        origN[b] = (u32)plain[b].size() - 8;
        src[b] = want[b].data(); dst[b] = back[b].data(); ok[b] = 0; newLen[b] = 0;
        len[b] = wantOk[b] ? (u32)wantLen[b] : 0;                    // 0 = the block takes no part
        cap[b] = origN[b];
    }
    launch_zrlt_inverse(nullptr, st);
    for (u32 b = 0; b < nBlocks; b++) {
        if (!wantOk[b]) continue;
        const u32 n = origN[b];
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
