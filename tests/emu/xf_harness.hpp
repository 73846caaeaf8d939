// Shared driver of the transform emulation tests (tests/emu/*_emu.cpp): reads a case file (u32 nBlocks, then per block u32 len +
// bytes), runs the oracle's forward transform, the kernels' forward (same capacity: same accept / refuse decision, same bytes) and
// the kernels' inverse of what the oracle produced. The including file defines, before including this header and after including
// its .hip file:  XF_TTYPE (kanzi transform id), XF_FWD(st), XF_INV(st)  and  XF_SCRATCH_U32(nBlocks, maxLen); optionally
// XF_MALFORM(bytes, len, variant, block) with XF_MALFORM_VARIANTS: damaged copies of the oracle's output go through both inverses,
// which have to agree on accept / refuse and on every byte of what they accept.
// A second argument below 6 is a bitstream version: the oracle then writes the transform's OLD block layout (the reference only reads
// those), the kernels' forward is not compared (it writes the current one), their inverse gets the version.
// Test infrastructure only.
#include <stdio.h>
#include <vector>
#include "emu_corrupt.hpp"

extern "C" int knzo_transform_forward(int ttype, const uint8_t* src, int n, uint8_t* dst, int dstCap, int etype, int* outLen);
extern "C" int knzo_transform_inverse(int ttype, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* outLen);
extern "C" void knzo_set_bs_version(int v);

namespace knz { thread_local ProfHook* g_prof = nullptr; }

int main(int argc, char** argv)
{
    using namespace knz;
    if (argc < 2) return 2;
    const int bsVersion = argc > 2 ? atoi(argv[2]) : 6;
    knzo_set_bs_version(bsVersion);
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    u32 nBlocks = 0;
    if (fread(&nBlocks, 4, 1, f) != 1) return 2;
    std::vector<std::vector<u8>> plain(nBlocks), want(nBlocks), fwd(nBlocks), back(nBlocks);
    std::vector<int> wantOk(nBlocks), wantLen(nBlocks);
    std::vector<u32> origN(nBlocks), fcap(nBlocks);
    u32 maxLen = 1;
    for (u32 b = 0; b < nBlocks; b++) {
        u32 n = 0;
        if (fread(&n, 4, 1, f) != 1) return 2;
        origN[b] = n;
        plain[b].assign((size_t)n + 64, 0);                      // (the LZ kernels read a few bytes past the end, like the reference)
        if (n && fread(plain[b].data(), 1, n, f) != n) return 2;
        fcap[b] = n + n / 8 + 2048;
        want[b].assign((size_t)fcap[b] + 64, 0);
        int el = 0;
        wantOk[b] = knzo_transform_forward(XF_TTYPE, plain[b].data(), (int)n, want[b].data(), (int)fcap[b], 5, &el);
        wantLen[b] = wantOk[b] ? el : 0;
        fwd[b].assign((size_t)fcap[b] + 64, 0xEE);
        back[b].assign((size_t)n + 64, 0xEE);
        maxLen = std::max(maxLen, fcap[b]);
    }
    fclose(f);
    std::vector<const u8*> src(nBlocks); std::vector<u8*> dst(nBlocks);
    std::vector<u32> len(nBlocks), cap(nBlocks), newLen(nBlocks, 0);
    std::vector<u8> ok(nBlocks, 0);
    std::vector<u32> scratch(XF_SCRATCH_U32((int)nBlocks, maxLen) + 64);
    XfStage st;
    st.src = src.data(); st.dst = dst.data(); st.len = len.data(); st.cap = cap.data(); st.ok = ok.data(); st.newLen = newLen.data();
    st.nBlocks = (int)nBlocks; st.maxLen = maxLen; st.scratchU32 = scratch.data(); st.entropyType = 5; st.bsVersion = bsVersion;
    int bad = 0;
    for (u32 b = 0; b < nBlocks; b++) { src[b] = plain[b].data(); dst[b] = fwd[b].data(); len[b] = origN[b]; cap[b] = fcap[b]; }
    if (bsVersion >= 6) XF_FWD(st);
    for (u32 b = 0; b < nBlocks && bsVersion >= 6; b++) {
        const bool same = (ok[b] != 0) == (wantOk[b] != 0) && (!wantOk[b] || ((int)newLen[b] == wantLen[b] && memcmp(fwd[b].data(), want[b].data(), (size_t)wantLen[b]) == 0));
        if (!same) {
            size_t at = 0;
            while (wantOk[b] && at < (size_t)wantLen[b] && fwd[b][at] == want[b][at]) at++;
            printf("FAIL forward block %u (n=%u): ok %d/%d len %u/%d first difference at %zu\n", b, origN[b], ok[b], wantOk[b], newLen[b], wantLen[b], at);
            bad++;
        }
    }
    for (u32 b = 0; b < nBlocks; b++) {
        if (wantOk[b]) emu_corrupt(want[b].data(), (size_t)wantLen[b], b);
        src[b] = want[b].data(); dst[b] = back[b].data(); ok[b] = 0; newLen[b] = 0;
        len[b] = wantOk[b] ? (u32)wantLen[b] : 0;                // 0 = the block takes no part
        cap[b] = origN[b];
    }
    XF_INV(st);
    if (emu_corrupt_on()) { int no = 0; for (u32 b = 0; b < nBlocks; b++) no += wantOk[b] && !ok[b]; printf("damaged input: %d of %u blocks refused\n", no, nBlocks); return 0; }
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
#ifdef XF_MALFORM
    for (int variant = 0; variant < XF_MALFORM_VARIANTS; variant++) {
        std::vector<std::vector<u8>> mal(nBlocks), wantBack(nBlocks);
        std::vector<int> wOk(nBlocks, 0), wLen(nBlocks, 0);
        for (u32 b = 0; b < nBlocks; b++) {
            mal[b] = want[b];
            if (wantOk[b]) XF_MALFORM(mal[b].data(), wantLen[b], variant, (int)b);
            wantBack[b].assign((size_t)origN[b] + 64, 0xEE);
            if (wantOk[b]) { int ol = 0; wOk[b] = knzo_transform_inverse(XF_TTYPE, mal[b].data(), wantLen[b], wantBack[b].data(), (int)origN[b], &ol); wLen[b] = wOk[b] ? ol : 0; }
            std::fill(back[b].begin(), back[b].end(), (u8)0xEE);
            src[b] = mal[b].data(); dst[b] = back[b].data(); ok[b] = 0; newLen[b] = 0;
            len[b] = wantOk[b] ? (u32)wantLen[b] : 0;
            cap[b] = origN[b];
        }
        XF_INV(st);
        for (u32 b = 0; b < nBlocks; b++) {
            if (!wantOk[b]) continue;
            const bool same = (ok[b] != 0) == (wOk[b] != 0) && (!wOk[b] || ((int)newLen[b] == wLen[b] && memcmp(back[b].data(), wantBack[b].data(), (size_t)wLen[b]) == 0));
            if (!same) {
                size_t at = 0;
                while (wOk[b] && at < (size_t)wLen[b] && back[b][at] == wantBack[b][at]) at++;
                printf("FAIL malformed inverse, variant %d block %u (n=%u): ok %d/%d len %u/%d first difference at %zu\n", variant, b, origN[b], ok[b], wOk[b], newLen[b], wLen[b], at);
                bad++;
            }
        }
    }
#endif
    printf(bad ? "FAILED %d blocks\n" : "OK %u blocks\n", bad ? bad : nBlocks, nBlocks);
    return bad ? 1 : 0;
}
