// CPU-only check of the inverse BWT kernels' logic: kanzi-cpp_amd/csrc/bwt.hip compiled as plain C++ against tools/hipemu,
// fed with the oracle's forward BWT block codec output, must reproduce the input. Test infrastructure only.
//   usage: bwt_inv_emu <case file> [bitstream version]   (binary: u32 nBlocks, then per block u32 len + bytes; a version below 6 selects
//   the block header of BWTBlockCodec.cpp:140-164 on both sides)
#include "hip/hip_runtime.h"
#include "../../kanzi-cpp_amd/csrc/bwt.hip"

#include <stdio.h>
#include <vector>
#include "emu_corrupt.hpp"

extern "C" int knzo_transform_forward(int ttype, const uint8_t* src, int n, uint8_t* dst, int dstCap, int etype, int* outLen);
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
    std::vector<std::vector<u8>> plain(nBlocks), enc(nBlocks), out(nBlocks);
    u32 maxLen = 1;
    for (u32 b = 0; b < nBlocks; b++) {
        u32 n = 0;
        if (fread(&n, 4, 1, f) != 1) return 2;
        plain[b].resize(n);
        if (n && fread(plain[b].data(), 1, n, f) != n) return 2;
        enc[b].resize(n + 64);
        int el = 0;
        if (!knzo_transform_forward(1, plain[b].data(), (int)n, enc[b].data(), (int)n + 33, -1, &el)) { printf("oracle forward refused block %u\n", b); return 2; }
        enc[b].resize(el);
        emu_corrupt(enc[b].data(), enc[b].size() < 64 ? enc[b].size() : 64, b);      // (the header is where damage changes the course; the body is any bytes)
        out[b].assign(n + 64, 0xEE);
        maxLen = std::max(maxLen, (u32)el);
    }
    fclose(f);
    std::vector<const u8*> src(nBlocks); std::vector<u8*> dst(nBlocks);
    std::vector<u32> len(nBlocks), cap(nBlocks), newLen(nBlocks, 0);
    std::vector<u8> ok(nBlocks, 0);
    for (u32 b = 0; b < nBlocks; b++) { enc[b].reserve(enc[b].size() + 64); src[b] = enc[b].data(); dst[b] = out[b].data(); len[b] = (u32)enc[b].size(); cap[b] = (u32)plain[b].size() + 16; }
    XfStage st;
    st.src = src.data(); st.dst = dst.data(); st.len = len.data(); st.cap = cap.data(); st.ok = ok.data(); st.newLen = newLen.data();
    st.nBlocks = (int)nBlocks; st.maxLen = maxLen; st.scratchU32 = nullptr; st.entropyType = -1; st.bsVersion = bsVersion;
    const size_t bytes = bwt_inverse_scratch_bytes((int)nBlocks, maxLen, (size_t)nBlocks * maxLen);
    std::vector<u8> scratch(bytes + 256);
    u8* sc = reinterpret_cast<u8*>((reinterpret_cast<uintptr_t>(scratch.data()) + 255) & ~(uintptr_t)255);
    u32 pinned[64];
    const int rc = launch_bwt_inverse(nullptr, st, sc, bytes, pinned);
    if (rc != 0) { printf("FAIL launch rc=%d\n", rc); return 1; }
    if (emu_corrupt_on()) { int no = 0; for (u32 b = 0; b < nBlocks; b++) no += !ok[b]; printf("damaged input: %d of %u blocks refused\n", no, nBlocks); return 0; }
    int bad = 0;
    for (u32 b = 0; b < nBlocks; b++) {
        const u32 n = (u32)plain[b].size();
        if (!ok[b] || newLen[b] != n || memcmp(out[b].data(), plain[b].data(), n) != 0) {
            u32 at = 0;
            while (at < n && out[b][at] == plain[b][at]) at++;
            printf("FAIL block %u (n=%u): ok %d len %u, first difference at %u\n", b, n, ok[b], newLen[b], at);
            bad++;
        }
    }
    printf(bad ? "FAILED %d blocks\n" : "OK %u blocks\n", bad ? bad : nBlocks, nBlocks);
    return bad ? 1 : 0;
}
