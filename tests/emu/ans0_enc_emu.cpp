// CPU-only check of the rANS order-0 encoder kernels and of the bit assembly (k_ans0_stats: alphabet, normalised frequencies and
// chunk header; k_ans0_encode: four interleaved states per chunk; k_block_sum / k_block_scan / k_assemble: the chunk pieces OR-ed
// into the block's bit stream): kanzi-cpp_amd/csrc/ans.hip + bitasm.hip compiled as plain C++ against tools/hipemu, one block per
// call in the per-stage form (no framing), compared bit for bit with the oracle's ANS0 stream. Test infrastructure only.
//   usage: ans0_enc_emu <case file>    (binary: u32 nBlocks, then per block u32 len + bytes)
#define KNZ_EMU 1
#include "hip/hip_runtime.h"
#include "../../kanzi-cpp_amd/csrc/ans.hip"
#include "../../kanzi-cpp_amd/csrc/bitasm.hip"

#include <stdio.h>
#include <vector>

extern "C" int64_t knzo_entropy_encode(int etype, const uint8_t* in, uint32_t n, uint8_t* out, size_t cap);

namespace knz { thread_local ProfHook* g_prof = nullptr; }

int main(int argc, char** argv)
{
    using namespace knz;
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    u32 nBlocks = 0;
    if (fread(&nBlocks, 4, 1, f) != 1) return 2;
    int bad = 0;
    for (u32 b = 0; b < nBlocks; b++) {
        u32 n = 0;
        if (fread(&n, 4, 1, f) != 1) return 2;
        std::vector<u8> plain((size_t)n + 64, 0);
        if (n && fread(plain.data(), 1, n, f) != n) return 2;
        std::vector<u8> want((size_t)n + n / 2 + 4096, 0);
        const int64_t wantBits = knzo_entropy_encode(5, plain.data(), n, want.data(), want.size());
        if (wantBits < 0) { printf("oracle refused block %u\n", b); return 2; }
        const int maxChunks = (int)((n + ENT_CHUNK - 1) / ENT_CHUNK);
        const size_t nSlots = (size_t)maxChunks;
        std::vector<ChunkDesc> desc(nSlots);
        std::vector<uint2> encTab(256 * nSlots);
        std::vector<u8> tmp((size_t)TMP_STRIDE * nSlots + 256);
        std::vector<BlockInfo> info(1);
        const u8* ptr = plain.data();
        u32 len = n, origLen = n;
        u8 skip = 0;
        u64 total = 0;
        BlockView view; view.ptr = &ptr; view.len = &len;
        launch_ans0_encode(nullptr, view, 1, maxChunks, desc.data(), encTab.data(), tmp.data());
        FrameParams fp; fp.framing = 0; fp.nTransforms = 1; fp.checksumBits = 0; fp.finish = 0; fp.prologueBits = 0;
        launch_block_sum(nullptr, desc.data(), info.data(), &len, 1, maxChunks, ENT_CHUNK, 1u);
        launch_block_scan(nullptr, info.data(), &len, &origLen, 1, fp, &total);
        std::vector<u32> out(((size_t)((total + 7) >> 3) + 8 + 3) / 4 + 16, 0);
        launch_assemble(nullptr, desc.data(), info.data(), &len, &origLen, &skip, nullptr, tmp.data(), 1, maxChunks, ENT_CHUNK, 1u, TMP_STRIDE, fp, out.data());
        const u8* got = reinterpret_cast<const u8*>(out.data());
        const size_t bytes = (size_t)((wantBits + 7) >> 3);
        if ((int64_t)total != wantBits || memcmp(got, want.data(), bytes) != 0) {
            size_t at = 0;
            while (at < bytes && got[at] == want[at]) at++;
            printf("FAIL block %u (n=%u): %llu bits, oracle %lld, first different byte %zu\n", b, n, (unsigned long long)total, (long long)wantBits, at);
            bad++;
        }
    }
    fclose(f);
    printf(bad ? "FAILED %d blocks\n" : "OK %u blocks\n", bad ? bad : nBlocks, nBlocks);
    return bad ? 1 : 0;
}
