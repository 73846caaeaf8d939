// CPU-only check of the rANS order-0 decoder kernels (k_ans_scan<0>: per-block walk over the chunk headers; k_ans0_decode: slot tables,
// four interleaved states per chunk sharing one byte pointer, payload ring in LDS):
// kanzi-cpp_amd/csrc/ans_dec.hip compiled as plain C++ against tools/hipemu, fed with the oracle's ANS0 streams, must
// reproduce the input and consume exactly the stream's bits. Test infrastructure only.
//   usage: ans0_emu <case file>    (binary: u32 nBlocks, then per block u32 len + bytes)
#define KNZ_EMU 1
#include "hip/hip_runtime.h"
#include "../../kanzi-cpp_amd/csrc/ans_dec.hip"

#include <stdio.h>
#include <vector>
#include "emu_corrupt.hpp"

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
    std::vector<std::vector<u8>> plain(nBlocks), out(nBlocks);
    std::vector<u8> stream;
    std::vector<DecBlock> blocks(nBlocks);
    u32 maxLen = 1;
    for (u32 b = 0; b < nBlocks; b++) {
        u32 n = 0;
        if (fread(&n, 4, 1, f) != 1) return 2;
        plain[b].resize(n);
        if (n && fread(plain[b].data(), 1, n, f) != n) return 2;
        std::vector<u8> enc((size_t)n + n / 2 + 4096);
        const int64_t bits = knzo_entropy_encode(5, plain[b].data(), n, enc.data(), enc.size());
        if (bits < 0) { printf("oracle refused block %u\n", b); return 2; }
        // blocks back to back at bit granularity, like in a stream: an odd bit offset for every second block
        const u64 at = (u64)stream.size() * 8 + ((b & 1) ? 3 : 0);
        stream.resize((size_t)((at + (u64)bits + 7) / 8) + 1, 0);
        for (int64_t i = 0; i < bits; i++) if ((enc[(size_t)(i >> 3)] >> (7 - (i & 7))) & 1) stream[(size_t)((at + (u64)i) >> 3)] |= (u8)(0x80 >> ((at + (u64)i) & 7));
        DecBlock& d = blocks[b];
        memset(&d, 0, sizeof(d));
        d.payloadBit = at; d.bits = (u64)bits; d.entropyBit = at; d.preLen = n;
        out[b].assign((size_t)n + 64, 0xEE);
        maxLen = std::max(maxLen, n);
    }
    fclose(f);
    emu_corrupt(stream.data(), stream.size(), 1);
    stream.resize((stream.size() + 64 + 3) & ~(size_t)3, 0);
    std::vector<u32> words(stream.size() / 4);
    memcpy(words.data(), stream.data(), stream.size());
    BitSrc src;
    src.words = words.data(); src.nWords = words.size(); src.nBytes = stream.size(); src.limitBits = (u64)stream.size() * 8;
    const int maxChunks = (int)((maxLen + ENT_CHUNK - 1) / ENT_CHUNK);
    std::vector<u8> meta(ans0_dec_chunk_bytes() * (size_t)nBlocks * maxChunks + 64);
    std::vector<u8*> outPtr(nBlocks);
    for (u32 b = 0; b < nBlocks; b++) outPtr[b] = out[b].data();
    launch_ans0_decode(nullptr, src, blocks.data(), (int)nBlocks, maxChunks, meta.data(), outPtr.data());
    int bad = 0;
    if (emu_corrupt_on()) { int errs = 0; for (u32 b = 0; b < nBlocks; b++) errs += blocks[b].error != 0; printf("damaged input: %d of %u blocks refused\n", errs, nBlocks); return 0; }
    for (u32 b = 0; b < nBlocks; b++) {
        const u32 n = (u32)plain[b].size();
        if (blocks[b].error || blocks[b].usedBits != blocks[b].bits || memcmp(out[b].data(), plain[b].data(), n) != 0) {
            u32 at = 0;
            while (at < n && out[b][at] == plain[b][at]) at++;
            printf("FAIL block %u (n=%u): error %d used %llu of %llu bits, first difference at %u\n", b, n, blocks[b].error,
                   (unsigned long long)blocks[b].usedBits, (unsigned long long)blocks[b].bits, at);
            bad++;
        }
    }
    printf(bad ? "FAILED %d blocks\n" : "OK %u blocks\n", bad ? bad : nBlocks, nBlocks);
    return bad ? 1 : 0;
}
