// Internal definitions shared by the gfx950 kernels and the C-ABI layer.
// Wave size is 64 on CDNA4; all wave-level code below hard-codes it.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/knz_hip.h"

namespace knz {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

constexpr int WAVE = 64;
constexpr u32 ENT_CHUNK = 16384;          // ANS0 / Huffman chunk (ANSRangeEncoder.cpp:59, HuffmanCommon.cpp:20-21)
constexpr u32 HDR_BYTES = 576;            // per-chunk header bit buffer (max 3498 bits for ANS0)
constexpr u32 HDR_WORDS = HDR_BYTES / 4;
constexpr u32 PAY_BYTES = 33280;          // per-chunk payload staging (>= 2 bytes/symbol + tail), multiple of 256
constexpr u32 TMP_STRIDE = HDR_BYTES + PAY_BYTES;   // 33856, multiple of 64

// Blocks of one batch: block b occupies [ptr[b], +len[b]).
struct BlockView {
    const u8* const* ptr;   // device array of per-block pointers (16-byte aligned)
    const u32* len;         // device array
};

// What one entropy "chunk" contributes to the block's bit stream, in order:
// hdr bits (from the chunk's header buffer), mid bytes (inline), up to 4 pieces (device pointers,
// bit counts, MSB-first from the first byte), trailer bytes (inline).
struct ChunkDesc {
    u32 hdrBits;
    u32 midLen;          // bytes
    u32 mid[6];          // up to 24 inline bytes, memory order
    u32 nPieces;
    u32 pieceBits[4];
    u32 trailerLen;      // bytes
    u32 trailer[2];      // up to 8 inline bytes, memory order
    u32 aux;             // codec specific (alphabet size)
    const u8* piecePtr[4];
    u64 relBit;          // bit offset of this chunk inside the block's entropy payload (k_block_sum)
    u64 totalBits;
};

// Per-block framing results
struct BlockInfo {
    u64 payloadBits;     // entropy bits of the block
    u64 written;         // header + payload ("written" in EncodingTask::run)
    u64 bitOff;          // position of the 5-bit length prefix in the output stream
    u32 hdrBits;         // mode byte (+skip byte) + length field + checksum
    u32 lw;
};

__device__ __forceinline__ u32 bswap32(u32 v) { return __builtin_bswap32(v); }

__device__ __forceinline__ int ilog2_u32(u32 x) { return 31 - __clz((int)x); }

// number of bits needed to represent v (0 -> 0)
__device__ __forceinline__ u32 bitlen_u32(u32 v) { return v == 0 ? 0u : (u32)(32 - __clz((int)v)); }

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// A pointer read from an array of pointers (block sources, destinations) is a "flat" pointer to the compiler: it cannot know that the
// memory is global, every access becomes a flat_load / flat_store, and those count as LDS operations too -- a wait for an LDS read
// then also waits for the memory access. ldg / stg access through the global address space explicitly.
template <typename T>
__device__ __forceinline__ T ldg(const void* p)
{
#ifdef KNZ_EMU
    T v; __builtin_memcpy(&v, p, sizeof(T)); return v;
#else
    return *(const __attribute__((address_space(1))) T*)(uintptr_t)p;
#endif
}
template <typename T>
__device__ __forceinline__ void stg(void* p, T v)
{
#ifdef KNZ_EMU
    __builtin_memcpy(p, &v, sizeof(T));
#else
    *(__attribute__((address_space(1))) T*)(uintptr_t)p = v;
#endif
}
// eight bytes of a block's text from position q (little endian; bytes at or behind n read as zero), from three aligned dwords
__device__ __forceinline__ u64 text8(const u8* t, u32 q, u32 n)
{
    u64 x = 0;
    if (q + 12 <= n) {
        const uintptr_t a = reinterpret_cast<uintptr_t>(t + q);
        const u8* w = reinterpret_cast<const u8*>(a & ~(uintptr_t)3);
        const u32 sh = (u32)(a & 3) * 8;
        const u64 lo = (u64)ldg<u32>(w) | ((u64)ldg<u32>(w + 4) << 32);
        x = sh ? ((lo >> sh) | ((u64)ldg<u32>(w + 8) << (64 - sh))) : lo;
    } else {
        for (u32 j = 0; j < 8; j++) if (q + j < n) x |= (u64)ldg<u8>(t + q + j) << (8 * j);
    }
    return x;
}

// This library is written for ONE target: gfx950 (MI355X, 160 KiB of LDS per workgroup, wave64, gfx9-style s_waitcnt encoding). A device
// pass for anything else stops here instead of failing at launch or waiting on the wrong counters (KNZ_EMU: the CPU emulation of tests/emu).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(KNZ_EMU)
#error "kanzi-cpp_amd kernels are written for gfx950 only (LDS sizes, s_waitcnt encoding, wave64)"
#endif
constexpr unsigned KNZ_LDS_BYTES = 160u * 1024u;      // what one workgroup may declare on gfx950
// every load issued so far has arrived: vmcnt(0) in the gfx9 encoding, the other counters left alone. Used before a burst of stores --
// loads and stores share one counter that counts in issue order, so a wait for a load that follows a store is a wait for the store.
// Through the builtin (the compiler's own counter pass understands it; an asm wait leaves its conservative waits in place).
#ifdef KNZ_EMU
#define KNZ_LOADS_DONE() ((void)0)
#else
#define KNZ_LOADS_DONE() __builtin_amdgcn_s_waitcnt(0x0F70)
#endif

// Where a kernel relies on a wave executing its memory operations in program order across lanes (lane 0 stores, every lane loads
// right after), the CPU emulation of tests/emu -- whose lanes only meet at wave intrinsics -- needs a rendezvous; on the GPU it is nothing.
#ifdef KNZ_EMU
#define KNZ_WAVE_ORDER() ((void)__ballot(1))
#else
#define KNZ_WAVE_ORDER() ((void)0)
#endif
// A ballot inside divergent control flow: on the GPU the exec mask restricts it to the lanes that are there; the emulation has no
// exec mask, so the kernel says which lanes those are (all of them call it, nobody else does).
#ifdef KNZ_EMU
#define KNZ_BALLOT_OF(pred, lanes) hipemu_ballot_of((pred), (lanes))
#else
#define KNZ_BALLOT_OF(pred, lanes) __ballot(pred)
#endif

// Wave64 scans and reductions on the DPP data path (row shifts inside 16-lane rows, then row_bcast15 / row_bcast31
// to carry across rows): about a dozen VALU instructions, where the ds_bpermute-based __shfl versions pay an LDS
// round trip per step.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u32 dpp_or0(u32 v)
{
    // lanes without a source lane (or masked out) receive 0
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false);
}

__device__ __forceinline__ u32 wave_incl_scan(u32 v)
{
    v += dpp_or0<0x111, 0xF>(v);      // row_shr:1
    v += dpp_or0<0x112, 0xF>(v);      // row_shr:2
    v += dpp_or0<0x114, 0xF>(v);      // row_shr:4
    v += dpp_or0<0x118, 0xF>(v);      // row_shr:8
    v += dpp_or0<0x142, 0xA>(v);      // row_bcast:15 -> rows 1 and 3
    v += dpp_or0<0x143, 0xC>(v);      // row_bcast:31 -> rows 2 and 3
    return v;
}

__device__ __forceinline__ u32 wave_sum(u32 v)
{
    return (u32)__builtin_amdgcn_readlane((int)wave_incl_scan(v), 63);
}

__device__ __forceinline__ u32 wave_max(u32 v)
{
    u32 t;
    t = dpp_or0<0x111, 0xF>(v); v = t > v ? t : v;
    t = dpp_or0<0x112, 0xF>(v); v = t > v ? t : v;
    t = dpp_or0<0x114, 0xF>(v); v = t > v ? t : v;
    t = dpp_or0<0x118, 0xF>(v); v = t > v ? t : v;
    t = dpp_or0<0x142, 0xA>(v); v = t > v ? t : v;
    t = dpp_or0<0x143, 0xC>(v); v = t > v ? t : v;
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ u64 wave_incl_scan64(u64 v)
{
#define KNZ_SCAN64_STEP(CTRL, RM) { const u32 lo = dpp_or0<CTRL, RM>((u32)v); const u32 hi = dpp_or0<CTRL, RM>((u32)(v >> 32)); v += ((u64)hi << 32) | lo; }
    KNZ_SCAN64_STEP(0x111, 0xF)
    KNZ_SCAN64_STEP(0x112, 0xF)
    KNZ_SCAN64_STEP(0x114, 0xF)
    KNZ_SCAN64_STEP(0x118, 0xF)
    KNZ_SCAN64_STEP(0x142, 0xA)
    KNZ_SCAN64_STEP(0x143, 0xC)
#undef KNZ_SCAN64_STEP
    return v;
}

// OR `n` bits (value right-aligned, n <= 32) at bit position `pos` of an MSB-first bit buffer held
// as 32-bit words in "stream order" (word w holds stream bits [32w, 32w+32), MSB first). LDS or global.
__device__ __forceinline__ void or_bits_words(u32* words, u32 pos, u32 value, u32 n)
{
    if (n == 0) return;
    const u32 w = pos >> 5;
    const u32 o = pos & 31;
    const u64 v = ((u64)value << (64 - n)) >> o;     // value placed in a 64-bit window starting at word w
    const u32 hi = (u32)(v >> 32);
    const u32 lo = (u32)v;
    if (hi) atomicOr(&words[w], hi);
    if (lo) atomicOr(&words[w + 1], lo);
}

// Same, but the destination is a byte stream in memory (global), addressed as big-endian 32-bit words:
// memory word w must contain bswap32(stream word w).
__device__ __forceinline__ void or_bits_mem(u32* memWords, u64 pos, u64 value, u32 n)
{
    // n <= 64 - 31 is not required: handle up to 57 bits by splitting
    while (n > 0) {
        const u32 take = n > 32 ? n - 32 : n;        // first the high part
        const u32 part = (u32)((value >> (n - take)) & (take == 32 ? 0xFFFFFFFFull : ((1ull << take) - 1)));
        const u64 w = pos >> 5;
        const u32 o = (u32)(pos & 31);
        const u64 v = ((u64)part << (64 - take)) >> o;
        const u32 hi = (u32)(v >> 32);
        const u32 lo = (u32)v;
        if (hi) atomicOr(&memWords[w], bswap32(hi));
        if (lo) atomicOr(&memWords[w + 1], bswap32(lo));
        pos += take;
        n -= take;
    }
}

// --- launch helpers (host) ---
struct Ctx;
const char* hipErrStr(hipError_t e);

// Optional per-kernel timing hook (HIP events on the launch stream), installed by the API layer.
struct ProfHook {
    virtual void begin(const char* name) = 0;
    virtual void end() = 0;
    virtual ~ProfHook() {}
};
extern thread_local ProfHook* g_prof;
struct KScope {
    explicit KScope(const char* n) { if (g_prof) g_prof->begin(n); }
    ~KScope() { if (g_prof) g_prof->end(); }
};

}  // namespace knz

namespace knz {

// MSB-first bit reader over a byte stream held in global memory, read as aligned 32-bit words.
// `nWords` bounds the loads (words past the end read as zero); `error` latches reads past `limitBits`.
struct BitSrc {
    const u32* words;    // 4-byte aligned
    u64 nWords;          // complete words available
    u64 nBytes;          // bytes available (the last 1-3 bytes are read individually)
    u64 limitBits;
};

__device__ __forceinline__ u32 src_word(const BitSrc& s, u64 w)
{
    if (w < s.nWords) return bswap32(s.words[w]);
    u32 v = 0;
    const u8* p = reinterpret_cast<const u8*>(s.words);
    for (u64 i = w * 4, k = 0; k < 4; i++, k++) v = (v << 8) | (i < s.nBytes ? p[i] : 0u);
    return v;
}

// n in [0, 32]
__device__ __forceinline__ u32 peek_bits(const BitSrc& s, u64 pos, u32 n)
{
    if (n == 0) return 0;
    const u64 w = pos >> 5;
    const u64 win = ((u64)src_word(s, w) << 32) | (u64)src_word(s, w + 1);
    return (u32)((win << (pos & 31)) >> (64 - n));
}

__device__ __forceinline__ u32 take_bits(const BitSrc& s, u64& pos, u32 n, int& err)
{
    if (pos + n > s.limitBits) { err = 1; pos = s.limitBits; return 0; }
    const u32 v = peek_bits(s, pos, n);
    pos += n;
    return v;
}

// EntropyUtils::readVarInt (entropy/EntropyUtils.cpp:261-285)
__device__ __forceinline__ u32 take_varint(const BitSrc& s, u64& pos, int& err)
{
    u32 value = take_bits(s, pos, 8, err);
    u32 res = value & 0x7F;
    for (int shift = 7; value >= 128 && !err; shift += 7) {
        value = take_bits(s, pos, 8, err);
        if (shift == 28) {
            if (value >= 128 || (value & 0x70) != 0) { err = 1; return 0; }
            res |= (value & 0x0F) << shift;
            return res;
        }
        res |= (value & 0x7F) << shift;
    }
    return res;
}

// Per-block decode bookkeeping
struct DecBlock {
    u64 payloadBit;      // first bit of the block's private stream (mode byte)
    u64 bits;            // length of the private stream
    u64 entropyBit;      // first entropy bit (after mode/len/checksum)
    u64 checksum;
    u32 preLen;          // preTransformLength
    u32 skipFlags;
    u32 copyBlock;
    int32_t error;       // 0 or kanzi error code
    u64 usedBits;        // bits consumed by the entropy decoder (relative to entropyBit)
};

}  // namespace knz
