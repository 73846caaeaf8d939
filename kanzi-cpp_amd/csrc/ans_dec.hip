// rANS order-0 decoder (kanzi "ANS0") on gfx950.
//
// Reference being replaced: entropy/ANSRangeDecoder.cpp:80-175 (decodeHeader), :177-216 (decode),
// :218-292 (decodeChunk), entropy/ANSRangeDecoder.hpp:85-103 (decodeSymbol),
// entropy/EntropyUtils.cpp:91-123 (decodeAlphabet), :261-285 (readVarInt).
//
//   k_ans_scan<0>  The stream has no chunk directory and chunk headers are bit-granular, so finding
//                  chunk c+1 needs the header of chunk c: a serial chain of (#chunks) steps per block.
//                  One WAVE per block: a 1 KiB window of the stream is staged in LDS (bit order), the window
//                  of the next chunk is prefetched at a guessed position while this one is parsed, the
//                  presence masks are counted in parallel (ballot/popcount), and the group walk
//                  (<= 32 dependent 4-bit reads) is a short uniform chain from LDS.  <1> walks the 256 context
//                  tables of an order-1 chunk with the same code.
//   k_ans0_decode  16 chunks per wave. Phase 1 rebuilds each chunk's tables with the whole wave (parallel
//                  frequency parse from the group offsets the scan recorded; two-level slot lookup:
//                  4-slot buckets -> rank, compact entries by rank). Phase 2: 4 lanes per chunk = the 4
//                  interleaved states; the shared forward byte pointer of the reference is a ballot +
//                  popcount per step; the payload is streamed through a 256-byte LDS ring per chunk kept
//                  filled by the chunk's own lanes (loads issued 8 steps ahead), so that no global load
//                  sits on the dependent chain; 4x4 symbols are transposed with DPP so that every lane
//                  stores one aligned dword.
//   k_ans1_*       order 1: one wave per 4 MiB chunk, the chunk's cumulative frequencies in LDS, a wave-wide search per step (see below).
#include "common.hpp"
#include "stages.hpp"

namespace knz {

constexpr u32 ANS_TOP = 1u << 15;
constexpr u32 ANS_LR = 12;
constexpr u32 ANS_MAX_CHUNK = 1u << 27;

struct AnsDecChunk {
    u64 maskBit;         // first presence-mask byte (partial alphabet), unused when asz == 256
    u64 freqBit;         // first frequency group
    u64 payloadBit;      // first payload byte
    u32 sz;              // payload bytes
    u32 st[4];
    u16 asz;
    u8 lr;
    u8 kind;             // 0 = coded, 1 = single symbol fill, 2 = raw bytes, 3 = unused slot
    u8 sym;              // fill symbol for kind 1
    u8 pad[3];
    u16 grp[44];         // per frequency group: (bit offset relative to freqBit of the logMax field) | logMax << 12
};

// The scan stages a 1 KiB window of the stream (4 words per lane, byte-swapped to bit order) in LDS.  Everything
// the header walk reads is wave-uniform: all lanes run the same short dependent chain (one LDS read + a
// handful of VALU operations per field), with no cross-lane traffic.
constexpr u32 SCAN_WIN_BITS = 8192;
constexpr u32 SCAN_NEED_BITS = 3498 + 40 + 128 + 64;   // longest header + var-int + 4 states + one spare dword pair

#ifdef KNZ_EMU
struct u32x4 { u32 x, y, z, w; };                     // (the CPU emulation of tests/emu is built with g++)
#else
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#endif

__device__ __forceinline__ u32 rl(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)l); }
// values the compiler must have in registers at this point (it would otherwise move an LDS read it thinks is optional behind the
// condition that picks it, i.e. behind the memory load the read is meant to overlap)
#ifdef KNZ_EMU
#define KNZ_KEEP3(a, b, c) ((void)0)
#else
#define KNZ_KEEP3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))
#endif
__device__ __forceinline__ u32 uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 uni64(u64 v) { return ((u64)uni((u32)(v >> 32)) << 32) | uni((u32)v); }

// n in [1, 32]; rel = bit offset from the window start
__device__ __forceinline__ u32 lds_bits(const u32* win, u32 rel, u32 n)
{
    const u32 i = rel >> 5;
    const u64 v = ((u64)win[i] << 32) | (u64)win[i + 1];
    return (u32)((v << (rel & 31)) >> (64 - n));
}

// ORDER 0: one table per 16 KiB chunk, meta slot ci.  ORDER 1: 256 tables per 4 MiB chunk, meta slots
// ci*257 + ctx, the chunk record (payload, states) in slot ci*257 + 256.
template <int ORDER>
__global__ __launch_bounds__(64) void k_ans_scan(BitSrc src, DecBlock* __restrict__ blocks, int nBlocks, int maxChunks,
                                                 AnsDecChunk* __restrict__ chunks)
{
    constexpr u32 CHUNK = ORDER ? ANS1_CHUNK : ENT_CHUNK;
    constexpr u32 DIM = ORDER ? 256u : 1u;
    constexpr u32 MAX_LR = ORDER ? 11u : ANS_LR;       // kanzi encoders emit 12 (order 0) / 11 (order 1); larger tables are not provisioned
    const int b = blockIdx.x;
    const int lane = lane_id();
    DecBlock& db = blocks[b];
    AnsDecChunk* cs = chunks + (size_t)b * maxChunks;
    for (int i = lane; i < maxChunks; i += 64) cs[i].kind = 3;
    if (db.error) return;
    __shared__ u32 win[256 + 8];
    const u64 limit = uni64(db.payloadBit + ((db.bits + 7) & ~7ull));
    u64 pos = uni64(db.entropyBit);
    const u64 entropyBit = pos;
    const u32 preLen = uni(db.preLen);
    if (db.copyBlock || preLen <= 32) {
        if (lane == 0) {
            cs[0].kind = 2;
            cs[0].payloadBit = pos;
            cs[0].sz = preLen;
            pos += 8ull * preLen;
            if (pos > limit) db.error = KNZ_ERR_PROCESS_BLOCK;
            db.usedBits = pos - db.entropyBit;
        }
        return;
    }
    const u32 nChunks = (preLen + CHUNK - 1) / CHUNK;
    const u64 lastWord = ((src.nBytes + 3) >> 2) - 1;
    // words past the end of the buffer are clamped to its last word: positions past `limit` are rejected
    // before anything read there is trusted
    auto issue = [&](u64 bit0, u32x4& dstv) {
        const u64 w = (bit0 >> 5) + 4ull * (u32)lane;
        const u64 w1 = w + 1, w2 = w + 2, w3 = w + 3;
        const u32* p = src.words;
        dstv.x = p[w < lastWord ? w : lastWord];
        dstv.y = p[w1 < lastWord ? w1 : lastWord];
        dstv.z = p[w2 < lastWord ? w2 : lastWord];
        dstv.w = p[w3 < lastWord ? w3 : lastWord];
    };
    u32x4 wr = { 0, 0, 0, 0 };
    u64 winBit0 = 0;
    bool winValid = false;
    u32x4 guess = { 0, 0, 0, 0 };
    u64 guessBit0 = 0;
    bool guessValid = false;
    // make stream bits [pos, pos + need) readable through `win`: keep the current window if it still covers
    // them, else take the prefetched guess if it does, else load exactly (blocking)
    auto ensure = [&](u32 need) {
        if (winValid && winBit0 <= pos && pos + need <= winBit0 + SCAN_WIN_BITS) return;
        if (guessValid && guessBit0 <= pos && pos + need <= guessBit0 + SCAN_WIN_BITS) {
            wr = guess; winBit0 = guessBit0;
        } else {
            winBit0 = pos & ~31ull;
            issue(winBit0, wr);
        }
        guessValid = false;
        __syncthreads();
        {
            u32x4 sw = { bswap32(wr.x), bswap32(wr.y), bswap32(wr.z), bswap32(wr.w) };
            *reinterpret_cast<u32x4*>(&win[4 * lane]) = sw;
        }
        __syncthreads();
        winValid = true;
    };
    u64 prevPos = pos;
    u32 myGrp = 0;
    int err = 0;
    for (u32 ci = 0; ci < nChunks && !err; ci++) {
        ensure(ORDER ? 3498u + 64u : SCAN_NEED_BITS);
        if (ORDER == 0) {
            // prefetch for the next chunk: same compressed size as this one, window centred on the estimate
            guessValid = false;
            if (ci >= 1 && ci + 1 < nChunks) {
                const u64 est = pos + (pos - prevPos);
                guessBit0 = (est > 2304 ? est - 2304 : 0) & ~31ull;
                issue(guessBit0, guess);
                guessValid = true;
            }
            prevPos = pos;
        }
        const u32 lr = 8 + lds_bits(win, (u32)(pos - winBit0), 3);
        pos += 3;
        if (lr > MAX_LR) { err = 1; break; }
        u32 aszSum = 0;
        u32 firstSym = 0;
        for (u32 k = 0; k < DIM; k++) {
            if (ORDER) ensure(3498u + 64u);
            AnsDecChunk& c = cs[ORDER ? ci * 257 + k : ci];
            u32 p = (u32)(pos - winBit0);                  // header positions are window-relative from here
            u32 asz = 0;
            u64 maskBit = 0;
            const u32 hb = lds_bits(win, p, 7);            // flag bits and (partial alphabets) the mask count
            if ((hb >> 6) == 0) { asz = ((hb >> 5) & 1) ? 0u : 256u; p += 2; }
            else {
                const u32 lastMask = (hb >> 1) & 31;
                p += 6;
                maskBit = winBit0 + p;
                // one mask byte per lane
                const u32 byte = ((u32)lane <= lastMask) ? lds_bits(win, p + 8u * (u32)lane, 8) : 0u;
                const u32 pc = __popc(byte);
                asz = wave_sum(pc);
                const u64 nz = __ballot(byte != 0);
                if (nz) {
                    const int fl = __ffsll((long long)nz) - 1;
                    const u32 fb = rl(byte, (u32)fl);
                    firstSym = 8u * (u32)fl + (u32)(__ffs((int)fb) - 1);
                }
                p += 8u * (lastMask + 1);
            }
            asz = uni(asz);
            if (asz == 0) {
                if (ORDER == 0) { err = 2; break; }        // decode() returns a short count -> failure
                pos = winBit0 + p;                         // order 1: unused context, no table
                continue;
            }
            aszSum += asz;
            // ---- group walk (uniform): remember (offset | logMax << 12) of group g in lane g
            const u32 chk = (asz >= 64) ? 8u : 6u;
            const u32 llr = (u32)ilog2_u32(lr) + 1u;       // 4 for every legal lr (8..12)
            u32 q = p;
            const u32 nGroups = (asz - 1 + chk - 1) / chk;
            const u32 lastCnt = (asz - 1) - (nGroups - 1) * chk;
            u32 tooBig = 0;
            if (chk == 8) {
                // Groups of 8: every width field sits at a multiple of 4 bits from p.  Re-align the header so that p
                // is bit 0 of a 128-word register window (word k in lane k & 63 of r0 / r1): a field never straddles
                // a word, and the walk becomes readlane + scalar arithmetic with no LDS round trip on the chain.
                const u32 wi = uni(p) >> 5, sh = uni(p) & 31;
                const u32 a0 = win[wi + (u32)lane], b0 = win[wi + (u32)lane + 1];
                const u32 a1 = win[wi + (u32)lane + 64], b1 = win[wi + (u32)lane + 65];
                const u32 r0 = sh ? ((a0 << sh) | (b0 >> (32 - sh))) : a0;
                const u32 r1 = sh ? ((a1 << sh) | (b1 >> (32 - sh))) : a1;
                // (per group: two readlanes, a shift, a scalar maximum for the width check and the lane write -- the walk is the scan's
                // chain, 512 chunks x up to 32 groups per block, on a wave that issues one instruction per four cycles; the last group,
                // whose count differs, is peeled so that the loop body has no select)
                u32 rel = 0;
                u32 worst = 0;
                auto width_at = [&](u32 r) -> u32 {
                    const u32 i = r >> 5;
                    const u32 wlo = rl(r0, i & 63), whi = rl(r1, i & 63);
                    const u32 wsel = (i < 64) ? wlo : whi;
                    return (i < 128) ? ((wsel >> (28 - (r & 31))) & 15u) : 15u;                  // past the window: invalid
                };
                u32 g = 0;
                for (; g + 1 < nGroups; g++) {
                    const u32 logMax = width_at(rel);
                    worst = (logMax > worst) ? logMax : worst;
                    if ((u32)lane == g) myGrp = rel | (logMax << 12);
                    rel += llr + chk * logMax;
                }
                {
                    const u32 logMax = width_at(rel);
                    worst = (logMax > worst) ? logMax : worst;
                    if ((u32)lane == g) myGrp = rel | (logMax << 12);
                    rel += llr + lastCnt * logMax;
                }
                tooBig |= (worst > lr) ? 1u : 0u;
                q = p + rel;
            } else {
                for (u32 g = 0; g < nGroups; g++) {
                    const u32 logMax = lds_bits(win, q, llr);
                    tooBig |= (logMax > lr) ? 1u : 0u;          // checked after the walk; a bad value cannot fault:
                    if ((u32)lane == g) myGrp = (q - p) | (logMax << 12);   // LDS reads past the window just return
                    q += llr + ((g + 1 == nGroups) ? lastCnt : chk) * logMax;   // unrelated words
                }
            }
            if (tooBig) { err = 1; break; }
            if ((u32)lane < nGroups) c.grp[lane] = (u16)myGrp;
            if (lane == 0) {
                c.lr = (u8)lr;
                c.maskBit = maskBit;
                c.asz = (u16)asz;
                c.freqBit = winBit0 + p;
                c.sym = (u8)firstSym;
                if (ORDER) c.kind = 0;
            }
            pos = winBit0 + q;
        }
        if (err) break;
        AnsDecChunk& cr = cs[ORDER ? ci * 257 + 256 : ci];
        u32 kind, sz = 0, st0 = 0, st1 = 0, st2 = 0, st3 = 0;
        u64 payloadBit = 0;
        u64 endPos = pos;
        if (ORDER == 0 && aszSum == 1) {
            kind = 1;
        } else {
            if (ORDER && aszSum == 0) { err = 2; break; }   // ANSRangeDecoder.cpp:196-199: short count
            // var-int and the 4 states
            if (ORDER) ensure(40u + 128u + 64u);
            u32 q = (u32)(pos - winBit0);
            u32 value = lds_bits(win, q, 8); q += 8;
            u32 res = value & 0x7F;
            for (int shift = 7; value >= 128; shift += 7) {
                value = lds_bits(win, q, 8); q += 8;
                if (shift == 28) { if (value >= 128 || (value & 0x70) != 0) err = 1; res |= (value & 0x0F) << shift; break; }
                res |= (value & 0x7F) << shift;
            }
            sz = res;
            if (sz >= ANS_MAX_CHUNK || sz > 2 * CHUNK - 2) err = 1;
            st0 = lds_bits(win, q, 32); st1 = lds_bits(win, q + 32, 32); st2 = lds_bits(win, q + 64, 32); st3 = lds_bits(win, q + 96, 32);
            q += 128;
            payloadBit = winBit0 + q;
            endPos = payloadBit + 8ull * sz;
            kind = 0;
        }
        if (endPos > limit) err = 1;
        if (err) break;
        if (lane == 0) {
            cr.sz = sz;
            cr.st[0] = st0; cr.st[1] = st1; cr.st[2] = st2; cr.st[3] = st3;
            cr.payloadBit = payloadBit;
            cr.kind = (u8)kind;
            if (ORDER) cr.lr = (u8)lr;
        }
        pos = endPos;
    }
    if (lane == 0) {
        if (err) db.error = KNZ_ERR_PROCESS_BLOCK;
        db.usedBits = pos - entropyBit;
    }
}

// ------------------------------------------------------------------------------------------------
// k_ans0_decode.  The kernel is bound by the dependent chain of one decode step times the 4096 steps of a
// chunk, so the layout is chosen to (1) keep every chunk of a 200 MB job resident at once (2.3 KiB of LDS
// per chunk, 16 chunks per wave, 4 waves per CU) and (2) keep that chain at two LDS round trips:
//   bkt[slot >> 2]  -> rank of the first present symbol that covers the 4-slot bucket        (1 KiB / chunk)
//   symt[rank..+3]  -> the (at most 4) symbols a bucket can hold, cum << 20 | freq << 8 | sym  (1 KiB / chunk)
// and a branch-free pick of the entry with the largest cum <= slot.  The payload bytes a step may need
// (4 big-endian 16-bit items at the shared pointer) are read from a 256-byte LDS ring at the top of the step,
// off the chain; the ring is topped up 64 bytes at a time by the chunk's own 4 lanes: every 8 steps each lane
// loads the next 16 bytes behind the filled part unconditionally (so the loads are plain, pipelined loads) and
// commits the data loaded 8 steps earlier if the consumer has made room for it.
constexpr int DCH = 16;              // chunks per wave (4 lanes each)
constexpr u32 RB = 256;              // ring bytes per chunk
constexpr u32 RQ = 64;               // refill quantum (16 bytes per lane)
constexpr u32 RSTRIDE = RB / 4 + 2;  // ring words per chunk incl. 8 mirrored bytes
constexpr u32 CHECK_STEPS = 8;       // <= 8 bytes consumed per step -> <= RQ per interval
constexpr u32 SYM_STRIDE = 260;
constexpr u32 OST = 32;              // steps of output staged in LDS per chunk (dwords)

// a store that is a global store whatever the compiler knows about the pointer (a flat store also counts as an LDS operation)
__device__ __forceinline__ void st_global_u32(u8* p, u32 v)
{
#ifdef KNZ_EMU
    *reinterpret_cast<u32*>(p) = v;
#else
    *reinterpret_cast<__attribute__((address_space(1))) u32*>(reinterpret_cast<uintptr_t>(p)) = v;
#endif
}

__device__ __forceinline__ void st_global_u32x4(u8* p, u32x4 v)
{
#ifdef KNZ_EMU
    *reinterpret_cast<u32x4*>(p) = v;
#else
    *reinterpret_cast<__attribute__((address_space(1))) u32x4*>(reinterpret_cast<uintptr_t>(p)) = v;
#endif
}
// every load issued so far has arrived (before a burst of stores: one counter for loads and stores, counted in issue order)
// (KNZ_LOADS_DONE: common.hpp)

template <int K>
__device__ __forceinline__ u32 quad_bcast(u32 v)
{
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xF, 0xF, false);
}

__device__ __forceinline__ u32 quad_transpose_word(u32 acc, int j)
{
    // lanes of a quad hold acc = symbols of steps 0..3 (byte k = step k); lane j returns the dword
    // {state0, state1, state2, state3} of step j, i.e. output bytes [4j, 4j+4)
    const u32 a0 = quad_bcast<0>(acc), a1 = quad_bcast<1>(acc), a2 = quad_bcast<2>(acc), a3 = quad_bcast<3>(acc);
    const u32 sh = 8u * (u32)j;
    return ((a3 >> sh) & 0xFF) | (((a2 >> sh) & 0xFF) << 8) | (((a1 >> sh) & 0xFF) << 16) | (((a0 >> sh) & 0xFF) << 24);
}

__global__ __launch_bounds__(64) void k_ans0_decode(BitSrc src, DecBlock* __restrict__ blocks, int maxChunks, int nSlots,
                                                    const AnsDecChunk* __restrict__ chunks, u8* const* __restrict__ outPtr)
{
    __shared__ u8 bktAll[DCH * 1024];                          // (slot >> 2) -> rank of first covering symbol
    __shared__ u32 symAll[DCH * SYM_STRIDE];                   // by rank: cum << 20 | freq << 8 | sym
    __shared__ u32 ringAll[DCH * RSTRIDE];                     // payload as 16-bit items (value form)
    __shared__ u16 cumArr[260];
    __shared__ int chunkErr[DCH];
    __shared__ __attribute__((aligned(16))) u32 stageAll[DCH * OST];                        // output of OST steps per chunk
    const int lane = lane_id();
    const int slotBase = blockIdx.x * DCH;
    if (lane < DCH) chunkErr[lane] = 0;
    __syncthreads();

    // ---- phase 1: tables (ANSRangeDecoder.cpp:113-171) and simple chunk kinds
    for (int gg = 0; gg < DCH; gg++) {
        const int slot = slotBase + gg;
        if (slot >= nSlots) break;
        const AnsDecChunk& c = chunks[slot];
        const int b = slot / maxChunks;
        const int ci = slot - b * maxChunks;
        if (c.kind == 3 || blocks[b].error) continue;
        u8* dst = outPtr[b] + (size_t)ci * ENT_CHUNK;
        const u32 preLen = blocks[b].preLen;
        if (c.kind == 2) {
            for (u32 i = lane; i < c.sz; i += 64) dst[i] = (u8)peek_bits(src, c.payloadBit + 8ull * i, 8);
            continue;
        }
        const u32 n = (preLen - (u32)ci * ENT_CHUNK < ENT_CHUNK) ? (preLen - (u32)ci * ENT_CHUNK) : ENT_CHUNK;
        if (c.kind == 1) {
            const u32 v4 = 0x01010101u * c.sym;
            if ((reinterpret_cast<uintptr_t>(dst) & 3) == 0) {
                for (u32 i = lane; i < (n >> 2); i += 64) reinterpret_cast<u32*>(dst)[i] = v4;
                for (u32 i = (n & ~3u) + lane; i < n; i += 64) dst[i] = c.sym;
            } else {
                for (u32 i = lane; i < n; i += 64) dst[i] = c.sym;
            }
            continue;
        }
        const u32 lr = c.lr;
        const u32 scale = 1u << lr;
        const u32 asz = c.asz;
        u32 present;
        if (asz == 256) present = 0xF;
        else {
            const u32 m = (u32)lane >> 1;
            const u64 mb = c.maskBit + 8ull * m;
            const u32 byte = (mb + 8 <= c.freqBit) ? peek_bits(src, mb, 8) : 0u;
            present = (lane & 1) ? (byte >> 4) : (byte & 0xF);
        }
        const u32 myCount = __popc(present);
        const u32 incl = wave_incl_scan(myCount);
        const u32 rank0 = incl - myCount;
        u32 r = rank0;
        const u32 chk = (asz >= 64) ? 8u : 6u;
        const u32 llr = (u32)ilog2_u32(lr) + 1u;
        u32 f[4] = { 0, 0, 0, 0 };
        u32 lsum = 0;
        int bad = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((present >> k) & 1) {
                if (r >= 1) {
                    const u32 g = (r - 1) / chk;
                    const u32 ge = c.grp[g];
                    const u32 lm = ge >> 12;
                    const u32 fv = lm ? peek_bits(src, c.freqBit + (ge & 0xFFF) + llr + (u64)((r - 1) - g * chk) * lm, lm) + 1u : 1u;
                    if (fv >= scale) bad = 1;
                    f[k] = fv;
                    lsum += fv;
                }
                r++;
            }
        }
        const u32 sumOthers = wave_sum(lsum);
        if (scale <= sumOthers) bad = 1;
        if (!bad) {
            u32 rr = rank0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if ((present >> k) & 1) { if (rr == 0) f[k] = scale - sumOthers; rr++; }
            }
        }
        if (__ballot(bad) != 0) { if (lane == 0) chunkErr[gg] = 1; continue; }
        const u32 tot = f[0] + f[1] + f[2] + f[3];
        const u32 cincl = wave_incl_scan(tot);
        u32 cum = cincl - tot;
        u32* symt = symAll + gg * SYM_STRIDE;
        {
            u32 rr = rank0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if ((present >> k) & 1) {
                    const u32 fr = f[k];
                    const u32 fclip = (fr >= scale) ? scale - 1 : fr;       // ANSRangeDecoder.hpp:47-49
                    const u32 e = (cum << 20) | (fclip << 8) | (4u * (u32)lane + (u32)k);
                    symt[rr] = e;
                    cumArr[rr] = (u16)cum;
                    if (rr + 1 == asz) { symt[rr + 1] = e; symt[rr + 2] = e; symt[rr + 3] = e; cumArr[rr + 1] = (u16)scale; }
                    cum += fr;
                    rr++;
                }
            }
        }
        __syncthreads();
        // bucket -> rank: lane fills buckets [16*lane, 16*lane+16) (+ 1024-bucket strides for completeness)
        {
            u8* bkt = bktAll + gg * 1024;
            for (u32 base = (u32)lane * 64; base < scale; base += 4096) {
                u32 lo = 0, hi = asz - 1;
                while (lo < hi) {
                    const u32 mid = (lo + hi + 1) >> 1;
                    if (cumArr[mid] <= base) lo = mid; else hi = mid - 1;
                }
                u32 s = lo;
                u32 t = base;
                u32 word = 0;
                for (u32 k = 0; k < 16; k++, t += 4) {
                    while (cumArr[s + 1] <= t) s++;                 // cumArr[asz] = scale > t
                    word |= s << (8 * (k & 3));
                    if ((k & 3) == 3) { *reinterpret_cast<u32*>(bkt + (base >> 2) + (k & ~3u)) = word; word = 0; }
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();

    // ---- phase 2
    const int g = lane >> 2;
    const int j = lane & 3;
    bool act = false;
    u32 n = 0, sz = 0, lr = ANS_LR;
    u64 payBit = 0;
    u8* dst = nullptr;
    u32 st = 0;
    {
        const int slot = slotBase + g;
        if (slot < nSlots) {
            const int b = slot / maxChunks;
            const int ci = slot - b * maxChunks;
            const AnsDecChunk& c = chunks[slot];
            if (c.kind == 0 && !blocks[b].error && !chunkErr[g]) {
                act = true;
                const u32 preLen = blocks[b].preLen;
                n = (preLen - (u32)ci * ENT_CHUNK < ENT_CHUNK) ? (preLen - (u32)ci * ENT_CHUNK) : ENT_CHUNK;
                dst = outPtr[b] + (size_t)ci * ENT_CHUNK;
                payBit = c.payloadBit; sz = c.sz; lr = c.lr;
                st = c.st[j];
            }
        }
    }
    const u64 anyAct = __ballot(act);
    if (anyAct == 0) {
        if (lane < DCH && chunkErr[lane] && slotBase + lane < nSlots) blocks[(slotBase + lane) / maxChunks].error = KNZ_ERR_PROCESS_BLOCK;
        return;
    }
    u32* ringW = ringAll + g * RSTRIDE;
    // Refill loads are branch-free so that their latency is not waited for at the load: word indices are
    // clamped to the last (possibly partial) word of the buffer -- bytes past the stream end are never consumed,
    // and an aligned dword never straddles a page.  Raw words are converted when they are stored to LDS.
    const u64 lastWord = ((src.nBytes + 3) >> 2) - 1;
    auto load5 = [&](u32 streamOff, u32 raw[5]) {
        const u64 w0 = (payBit + 8ull * streamOff) >> 5;
#pragma unroll
        for (int k = 0; k < 5; k++) { const u64 w = w0 + k; raw[k] = src.words[w < lastWord ? w : lastWord]; }
    };
    // item form of 4 stream bytes s0 s1 s2 s3 (big-endian word v): low half = s0<<8|s1, high half = s2<<8|s3
    auto store4 = [&](u32 streamOff, const u32 raw[5]) {
        const u32 sh = (u32)(payBit + 8ull * streamOff) & 31;
        const u32 wo = (streamOff & (RB - 1)) >> 2;
        u32 it[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 v = (u32)(((((u64)bswap32(raw[k]) << 32) | bswap32(raw[k + 1])) << sh) >> 32);
            it[k] = (v >> 16) | (v << 16);
            ringW[wo + k] = it[k];
        }
        if (wo == 0) { ringW[RB / 4] = it[0]; ringW[RB / 4 + 1] = it[1]; }
    };
    // initial fill: stream bytes [0, RB) of each active chunk, 64 bytes per lane
    if (act) {
        for (u32 t = 0; t < RB / RQ; t++) {
            u32 tmp[5];
            load5(t * RQ + 16u * (u32)j, tmp);
            store4(t * RQ + 16u * (u32)j, tmp);
        }
    }
    __syncthreads();

    const u32 mask = (1u << lr) - 1;
    const u32 count4 = n & ~3u;
    const u32 steps = act ? (count4 >> 2) : 0;
    const u32 maxSteps = wave_max(steps);
    const u8* bkt = bktAll + g * 1024;
    const u32* symt = symAll + g * SYM_STRIDE;
    u32 q = 0;                            // 16-bit items consumed (shared by the chunk's 4 lanes)
    u32 F = RB;                           // stream bytes [F - RB, F) are in the ring
    u32 pend[5];                          // raw words of stream bytes [F + 16j, +16), loaded one interval ago
    const u32 grpShift64 = (u32)(lane & ~3);
    const u32 higherMask = (0xFu << (j + 1)) & 0xFu;
    const bool aligned4 = act && ((reinterpret_cast<uintptr_t>(dst) & 3) == 0);

    // one decode step; returns the symbol in the low byte of `e`
    auto step = [&](u32 s) -> u32 {
        const u32 slotv = st & mask;
        const u32 rk = bkt[slotv >> 2];
        // (three aligned words and a shift instead of one 64-bit read at a 2-byte-aligned address: LDS reads return in order, the wait for
        // the bucket read above waits for this one too, and a read that is not aligned to its size costs about 40 cycles more)
        const u32 qw = (q & (RB / 2 - 1)) >> 1;
        const u32 rw0 = ringW[qw], rw1 = ringW[qw + 1], rw2 = ringW[qw + 2];
        const u32 ish = 16 * (q & 1);
        const u64 items = ((u64)(u32)((((u64)rw2 << 32) | rw1) >> ish) << 32) | (u32)((((u64)rw1 << 32) | rw0) >> ish);
        const u32 e0 = symt[rk], e1 = symt[rk + 1], e2 = symt[rk + 2], e3 = symt[rk + 3];
        const u32 key = (slotv << 20) | 0xFFFFFu;
        u32 e = e0;
        e = (e1 <= key) ? e1 : e;
        e = (e2 <= key) ? e2 : e;
        e = (e3 <= key) ? e3 : e;
        // lanes past their last step (or without a chunk) keep computing on in-bounds garbage; they never
        // flag, store or move the shared pointer
        st = __umul24((e >> 8) & 0xFFF, st >> lr) + slotv - (e >> 20);
        const bool flag = (st < ANS_TOP) && (s < steps);
        const u64 m = __ballot(flag);
        const u32 grp = (u32)(m >> grpShift64) & 0xF;
        const u32 kk = __popc(grp & higherMask);
        const u32 item = (u32)(items >> (16 * kk)) & 0xFFFF;
        st = flag ? ((st << 16) | item) : st;
        q += __popc(grp);
        return e & 0xFF;
    };
    auto put_word = [&](u32 word, u32 stepIdx) {
        if (stepIdx < steps) {
            u8* o = dst + 4 * (size_t)stepIdx;
            if (aligned4) st_global_u32(o, word);
            else { stg<u8>(o, (u8)word); stg<u8>(o + 1, (u8)(word >> 8)); stg<u8>(o + 2, (u8)(word >> 16)); stg<u8>(o + 3, (u8)(word >> 24)); }
        }
    };
    // Output is staged in LDS, 32 steps (128 bytes) per chunk (more would cost the fourth workgroup of a CU its LDS), and leaves between two intervals, BEFORE the interval's refill loads are
    // issued: loads and stores share one counter that counts in issue order, so a store inside the interval made the wait for the
    // refill data of the next interval a wait for that store to reach memory -- once per 8 steps, ~ 100 ns per step (round 5; found
    // with the order-1 coders). Now three intervals of four have no store in front of their loads, and the fourth has had 8 steps.
    u32* stgw = stageAll + g * OST;
    const bool aligned16 = act && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
    auto flush = [&](u32 base, u32 upto) {                // this chunk's output dwords [base, upto), base a multiple of OST
        KNZ_WAVE_ORDER();
        if (act) {
            const u32 lim = (upto < steps) ? upto : steps;
#pragma unroll
            for (u32 k = 0; k < OST / 16; k++) {
                const u32 d0 = base + (OST / 4) * (u32)j + 4u * k;     // lane j of the chunk writes a quarter of the staged dwords
                if (d0 >= lim) continue;
                const u32x4 v = *reinterpret_cast<const u32x4*>(&stgw[(OST / 4) * (u32)j + 4u * k]);
                if (aligned16 && d0 + 4 <= lim) st_global_u32x4(dst + 4 * (size_t)d0, v);
                else {
                    const u32 w[4] = { v.x, v.y, v.z, v.w };
                    for (u32 i = 0; i < 4; i++) put_word(w[i], d0 + i);
                }
            }
        }
        KNZ_WAVE_ORDER();
    };
    load5(F + 16u * (u32)j, pend);
    u32 s0 = 0;
    for (; s0 + CHECK_STEPS <= maxSteps; s0 += CHECK_STEPS) {
        // ring upkeep.  Every interval every lane loads the next quantum at F (plain loads, no divergence, so
        // the compiler pipelines them across the interval); one interval later the data is committed to the
        // ring if the consumer has made room for it, else it is simply loaded again.
        // (the refill data was asked for a whole interval ago: waiting for it here, on every path, costs nothing and tells the compiler
        // that no load is outstanding when the stores of a flush are issued -- else it waits for them where it reuses a register)
        KNZ_LOADS_DONE();
        if (s0 && act && F < sz && F + RQ <= 2 * q + RB) { store4(F + 16u * (u32)j, pend); F += RQ; }
        if (s0 && (s0 & (OST - 1)) == 0) flush(s0 - OST, s0);
        load5(F + 16u * (u32)j, pend);
        u32 acc = 0;
#pragma unroll
        for (u32 u = 0; u < CHECK_STEPS; u++) {
            acc |= step(s0 + u) << (8 * (u & 3));
            if (u == 3) { stgw[(s0 & (OST - 1)) + (u32)j] = quad_transpose_word(acc, j); acc = 0; }
            if (u == 7) stgw[(s0 & (OST - 1)) + 4 + (u32)j] = quad_transpose_word(acc, j);
        }
    }
    if (s0 & (OST - 1)) flush(s0 & ~(OST - 1), s0);
    else if (s0) flush(s0 - OST, s0);
    // remaining (< 8) steps; the ring still holds >= 64 unread bytes or everything up to the chunk's end
    if (s0 && act && F < sz && F + RQ <= 2 * q + RB) { store4(F + 16u * (u32)j, pend); F += RQ; }
    {
        u32 acc = 0;
        for (u32 s = s0; s < maxSteps; s++) {
            acc |= step(s) << (8 * (s & 3));
            if ((s & 3) == 3) { put_word(quad_transpose_word(acc, j), (s & ~3u) + (u32)j); acc = 0; }
        }
        if (maxSteps & 3) {
            const u32 word = quad_transpose_word(acc, j);
            const u32 stepIdx = (maxSteps & ~3u) + (u32)j;
            if (stepIdx < steps) { dst[4 * stepIdx] = (u8)word; dst[4 * stepIdx + 1] = (u8)(word >> 8); dst[4 * stepIdx + 2] = (u8)(word >> 16); dst[4 * stepIdx + 3] = (u8)(word >> 24); }
        }
    }
    if (act && j == 0) {
        const u32 p = 2 * q;
        const u32 tail = n - count4;
        for (u32 t = 0; t < tail; t++) dst[count4 + t] = (u8)peek_bits(src, payBit + 8ull * (p + t), 8);
        if (p + tail != sz) chunkErr[g] = 1;          // ANSRangeDecoder.cpp:291
    }
    __syncthreads();
    if (lane < DCH && chunkErr[lane] && slotBase + lane < nSlots) blocks[(slotBase + lane) / maxChunks].error = KNZ_ERR_PROCESS_BLOCK;
}

void launch_ans0_decode(hipStream_t s, BitSrc src, DecBlock* blocks, int nBlocks, int maxChunks, void* chunkMeta, u8* const* outPtr)
{
    AnsDecChunk* chunks = reinterpret_cast<AnsDecChunk*>(chunkMeta);
    const int nSlots = nBlocks * maxChunks;
    { KScope ks_("k_ans0_scan"); hipLaunchKernelGGL(k_ans_scan<0>, dim3(nBlocks), dim3(64), 0, s, src, blocks, nBlocks, maxChunks, chunks); }
    { KScope ks_("k_ans0_decode"); hipLaunchKernelGGL(k_ans0_decode, dim3((nSlots + DCH - 1) / DCH), dim3(64), 0, s, src, blocks, maxChunks, nSlots, chunks, outPtr); }
}

// ------------------------------------------------------------------------------------------------
// order 1 (ANSRangeDecoder.cpp:80-175 with 256 contexts, :218-292 order-1 branch)
// ------------------------------------------------------------------------------------------------
// Round 5: the tables of a chunk are the CUMULATIVE FREQUENCIES of its 256 contexts, 257 16-bit values each = 129 KiB, and live in the
// LDS of the CU that decodes the chunk. The slot tables of the rounds before (symbol | frequency | cumulative per slot: 2 MiB per chunk,
// 6-7 chunks per XCD against 4 MiB of L2, every context hot in LZ output) made every step wait for a load from beyond the L2: 245 ns per
// step, 257 ms per 4 MiB chunk. A step now finds its symbol by searching the context's row with the wave: 16 lanes per state, two levels
// of 16 (every 16th value, then the 16 values of the block that holds the slot), each one LDS read and one ballot.
constexpr u32 A1_ROW = 258;                  // 16-bit values per context row: cum[0 .. 256] (cum[256] = 1 << lr) and one of padding
constexpr u32 A1_TAB_WORDS = 256 * A1_ROW / 2;

// one wave per (chunk, context): row[s] = sum of the frequencies of the symbols below s
__global__ __launch_bounds__(64) void k_ans1_tables(BitSrc src, DecBlock* __restrict__ blocks, int chunksPerBlock, int maxChunks,
                                                    const AnsDecChunk* __restrict__ chunks, u32* __restrict__ cumTab)
{
    const int gc = blockIdx.x >> 8;
    const u32 ctx = blockIdx.x & 255;
    const int b = gc / chunksPerBlock;
    const int ci = gc - b * chunksPerBlock;
    if (blocks[b].error) return;
    const AnsDecChunk* cs = chunks + (size_t)b * maxChunks + (size_t)ci * 257;
    if (cs[256].kind != 0) return;
    const AnsDecChunk& c = cs[ctx];
    if (c.kind != 0) return;
    const int lane = lane_id();
    const u32 lr = c.lr;
    const u32 scale = 1u << lr;
    const u32 asz = c.asz;
    u32 present;
    if (asz == 256) present = 0xF;
    else {
        const u32 m = (u32)lane >> 1;
        const u64 mb = c.maskBit + 8ull * m;
        const u32 byte = (mb + 8 <= c.freqBit) ? peek_bits(src, mb, 8) : 0u;
        present = (lane & 1) ? (byte >> 4) : (byte & 0xF);
    }
    const u32 myCount = __popc(present);
    const u32 incl = wave_incl_scan(myCount);
    const u32 rank0 = incl - myCount;
    u32 r = rank0;
    const u32 chk = (asz >= 64) ? 8u : 6u;
    const u32 llr = (u32)ilog2_u32(lr) + 1u;
    u32 f[4] = { 0, 0, 0, 0 };
    u32 lsum = 0;
    int bad = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if ((present >> k) & 1) {
            if (r >= 1) {
                const u32 g = (r - 1) / chk;
                const u32 ge = c.grp[g];
                const u32 lm = ge >> 12;
                const u32 fv = lm ? peek_bits(src, c.freqBit + (ge & 0xFFF) + llr + (u64)((r - 1) - g * chk) * lm, lm) + 1u : 1u;
                if (fv >= scale) bad = 1;
                f[k] = fv;
                lsum += fv;
            }
            r++;
        }
    }
    const u32 sumOthers = wave_sum(lsum);
    if (scale <= sumOthers) bad = 1;
    if (__ballot(bad) != 0) { if (lane == 0) blocks[b].error = KNZ_ERR_PROCESS_BLOCK; return; }
    {
        u32 rr = rank0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((present >> k) & 1) { if (rr == 0) f[k] = scale - sumOthers; rr++; }
        }
    }
    const u32 tot = f[0] + f[1] + f[2] + f[3];
    const u32 cincl = wave_incl_scan(tot);
    const u32 c0 = cincl - tot, c1 = c0 + f[0], c2 = c1 + f[1], c3 = c2 + f[2];
    u32* row = cumTab + (size_t)gc * A1_TAB_WORDS + ctx * (A1_ROW / 2);
    row[2 * lane] = c0 | (c1 << 16);
    row[2 * lane + 1] = c2 | (c3 << 16);
    if (lane == 63) row[128] = scale;            // (= c3 + f[3])
}

constexpr u32 A1_RN = 1024;                  // ring items (16-bit)
constexpr u32 A1_INTERVAL = 64;              // steps between ring checks (<= 4 items per step)

// one wave per chunk: lanes 16 j .. 16 j + 15 are state j (state j decodes quarter j forwards, context = previous symbol of the same
// quarter); every lane of a group holds the group's state, and all lanes keep the payload ring filled
__global__ __launch_bounds__(64) void k_ans1_decode(BitSrc src, DecBlock* __restrict__ blocks, int chunksPerBlock, int maxChunks,
                                                    const AnsDecChunk* __restrict__ chunks, const u32* __restrict__ cumTab,
                                                    u8* const* __restrict__ outPtr, int nBlocks)
{
    const int ci = (int)blockIdx.x / nBlocks;
    const int b = (int)blockIdx.x - ci * nBlocks;
    const int gc = b * chunksPerBlock + ci;
    const int lane = lane_id();
    if (blocks[b].error) return;
    const AnsDecChunk* cs = chunks + (size_t)b * maxChunks + (size_t)ci * 257;
    if (ci == 0 && cs[0].kind == 2) {
        // raw block (<= 32 bytes or copy block)
        u8* dst = outPtr[b];
        for (u32 i = lane; i < cs[0].sz; i += 64) dst[i] = (u8)peek_bits(src, cs[0].payloadBit + 8ull * i, 8);
        return;
    }
    const AnsDecChunk& rec = cs[256];
    if (rec.kind != 0) return;
    const u32 preLen = blocks[b].preLen;
    const u32 n = (preLen - (u32)ci * ANS1_CHUNK < ANS1_CHUNK) ? (preLen - (u32)ci * ANS1_CHUNK) : ANS1_CHUNK;
    u8* dst = outPtr[b] + (size_t)ci * ANS1_CHUNK;
    const u32 count4 = n & ~3u;
    const u32 quarter = count4 >> 2;
    const u32 lr = rec.lr;
    const u32 mask = (1u << lr) - 1;
    const u64 payBit = rec.payloadBit;
    const u32 sz = rec.sz;

    __shared__ u32 cumL[A1_TAB_WORDS];           // 129 KiB: the chunk's 256 rows
    {
        const u32* tab = cumTab + (size_t)gc * A1_TAB_WORDS;
        for (u32 i = (u32)lane * 4; i < A1_TAB_WORDS; i += 256)
            *reinterpret_cast<uint4*>(&cumL[i]) = *reinterpret_cast<const uint4*>(&tab[i]);       // (A1_TAB_WORDS is a multiple of 4)
    }
    const u16* cum16 = reinterpret_cast<const u16*>(cumL);

    __shared__ u32 ring[A1_RN / 2 + 4];          // 16-bit items in value form, 2 per word, + 4 mirrored items (+ 2 words a read may touch)
    const u64 lastWord = ((src.nBytes + 3) >> 2) - 1;
    // lane loads stream bytes [off + 16*lane, +16) -> 8 items
    auto fill_half = [&](u32 streamOff) {
        const u32 off = streamOff + 16u * (u32)lane;
        const u64 bit = payBit + 8ull * off;
        const u64 w0 = bit >> 5;
        const u32 sh = (u32)bit & 31;
        u32 raw[5];
#pragma unroll
        for (int k = 0; k < 5; k++) { const u64 w = w0 + k; raw[k] = bswap32(src.words[w < lastWord ? w : lastWord]); }
        const u32 wo = (off & (2 * A1_RN - 1)) >> 2;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 v = (u32)(((((u64)raw[k] << 32) | raw[k + 1]) << sh) >> 32);
            const u32 it = (v >> 16) | (v << 16);
            ring[wo + k] = it;
            if (wo + k < 2) ring[A1_RN / 2 + wo + k] = it;
        }
    };
    fill_half(0);
    fill_half(A1_RN);                            // bytes: one half = A1_RN/2 items = A1_RN bytes
    __syncthreads();
    u32 base = 0;                                // item index of ring slot 0's current content start (multiple of RN/2)
    u32 q = 0;                                   // items consumed (the same in every lane)
    const u32 g = (u32)lane >> 4, l16 = (u32)lane & 15;
    const u32 gsh = 16 * g;
    u32 st = rec.st[g];
    const u64 GROUPS = 0x0001000100010001ull;    // one lane of every group
    const u64 higher = GROUPS & ~((2ull << gsh) - 1ull);           // the states above this one take their items first
    const u32 higherLo = (u32)higher, higherHi = (u32)(higher >> 32);
    u32 row = 0;                                 // first value of the context's row (context 0 at the start of a quarter)
    // Output goes through LDS: the symbols of 64 steps of the four states are staged (64 bytes each) and written by the whole wave
    // between two stretches (a store inside the step loop would sit in the same counter as the loads of the ring refill).
    __shared__ u32 stage[4 * (A1_INTERVAL / 4)];
    static_assert(sizeof(cumL) + sizeof(ring) + sizeof(stage) + 1024 <= KNZ_LDS_BYTES, "k_ans1_decode: table + ring + stage must fit the LDS of one gfx950 workgroup");
    u8* const stage8 = reinterpret_cast<u8*>(stage) + g * A1_INTERVAL;
    u8* const q0 = dst;
    const bool alignedAll = ((reinterpret_cast<uintptr_t>(dst) | quarter) & 3) == 0;
    u32 cC = cum16[16 * l16];                    // every 16th value of the row: the coarse level of the first step
    for (u32 s0 = 0; s0 < quarter; s0 += A1_INTERVAL) {
        // ring upkeep (uniform): once the consumer is in the upper half of what the ring holds, replace the lower half
        if (q - base >= A1_RN / 2) {
            __syncthreads();
            fill_half(2 * (base + A1_RN));       // next half in stream order lands on the slots of the oldest half
            base += A1_RN / 2;
            __syncthreads();
        }
        const u32 cnt = (quarter - s0 < A1_INTERVAL) ? (quarter - s0) : A1_INTERVAL;
        // One wave issues an instruction every four cycles whatever the instruction is: the step is written for few instructions. Both
        // levels test ONE bound (rows do not decrease, so "first value above the slot" finds the block and then the symbol).
        for (u32 t = 0; t < cnt; t++) {
            const u32 slotv = st & mask;
            // the four items this step may take (which one depends on the other states' flags), read before they are needed
            // (three aligned words and a shift: an LDS read that is not aligned to its size costs about 40 cycles more)
            const u32 qw = (q & (A1_RN - 1)) >> 1;
            u32 w0 = ring[qw], w1 = ring[qw + 1], w2 = ring[qw + 2];
            // coarse: blocks of 16 symbols that begin at or below the slot (cum[0] = 0: at least one), the last of them holds the symbol
            const u64 bC = __ballot(cC <= slotv);
            const u32 nblk = (u32)__popc((u32)(bC >> gsh) & 0xFFFFu);
            // fine: the first symbol of that block whose interval ends above the slot (intervals of absent symbols are empty)
            const u32 hi = cum16[row + 16 * nblk + l16 - 15];
            const u64 bF = __ballot(hi > slotv);
            const u32 sym = (16 * nblk - 17 + (u32)__ffs((int)(((u32)(bF >> gsh) & 0xFFFFu) | 0x10000u))) & 0xFFu;     // (none: a damaged table; any symbol will do)
            // its interval for every lane of the group, and the coarse level of the NEXT step (it depends on the symbol alone)
            const u32 cl = cum16[row + sym], ch = cum16[row + sym + 1];     // (two 16-bit reads: one unaligned 32-bit read is slower)
            row = __umul24(sym, A1_ROW);
            cC = cum16[row + 16 * l16];
            const u32 ish = 16 * (q & 1);
            KNZ_KEEP3(w0, w1, w2);                                       // (read by now, not where one of them is picked: behind a branch)
            const u32 itLo = (u32)((((u64)w1 << 32) | w0) >> ish), itHi = (u32)((((u64)w2 << 32) | w1) >> ish);
            u32 fr = ch - cl;
            fr = (fr > mask) ? mask : fr;                                // ANSRangeDecoder.hpp:47-49 (a frequency of 1 << lr is kept as (1 << lr) - 1)
            st = __umul24(fr, st >> lr) + (slotv - cl);                  // (fr < 2^11, st >> lr < 2^21)
            const bool flag = st < ANS_TOP;
            const u64 bR = __ballot(flag);
            const u32 kk = (u32)__popc((u32)bR & higherLo) + (u32)__popc((u32)(bR >> 32) & higherHi);
            const u32 it = (u32)((((u64)itHi << 32) | itLo) >> (16 * kk)) & 0xFFFFu;
            st = flag ? ((st << 16) | it) : st;
            q += (u32)__popcll(bR & GROUPS);
            if (l16 == 0) stage8[t] = (u8)sym;                             // (measured: the branch is cheaper than a write by all lanes)
        }
        __syncthreads();
        {
            const u32 full = cnt >> 2;                                // whole dwords per state in this stretch
            const u32 jj = (u32)lane >> 4, dw = (u32)lane & 15;
            if (dw < full) {
                const u32 v = stage[jj * (A1_INTERVAL / 4) + dw];
                u8* o = q0 + (size_t)jj * quarter + s0 + 4 * dw;
                if (alignedAll) st_global_u32(o, v);
                else { o[0] = (u8)v; o[1] = (u8)(v >> 8); o[2] = (u8)(v >> 16); o[3] = (u8)(v >> 24); }
            }
            // the last bytes of a quarter that is not a multiple of four
            if (dw == 0 && (cnt & 3)) {
                u8* o = q0 + (size_t)jj * quarter + s0 + 4 * full;
                for (u32 k = 0; k < (cnt & 3); k++) o[k] = reinterpret_cast<const u8*>(stage)[jj * A1_INTERVAL + 4 * full + k];
            }
        }
        __syncthreads();
    }
    if (lane == 0) {
        const u32 p = 2 * q;
        const u32 tail = n - count4;
        for (u32 t = 0; t < tail; t++) dst[count4 + t] = (u8)peek_bits(src, payBit + 8ull * (p + t), 8);
        if (p + tail != sz) blocks[b].error = KNZ_ERR_PROCESS_BLOCK;          // ANSRangeDecoder.cpp:291
    }
}

size_t ans1_meta_bytes(size_t nChunks) { return nChunks * 257 * sizeof(AnsDecChunk); }
size_t ans1_slottab_bytes(size_t nChunks) { return nChunks * A1_TAB_WORDS * sizeof(u32) + 64; }

void launch_ans1_decode(hipStream_t s, BitSrc src, DecBlock* blocks, int nBlocks, int chunksPerBlock, const Ans1DecWs& ws, u8* const* outPtr)
{
    AnsDecChunk* chunks = reinterpret_cast<AnsDecChunk*>(ws.meta);
    const int maxChunks = chunksPerBlock * 257;
    const int nCh = nBlocks * chunksPerBlock;
    { KScope ks_("k_ans1_scan"); hipLaunchKernelGGL(k_ans_scan<1>, dim3(nBlocks), dim3(64), 0, s, src, blocks, nBlocks, maxChunks, chunks); }
    (void)hipMemsetAsync(ws.slotTab, 0, ans1_slottab_bytes((size_t)nCh), s);      // (rows of contexts a chunk does not have stay empty)
    { KScope ks_("k_ans1_tables"); hipLaunchKernelGGL(k_ans1_tables, dim3(nCh * 256), dim3(64), 0, s, src, blocks, chunksPerBlock, maxChunks, chunks, ws.slotTab); }
    { KScope ks_("k_ans1_decode"); hipLaunchKernelGGL(k_ans1_decode, dim3(nCh), dim3(64), 0, s, src, blocks, chunksPerBlock, maxChunks, chunks, ws.slotTab, outPtr, nBlocks); }
}

size_t ans0_dec_chunk_bytes() { return sizeof(AnsDecChunk); }

}  // namespace knz
