// rANS order-0 (kanzi "ANS0") on gfx950.
//
// Reference being replaced (results are bit-identical):
//   encoder  entropy/ANSRangeEncoder.cpp:158-192 (encode), :264-287 (rebuildStatistics), :83-116
//            (updateFrequencies), :119-155 (encodeHeader), :194-261 (encodeChunk),
//            entropy/ANSRangeEncoder.hpp:92-131 ; entropy/EntropyUtils.cpp:57-89,131-245
//   decoder  entropy/ANSRangeDecoder.cpp:80-175 (decodeHeader), :177-216 (decode), :218-292 (decodeChunk)
//
// Mapping (not a translation of the CPU loops):
//   k_ans0_stats   one wave per 16 KiB chunk: 4-way privatised LDS histogram from 16 B/lane coalesced
//                  loads, wave-parallel normalizeFrequencies (4 symbols per lane, shuffle reductions and
//                  prefix sums reproduce the reference's order-dependent error spreading), encoder
//                  table (reciprocals) and the bit-granular chunk header built with LDS atomicOr.
//   k_ans0_encode  4 lanes per chunk = the 4 interleaved rANS states, 16 chunks per wave. The shared
//                  backward byte pointer of the reference becomes a ballot + popcount per step.
//   k_ans0_scan    one lane per block walks the chunk headers (they are bit-granular and carry no
//                  directory) to find every chunk's payload position.
//   k_ans0_decode  4 lanes per chunk, shared forward pointer again via ballot/popcount.
#include "common.hpp"
#include "stages.hpp"

namespace knz {

constexpr u32 ANS_TOP = 1u << 15;
constexpr u32 ANS_LR = 12;
constexpr u32 ANS_SCALE = 1u << ANS_LR;

// ------------------------------------------------------------------------------------------------
// stats + header
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ans0_stats(BlockView view, int maxChunks, ChunkDesc* __restrict__ desc,
                                                   uint2* __restrict__ encTab, u8* __restrict__ tmp)
{
    const int slot = blockIdx.x;
    const int b = slot / maxChunks;
    const int ci = slot - b * maxChunks;
    const u32 len = view.len[b];
    const u32 start = (u32)ci * ENT_CHUNK;
    if (start >= len) return;
    const int lane = lane_id();
    const u8* blk = view.ptr[b] + start;
    ChunkDesc* cd = desc + slot;

    if (len <= 32) {
        // ANSRangeEncoder.cpp:160-163 : tiny block stored raw (also the <= 15 byte copy-block mode)
        if (lane == 0) {
            cd->hdrBits = 0; cd->midLen = 0; cd->trailerLen = 0; cd->aux = 0;
            cd->nPieces = 1; cd->pieceBits[0] = 8 * len; cd->piecePtr[0] = blk;
        }
        return;
    }
    const u32 n = (len - start < ENT_CHUNK) ? (len - start) : ENT_CHUNK;

    __shared__ u32 hist[4][256];
    __shared__ u32 hdrw[HDR_WORDS];
    __shared__ u32 grpMax[64];
    for (int i = lane; i < 1024; i += 64) (&hist[0][0])[i] = 0;
    for (int i = lane; i < (int)HDR_WORDS; i += 64) hdrw[i] = 0;
    grpMax[lane] = 0;
    __syncthreads();

    // ---- histogram (Global.cpp:170-221): 16 bytes per lane per iteration, 4 private copies
    const u32 n16 = n & ~15u;
    const bool aligned = ((reinterpret_cast<uintptr_t>(blk) & 15) == 0);
    if (aligned) {
        const uint4* p4 = reinterpret_cast<const uint4*>(blk);
        for (u32 i = lane; i < (n16 >> 4); i += 64) {
            const uint4 v = p4[i];
            const u32 w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                atomicAdd(&hist[0][w[k] & 0xFF], 1u);
                atomicAdd(&hist[1][(w[k] >> 8) & 0xFF], 1u);
                atomicAdd(&hist[2][(w[k] >> 16) & 0xFF], 1u);
                atomicAdd(&hist[3][w[k] >> 24], 1u);
            }
        }
    } else {
        for (u32 i = lane; i < n16; i += 64) atomicAdd(&hist[i & 3][blk[i]], 1u);
    }
    for (u32 i = n16 + lane; i < n; i += 64) atomicAdd(&hist[0][blk[i]], 1u);
    __syncthreads();

    // lane owns symbols 4*lane .. 4*lane+3
    u32 f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int s = 4 * lane + k;
        f[k] = hist[0][s] + hist[1][s] + hist[2][s] + hist[3][s];
    }
    u32 present = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) present |= (f[k] != 0 ? 1u : 0u) << k;
    const u32 myCount = __popc(present);
    const u32 inclCount = wave_incl_scan(myCount);
    const u32 asz = (u32)__shfl((int)inclCount, 63, 64);
    const u32 rankBase = inclCount - myCount;      // alphabet index of my first present symbol

    // ---- normalizeFrequencies (EntropyUtils.cpp:131-245), totalFreq = n, scale = 4096
    if (n != ANS_SCALE) {
        if (asz == 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) if (f[k]) f[k] = ANS_SCALE;
        } else {
            u32 ssum = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (f[k]) {
                    const u32 sf = f[k] * ANS_SCALE;                        // <= 2^26
                    f[k] = (sf <= n) ? 1u : (sf + (n >> 1)) / n;
                    ssum += f[k];
                }
            }
            const u32 sumScaled = wave_sum(ssum);
            if (sumScaled != ANS_SCALE) {
                // idxMax = first index holding the maximum scaled frequency
                u32 key = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const u32 kk = (f[k] << 8) | (255u - (u32)(4 * lane + k));
                    key = kk > key ? kk : key;
                }
                key = wave_max(key);
                const u32 idxMax = 255u - (key & 0xFF);
                const u32 fmax = key >> 8;
                int delta = (int)sumScaled - (int)ANS_SCALE;
                const int errThr = (int)(fmax >> 4);
                const bool ownMax = ((idxMax >> 2) == (u32)lane);
                const int km = (int)(idxMax & 3);
                if (abs(delta) <= errThr) {
                    if (ownMax) {
#pragma unroll
                        for (int k = 0; k < 4; k++) if (k == km) f[k] -= (u32)delta;
                    }
                } else {
                    if (delta < 0) {
                        delta += errThr;
                        if (ownMax) {
#pragma unroll
                            for (int k = 0; k < 4; k++) if (k == km) f[k] += (u32)errThr;
                        }
                    } else {
                        delta -= errThr;
                        if (ownMax) {
#pragma unroll
                            for (int k = 0; k < 4; k++) if (k == km) f[k] -= (u32)errThr;
                        }
                    }
                    const int inc = (delta < 0) ? 1 : -1;
                    delta = abs(delta);
                    int round = 0;
                    while ((++round < 6) && (delta > 0)) {
                        // symbols with f > 2, in alphabet order, each get one adjustment until delta runs out
                        u32 elig = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) elig |= (f[k] > 2 ? 1u : 0u) << k;
                        const u32 ec = __popc(elig);
                        const u32 incl = wave_incl_scan(ec);
                        const u32 totalElig = (u32)__shfl((int)incl, 63, 64);
                        u32 r = incl - ec;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if ((elig >> k) & 1) {
                                if (r < (u32)delta) f[k] += (u32)inc;
                                r++;
                            }
                        }
                        if (totalElig == 0) break;
                        delta -= (int)((totalElig < (u32)delta) ? totalElig : (u32)delta);
                    }
                    if (ownMax) {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (k == km) {
                                const u32 v = f[k] - (u32)delta;      // uint32 wrap as in the reference
                                f[k] = v > 1u ? v : 1u;
                            }
                        }
                    }
                }
            }
        }
    }

    // ---- chunk header bits (ANSRangeEncoder.cpp:83-155, EntropyUtils.cpp:57-89)
    u32 pos = 0;
    if (lane == 0) or_bits_words(hdrw, 0, ANS_LR - 8, 3);
    pos = 3;
    if (asz == 256) {
        pos += 2;                                   // FULL_ALPHABET(0), ALPHABET_256(0)
    } else {
        // PARTIAL_ALPHABET(1), 5 bits lastMask, masks
        const u64 anyMask = __ballot(present != 0);
        const int lastLane = 63 - __clzll((long long)anyMask);
        const u32 lastMask = (u32)lastLane >> 1;    // symbol >> 3 == lane >> 1
        if (lane == 0) {
            or_bits_words(hdrw, pos, 1, 1);
            or_bits_words(hdrw, pos + 1, lastMask, 5);
        }
        pos += 6;
        // mask byte m: low nibble from lane 2m, high nibble from lane 2m+1 ; bit (s&7) = symbol present
        const u32 other = (u32)__shfl_xor((int)present, 1, 64);
        if ((lane & 1) == 0 && ((u32)lane >> 1) <= lastMask) {
            const u32 byte = present | (other << 4);
            or_bits_words(hdrw, pos + 8 * ((u32)lane >> 1), byte, 8);
        }
        pos += 8 * (lastMask + 1);
    }

    if (asz > 1) {
        const u32 chk = (asz >= 64) ? 8u : 6u;
        const u32 llr = 4;                           // log2(12) + 1
        // per group maximum of bitlen(f-1) over alphabet indices 1..asz-1
        u32 r = rankBase;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (f[k]) {
                if (r >= 1) atomicMax(&grpMax[(r - 1) / chk], bitlen_u32(f[k] - 1));
                r++;
            }
        }
        __syncthreads();
        const u32 nGroups = (asz - 1 + chk - 1) / chk;
        u32 gbits = 0;
        if ((u32)lane < nGroups) {
            const u32 first = 1 + (u32)lane * chk;
            const u32 cnt = (first + chk <= asz) ? chk : (asz - first);
            gbits = llr + cnt * grpMax[lane];
        }
        const u32 gincl = wave_incl_scan(gbits);
        const u32 totalFreqBits = (u32)__shfl((int)gincl, 63, 64);
        const u32 goff = gincl - gbits;
        if ((u32)lane < nGroups) or_bits_words(hdrw, pos + goff, grpMax[lane], llr);
        // stash group offsets for the members
        __shared__ u32 grpOff[64];
        grpOff[lane] = goff;
        __syncthreads();
        r = rankBase;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (f[k]) {
                if (r >= 1) {
                    const u32 g = (r - 1) / chk;
                    const u32 lm = grpMax[g];
                    if (lm) or_bits_words(hdrw, pos + grpOff[g] + llr + ((r - 1) - g * chk) * lm, f[k] - 1, lm);
                }
                r++;
            }
        }
        pos += totalFreqBits;
    }
    __syncthreads();

    // ---- write header buffer + encoder table
    u32* hdrOut = reinterpret_cast<u32*>(tmp + (size_t)slot * TMP_STRIDE);
    const u32 hdrWordsUsed = (pos + 31) >> 5;
    for (u32 i = lane; i < hdrWordsUsed; i += 64) hdrOut[i] = bswap32(hdrw[i]);

    if (asz > 1) {
        // cumulative frequencies in symbol order; ANSEncSymbol::reset (ANSRangeEncoder.hpp:92-117)
        const u32 lsum = f[0] + f[1] + f[2] + f[3];
        const u32 lincl = wave_incl_scan(lsum);
        u32 cum = lincl - lsum;
        uint2* tab = encTab + (size_t)slot * 256;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 fr = f[k];
            uint2 e = make_uint2(0, 0);
            if (fr) {
                if (fr >= ANS_SCALE) fr = ANS_SCALE - 1;
                if (fr < 2) {
                    e.x = 0xFFFFFFFFu;
                    e.y = fr | (0u << 13) | ((cum + ANS_SCALE - 1) << 17);
                } else {
                    u32 shift = 0;
                    while (fr > (1u << shift)) shift++;
                    const u64 inv = (((1ull << (shift + 31)) + fr - 1) / fr) & 0xFFFFFFFFull;
                    e.x = (u32)inv;
                    e.y = fr | ((shift - 1) << 13) | (cum << 17);      // invShift - 32 = shift - 1
                }
            }
            tab[4 * lane + k] = e;
            cum += f[k];
        }
    }
    if (lane == 0) {
        cd->hdrBits = pos; cd->midLen = 0; cd->trailerLen = 0; cd->nPieces = 0; cd->aux = asz;
    }
}

// ------------------------------------------------------------------------------------------------
// encode: 16 chunks per wave, 4 lanes (= 4 rANS states) per chunk
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ans0_encode(BlockView view, int maxChunks, int nSlots, ChunkDesc* __restrict__ desc,
                                                    const uint2* __restrict__ encTab, u8* __restrict__ tmp)
{
    __shared__ uint2 tab[16][256];                   // 32 KiB
    const int lane = lane_id();
    const int g = lane >> 2;
    const int j = lane & 3;
    const int slotBase = blockIdx.x * 16;
    const int slot = slotBase + g;

    bool act = false;
    u32 n = 0;
    const u8* blk = nullptr;
    if (slot < nSlots) {
        const int b = slot / maxChunks;
        const int ci = slot - b * maxChunks;
        const u32 len = view.len[b];
        const u32 start = (u32)ci * ENT_CHUNK;
        if (len > 32 && start < len && desc[slot].aux > 1) {
            act = true;
            n = (len - start < ENT_CHUNK) ? (len - start) : ENT_CHUNK;
            blk = view.ptr[b] + start;
        }
    }
    // cooperative table load (all 16 groups)
    const u64 actMask = __ballot(act);
    for (int gg = 0; gg < 16; gg++) {
        if (!((actMask >> (4 * gg)) & 1)) continue;
        const uint2* src = encTab + (size_t)(slotBase + gg) * 256;
#pragma unroll
        for (int k = 0; k < 4; k++) tab[gg][lane + 64 * k] = src[lane + 64 * k];
    }
    __syncthreads();
    if (actMask == 0) return;

    const u32 end4 = n & ~3u;
    const u32 tail = n - end4;
    u8* pay = tmp + (size_t)slot * TMP_STRIDE + HDR_BYTES;
    // exclusive end of the payload area; shifted by one when the raw tail is odd so that every
    // 16-bit emission lands on an even address
    u32 top = PAY_BYTES - (tail & 1);
    if (act && j == 0) {
        for (u32 t = 0; t < tail; t++) pay[top - tail + t] = blk[end4 + t];   // ANSRangeEncoder.cpp:204-205
    }
    u32 q = top - tail;                               // current (exclusive) low end of emitted bytes
    u32 st = ANS_TOP;
    const u32 steps = act ? (end4 >> 2) : 0;
    const u32 maxSteps = wave_max(steps);
    const u32 jsh = 8u * (3u - (u32)j);
    const uint2* mytab = tab[g];
    const u32 grpShift = (u32)(lane & ~3);
    const u32 lowMask = (1u << j) - 1u;

    for (u32 s = 0; s < maxSteps; s++) {
        const bool on = s < steps;
        u32 sym = 0;
        if (on) {
            const u32 w = *reinterpret_cast<const u32*>(blk + end4 - 4 - 4 * s);
            sym = (w >> jsh) & 0xFF;
        }
        const uint2 e = mytab[sym];
        const u32 fr = e.y & 0x1FFF;
        const u32 sh = (e.y >> 13) & 0xF;
        const u32 bias = e.y >> 17;
        const bool flag = on && (st >= (fr << 19));
        const u64 m = __ballot(flag);
        const u32 grp = (u32)(m >> grpShift) & 0xF;
        if (flag) {
            const u32 before = __popc(grp & lowMask);
            const u32 a = q - 2 * (before + 1);
            *reinterpret_cast<u16*>(pay + a) = (u16)(((st >> 8) & 0xFF) | ((st & 0xFF) << 8));
            st >>= 16;
        }
        q -= 2 * __popc(grp);
        if (on) {
            const u32 qd = __umulhi(st, e.x) >> sh;
            st = st + bias + qd * (ANS_SCALE - fr);
        }
    }

    // varint(size) + 4 states -> mid ; payload piece
    const u32 s0 = (u32)__shfl((int)st, (g << 2) + 0, 64);
    const u32 s1 = (u32)__shfl((int)st, (g << 2) + 1, 64);
    const u32 s2 = (u32)__shfl((int)st, (g << 2) + 2, 64);
    const u32 s3 = (u32)__shfl((int)st, (g << 2) + 3, 64);
    if (act && j == 0) {
        ChunkDesc* cd = desc + slot;
        const u32 sz = top - q;
        u8 mid[24];
        u32 ml = 0;
        u32 v = sz;
        while (v >= 128) { mid[ml++] = (u8)(0x80 | (v & 0x7F)); v >>= 7; }
        mid[ml++] = (u8)v;
        const u32 sts[4] = { s0, s1, s2, s3 };
        for (int k = 0; k < 4; k++) {
            mid[ml++] = (u8)(sts[k] >> 24); mid[ml++] = (u8)(sts[k] >> 16);
            mid[ml++] = (u8)(sts[k] >> 8);  mid[ml++] = (u8)sts[k];
        }
        u32 mw[6] = { 0, 0, 0, 0, 0, 0 };
        for (u32 i = 0; i < ml; i++) mw[i >> 2] |= (u32)mid[i] << (8 * (i & 3));
        for (int i = 0; i < 6; i++) cd->mid[i] = mw[i];
        cd->midLen = ml;
        if (sz) { cd->nPieces = 1; cd->pieceBits[0] = 8 * sz; cd->piecePtr[0] = pay + q; }
    }
}


void launch_ans0_encode(hipStream_t s, BlockView view, int nBlocks, int maxChunks, ChunkDesc* desc, uint2* encTab, u8* tmp)
{
    const int nSlots = nBlocks * maxChunks;
    { KScope ks_("k_ans0_stats"); hipLaunchKernelGGL(k_ans0_stats, dim3(nSlots), dim3(64), 0, s, view, maxChunks, desc, encTab, tmp); }
    { KScope ks_("k_ans0_encode"); hipLaunchKernelGGL(k_ans0_encode, dim3((nSlots + 15) / 16), dim3(64), 0, s, view, maxChunks, nSlots, desc, encTab, tmp); }
}

}  // namespace knz
