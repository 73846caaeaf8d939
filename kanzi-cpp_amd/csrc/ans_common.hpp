// rANS table construction shared by the order-0 (ans.hip) and order-1 (ans1.hip) encoders.
#pragma once
#include "common.hpp"

namespace knz {

// ------------------------------------------------------------------------------------------------
// pieces shared by order 0 (one table per 16 KiB chunk) and order 1 (one table per context of a 4 MiB chunk);
// one wave, lane owns symbols 4*lane .. 4*lane+3
// ------------------------------------------------------------------------------------------------
template <u32 LR>
__device__ void ans_normalize(int lane, u32 f[4], u32 n, u32 asz)
{
    constexpr u32 SCALE = 1u << LR;
    // normalizeFrequencies (EntropyUtils.cpp:131-245), totalFreq = n, scale = 2^LR
    if (n != SCALE) {
        if (asz == 1) {
#pragma unroll
            for (int k = 0; k < 4; k++) if (f[k]) f[k] = SCALE;
        } else {
            u32 ssum = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (f[k]) {
                    if (n <= (1u << (31 - LR))) {                       // 32-bit arithmetic is exact here
                        const u32 sf = f[k] * SCALE;
                        f[k] = (sf <= n) ? 1u : (sf + (n >> 1)) / n;
                    } else {
                        const u64 sf = (u64)f[k] * SCALE;
                        f[k] = (sf <= (u64)n) ? 1u : (u32)((sf + (n >> 1)) / n);
                    }
                    ssum += f[k];
                }
            }
            const u32 sumScaled = wave_sum(ssum);
            if (sumScaled != SCALE) {
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
                int delta = (int)sumScaled - (int)SCALE;
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

}

// Appends the alphabet + frequency groups of one table to the LDS bit buffer `hdrw` at bit `pos`
// (ANSRangeEncoder.cpp:119-155, EntropyUtils.cpp:57-89); returns the new position.
template <u32 LR>
__device__ u32 ans_header_bits(int lane, const u32 f[4], u32 present, u32 asz, u32 rankBase, bool lrPrefix,
                               u32* hdrw, u32* grpMax, u32* grpOff, u32 pos)
{
    if (lrPrefix) { if (lane == 0) or_bits_words(hdrw, pos, LR - 8, 3); pos += 3; }
    if (asz == 0) {
        if (lane == 0) or_bits_words(hdrw, pos, 1, 2);   // FULL_ALPHABET(0), ALPHABET_0(1)
        pos += 2;
    } else if (asz == 256) {
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

    return pos;
}

// ANSEncSymbol::reset for the 4 symbols of every lane (ANSRangeEncoder.hpp:92-117)
template <u32 LR>
__device__ void ans_enc_table(int lane, const u32 f[4], uint2* tab)
{
    constexpr u32 SCALE = 1u << LR;
    {
        // cumulative frequencies in symbol order; ANSEncSymbol::reset (ANSRangeEncoder.hpp:92-117)
        const u32 lsum = f[0] + f[1] + f[2] + f[3];
        const u32 lincl = wave_incl_scan(lsum);
        u32 cum = lincl - lsum;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 fr = f[k];
            uint2 e = make_uint2(0, 0);
            if (fr) {
                if (fr >= SCALE) fr = SCALE - 1;
                if (fr < 2) {
                    e.x = 0xFFFFFFFFu;
                    e.y = fr | (0u << 13) | ((cum + SCALE - 1) << 17);
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
}

}  // namespace knz
