// Canonical Huffman (<= 12-bit codes, 16 KiB chunks, 4 fragments per chunk) on gfx950.
//
// Reference being replaced (bit-identical results):
//   encoder  entropy/HuffmanEncoder.cpp:58-126 (updateFrequencies), :129-215 (limitCodeLengths),
//            :219-300 (computeCodeLengths, Moffat-Katajainen), :304-344 (encode), :348-421 (encodeChunk)
//            entropy/HuffmanCommon.cpp:29-63 (generateCanonicalCodes), entropy/ExpGolombEncoder.hpp:51-62
//   decoder  entropy/HuffmanDecoder.cpp:65-108 (readLengths), :111-140 (buildDecodingTable),
//            :156-201 (decodeV6), :204-347 (decodeChunk)
//
// Mapping:
//   k_huff_encode  one wave per 16 KiB chunk: LDS histogram, rank-by-counting sort of (freq<<8|sym),
//                  code lengths / canonical codes (the O(256) sequential part runs on lane 0 in LDS),
//                  then the 4 fragments are packed by the whole wave: 16 symbols per lane, wave prefix
//                  sum of code lengths -> bit offsets -> LDS atomicOr -> coalesced store.
//   k_huff_scan    one lane per block walks chunk headers (alphabet, Exp-Golomb length deltas, 4 var-ints)
//   k_huff_decode  8 chunks per wave: 4096-entry decode tables in LDS, one lane per fragment.
#include "common.hpp"
#include "stages.hpp"

namespace knz {

constexpr int HUF_MAX_LEN = 12;
constexpr u32 HUF_FRAG_STRIDE = 8192;      // bytes reserved per fragment in the chunk staging area (max 6144 used)
constexpr u32 HUF_FRAG_WORDS = 1544;       // 4096 symbols * 12 bits = 1536 words (+ slack)

// ---- helpers shared by lane-0 serial code (arrays in LDS) -------------------------------------
// EntropyUtils::normalizeFrequencies (entropy/EntropyUtils.cpp:131-245) in its general form, used
// only by the rare limitCodeLengths fallback with length = count, scale = 2048.
__device__ int normalize_serial(u32* freqs, u32* alphabet, int length, u32 totalFreq, u32 scale)
{
    if (length == 0 || totalFreq == 0) return 0;
    int alphabetSize = 0;
    if (totalFreq == scale) {
        for (int i = 0; i < 256; i++) if (freqs[i] != 0) alphabet[alphabetSize++] = (u32)i;
        return alphabetSize;
    }
    u32 sumScaledFreq = 0, sumFreq = 0;
    int idxMax = 0;
    for (int i = 0; i < length; i++) {
        alphabet[i] = 0;
        const u32 f = freqs[i];
        if (f == 0) continue;
        alphabet[alphabetSize++] = (u32)i;
        const long long sf = (long long)f * (long long)scale;
        const u32 scaledFreq = (sf <= (long long)totalFreq) ? 1u : (u32)((sf + ((long long)totalFreq >> 1)) / (long long)totalFreq);
        sumScaledFreq += scaledFreq;
        freqs[i] = scaledFreq;
        sumFreq += f;
        idxMax = (scaledFreq > freqs[idxMax]) ? i : idxMax;
        if (sumFreq >= totalFreq) break;
    }
    if (alphabetSize == 0) return 0;
    if (alphabetSize == 1) { freqs[alphabet[0]] = scale; return 1; }
    if (sumScaledFreq == scale) return alphabetSize;
    int delta = (int)(sumScaledFreq - scale);
    const int errThr = (int)freqs[idxMax] >> 4;
    if (abs(delta) <= errThr) { freqs[idxMax] -= (u32)delta; return alphabetSize; }
    if (delta < 0) { delta += errThr; freqs[idxMax] += (u32)errThr; }
    else { delta -= errThr; freqs[idxMax] -= (u32)errThr; }
    const int inc = (delta < 0) ? 1 : -1;
    delta = abs(delta);
    int round = 0;
    while ((++round < 6) && (delta > 0)) {
        int adjustments = 0;
        for (int i = 0; i < alphabetSize; i++) {
            const int idx = (int)alphabet[i];
            if (freqs[idx] <= 2) continue;
            freqs[idx] += (u32)inc;
            adjustments++;
            delta--;
            if (delta == 0) break;
        }
        if (adjustments == 0) break;
    }
    const u32 v = freqs[idxMax] - (u32)delta;
    freqs[idxMax] = v > 1u ? v : 1u;
    return alphabetSize;
}

// In-place Moffat-Katajainen (HuffmanEncoder.cpp:246-300). data[0..n) sorted ascending frequencies.
__device__ u32 mk_code_lengths(u32* data, int n)
{
    for (int s = 0, r = 0, t = 0; t < n - 1; t++) {
        u32 sum = 0;
        for (int i = 0; i < 2; i++) {
            if ((s >= n) || ((r < t) && (data[r] < data[s]))) {
                sum += data[r];
                data[r] = (u32)t;
                r++;
                continue;
            }
            sum += data[s];
            if (s > t) data[s] = 0;
            s++;
        }
        data[t] = sum;
    }
    if (n < 2) return 0;
    u32 topLevel = (u32)n - 2;
    u32 depth = 1;
    u32 totalNodesAtLevel = 2;
    int nn = n;
    while (nn > 0) {
        u32 k = topLevel;
        while ((k != 0) && (data[k - 1] >= topLevel)) k--;
        const int internalNodesAtLevel = (int)(topLevel - k);
        const int leavesAtLevel = (int)totalNodesAtLevel - internalNodesAtLevel;
        for (int j = 0; j < leavesAtLevel; j++) data[--nn] = depth;
        totalNodesAtLevel = (u32)internalNodesAtLevel << 1;
        topLevel = k;
        depth++;
    }
    return depth - 1;
}

// Signed Exp-Golomb of an int8 delta (ExpGolombEncoder.hpp:51-62): value and bit count.
__device__ __forceinline__ void eg_signed(int v, u32& code, u32& nbits)
{
    if (v == 0) { code = 1; nbits = 1; return; }
    const u32 m = (u32)(v < 0 ? -v : v);
    const u32 e = m + 1;
    const int L = ilog2_u32(e);
    code = (e << 1) | (v < 0 ? 1u : 0u);
    nbits = (u32)(2 * L + 2);
}

// ------------------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_huff_encode(BlockView view, int maxChunks, ChunkDesc* __restrict__ desc,
                                                    u8* __restrict__ tmp)
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
    const u32 n = (len - start < ENT_CHUNK) ? (len - start) : ENT_CHUNK;
    const bool copyBlock = false;
    (void)copyBlock;

    if (n < 32) {
        // HuffmanEncoder.cpp:326-329 : small chunk stored raw
        if (lane == 0) {
            cd->hdrBits = 0; cd->midLen = 0; cd->trailerLen = 0; cd->aux = 0;
            cd->nPieces = 1; cd->pieceBits[0] = 8 * n; cd->piecePtr[0] = blk;
        }
        return;
    }

    __shared__ u32 hist[8][264];     // 8 lane-selected copies, rows 8 banks apart (see k_ans0_stats)
    __shared__ u32 hdrw[HDR_WORDS];
    __shared__ u32 keys[256];        // (freq << 8) | sym of present symbols, then sorted
    __shared__ u32 sorted[256];
    __shared__ u32 work[256];        // Moffat-Katajainen working array
    __shared__ u32 alpha[256];
    __shared__ u16 sizes[256];
    __shared__ u16 codes[256];       // (len << 12) | code
    __shared__ u32 fragw[HUF_FRAG_WORDS];
    __shared__ u32 sh_maxCodeLen;

    for (int i = lane; i < 8 * 264; i += 64) (&hist[0][0])[i] = 0;
    for (int i = lane; i < (int)HDR_WORDS; i += 64) hdrw[i] = 0;
    __syncthreads();

    // ---- histogram
    u32* myHist = hist[lane & 7];
    const u32 n16 = n & ~15u;
    const bool aligned = ((reinterpret_cast<uintptr_t>(blk) & 15) == 0);
    if (aligned) {
        const uint4* p4 = reinterpret_cast<const uint4*>(blk);
        for (u32 i = lane; i < (n16 >> 4); i += 64) {
            const uint4 v = p4[i];
            const u32 w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                atomicAdd(&myHist[w[k] & 0xFF], 1u);
                atomicAdd(&myHist[(w[k] >> 8) & 0xFF], 1u);
                atomicAdd(&myHist[(w[k] >> 16) & 0xFF], 1u);
                atomicAdd(&myHist[w[k] >> 24], 1u);
            }
        }
    } else {
        for (u32 i = lane; i < n16; i += 64) atomicAdd(&myHist[blk[i]], 1u);
    }
    for (u32 i = n16 + lane; i < n; i += 64) atomicAdd(&myHist[blk[i]], 1u);
    __syncthreads();

    u32 f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int s = 4 * lane + k;
        u32 acc = 0;
#pragma unroll
        for (int c = 0; c < 8; c++) acc += hist[c][s];
        f[k] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) hist[0][4 * lane + k] = f[k];      // keep totals for the fallback path
    u32 present = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) present |= (f[k] != 0 ? 1u : 0u) << k;
    const u32 myCount = __popc(present);
    const u32 inclCount = wave_incl_scan(myCount);
    const u32 asz = (u32)__shfl((int)inclCount, 63, 64);
    const u32 rankBase = inclCount - myCount;

    // ---- alphabet (EntropyUtils.cpp:57-89)
    u32 pos = 0;
    if (asz == 256) {
        pos = 2;
    } else {
        const u64 anyMask = __ballot(present != 0);
        const int lastLane = 63 - __clzll((long long)anyMask);
        const u32 lastMask = (u32)lastLane >> 1;
        if (lane == 0) { or_bits_words(hdrw, 0, 1, 1); or_bits_words(hdrw, 1, lastMask, 5); }
        const u32 other = (u32)__shfl_xor((int)present, 1, 64);
        if ((lane & 1) == 0 && ((u32)lane >> 1) <= lastMask) or_bits_words(hdrw, 6 + 8 * ((u32)lane >> 1), present | (other << 4), 8);
        pos = 6 + 8 * (lastMask + 1);
    }

    // alphabet[] and keys in alphabet order
    {
        u32 r = rankBase;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            sizes[4 * lane + k] = 0;
            codes[4 * lane + k] = 0;
            if (f[k]) { alpha[r] = (u32)(4 * lane + k); keys[r] = (f[k] << 8) | (u32)(4 * lane + k); r++; }
        }
    }
    __syncthreads();

    if (asz > 1) {
        // ---- sort keys ascending (unique keys): rank by counting
        for (u32 i = lane; i < asz; i += 64) {
            const u32 k = keys[i];
            u32 r = 0;
            for (u32 j = 0; j < asz; j++) r += (keys[j] < k) ? 1u : 0u;
            sorted[r] = k;
        }
        __syncthreads();
    }

    if (lane == 0) {
        if (asz == 1) {
            codes[alpha[0]] = 1 << 12;
            sizes[alpha[0]] = 1;
        } else {
            // computeCodeLengths (HuffmanEncoder.cpp:219-244)
            for (u32 i = 0; i < asz; i++) { work[i] = sorted[i] >> 8; sorted[i] &= 0xFF; }
            u32 maxCodeLen = mk_code_lengths(work, (int)asz);
            for (u32 i = 0; i < asz; i++) sizes[sorted[i]] = (u16)work[i];
            if (maxCodeLen > HUF_MAX_LEN) {
                // limitCodeLengths (HuffmanEncoder.cpp:129-215); sorted[] plays 'ranks'
                u32* ranks = sorted;
                int nn = 0, debt = 0;
                while (sizes[ranks[nn]] >= HUF_MAX_LEN) {
                    debt += (sizes[ranks[nn]] - HUF_MAX_LEN);
                    sizes[ranks[nn]] = HUF_MAX_LEN;
                    nn++;
                }
                if (debt == 0) maxCodeLen = HUF_MAX_LEN;
                else {
                    // six queues of rank indices; stored in 'keys' as consecutive ranges because ranks
                    // are visited in order: queue idx holds a contiguous run [qStart[idx], qEnd[idx])
                    int qStart[6], qEnd[6], qHead[6];
                    for (int i = 0; i < 6; i++) { qStart[i] = 0; qEnd[i] = 0; qHead[i] = 0; }
                    // the reference appends n to v[idx] where idx is derived from the (non-decreasing in n)
                    // sizes; runs per idx need not be contiguous in general, so keep explicit lists in 'keys'
                    // laid out as 6 x 256 would not fit: use per-entry tags instead
                    // tag[n] = idx + 1 for queued entries (0 = not queued), order inside a queue = increasing n
                    u32* tag = keys;
                    for (u32 i = 0; i < asz; i++) tag[i] = 0;
                    while (nn < (int)asz) {
                        const int idx = HUF_MAX_LEN - 1 - (int)sizes[ranks[nn]];
                        if ((idx > 5) || (debt < (1 << idx))) break;
                        tag[nn] = (u32)idx + 1;
                        nn++;
                    }
                    (void)qStart; (void)qEnd;
                    // next unread element of queue idx: smallest n >= qHead[idx] with tag == idx+1
                    auto qNext = [&](int idx) -> int {
                        int p = qHead[idx];
                        while (p < (int)asz && tag[p] != (u32)idx + 1) p++;
                        return p;
                    };
                    int idx = 5;
                    while ((debt > 0) && (idx >= 0)) {
                        const int p = qNext(idx);
                        if ((p >= (int)asz) || (debt < (1 << idx))) { idx--; continue; }
                        sizes[ranks[p]]++;
                        debt -= (1 << idx);
                        qHead[idx] = p + 1;
                    }
                    idx = 0;
                    while ((debt > 0) && (idx < 6)) {
                        const int p = qNext(idx);
                        if (p >= (int)asz) { idx++; continue; }
                        sizes[ranks[p]]++;
                        debt -= (1 << idx);
                        qHead[idx] = p + 1;
                    }
                    if (debt > 0) {
                        // slow path: renormalise to 2048 and recompute (HuffmanEncoder.cpp:196-212)
                        u32* fl = work;          // f[i] = freqs[alphabet[i]]
                        u32* al = keys;          // scratch alphabet of the normaliser
                        u32 totalFreq = 0;
                        for (u32 i = 0; i < asz; i++) { fl[i] = hist[0][alpha[i]]; totalFreq += fl[i]; }
                        for (u32 i = asz; i < 256; i++) fl[i] = 0;
                        normalize_serial(fl, al, (int)asz, totalFreq, ENT_CHUNK >> 3);
                        for (u32 i = 0; i < asz; i++) ranks[i] = (fl[i] << 8) | alpha[i];
                        // sort ascending (insertion sort; rare path)
                        for (u32 i = 1; i < asz; i++) {
                            const u32 k = ranks[i];
                            int j = (int)i - 1;
                            while (j >= 0 && ranks[j] > k) { ranks[j + 1] = ranks[j]; j--; }
                            ranks[j + 1] = k;
                        }
                        for (u32 i = 0; i < asz; i++) { fl[i] = ranks[i] >> 8; ranks[i] &= 0xFF; }
                        maxCodeLen = mk_code_lengths(fl, (int)asz);
                        for (u32 i = 0; i < asz; i++) sizes[ranks[i]] = (u16)fl[i];
                    } else {
                        maxCodeLen = HUF_MAX_LEN;
                    }
                }
            }
            sh_maxCodeLen = maxCodeLen;
        }
    }
    __syncthreads();
    if (asz > 1) {
        if (sh_maxCodeLen > (u32)HUF_MAX_LEN) {
            for (u32 i = lane; i < asz; i += 64) { codes[alpha[i]] = (u16)i; sizes[alpha[i]] = 8; }
        } else {
            // generateCanonicalCodes (HuffmanCommon.cpp:29-63): symbols ordered by (length, symbol). Round 5: by the whole wave -- per
            // length a prefix count over the lanes' four symbols (the symbols of a length in symbol order), the first code of a length
            // from the one before; as one lane's loop over 12 lengths x the alphabet this was 40 % of the kernel.
            u32 sz4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) sz4[k] = sizes[4 * lane + k];
            u32 code = 0, curLen = 0;
            bool first = true;
            for (u32 l = 1; l <= (u32)HUF_MAX_LEN; l++) {
                u32 mineCount = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) mineCount += (sz4[k] == l) ? 1u : 0u;
                const u32 incl = wave_incl_scan(mineCount);
                const u32 total = (u32)__shfl((int)incl, 63, 64);
                if (total == 0) continue;
                if (first) { curLen = l; first = false; }
                code <<= (l - curLen);
                curLen = l;
                u32 r = code + incl - mineCount;
#pragma unroll
                for (int k = 0; k < 4; k++) if (sz4[k] == l) codes[4 * lane + k] = (u16)(r++);
                code += total;
            }
        }
    }
    __syncthreads();
    // length deltas (HuffmanEncoder.cpp:111-123), by the whole wave: lane l codes the alphabet entries 4 l .. 4 l + 3 (each against the
    // length of the entry before it), the bit positions are a prefix sum
    u32 hdrBits;
    {
        u32 cc[4], nn[4], nbSum = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const u32 r = 4 * (u32)lane + t;
            cc[t] = 0; nn[t] = 0;
            if (r < asz) {
                const u32 sy = alpha[r];
                const u32 szr = sizes[sy];
                const u32 prevSize = r ? (u32)sizes[alpha[r - 1]] : 2u;
                eg_signed((int)(int8_t)(u8)(szr - prevSize), cc[t], nn[t]);
                codes[sy] = (u16)(codes[sy] | (szr << 12));
                nbSum += nn[t];
            }
        }
        const u32 incl = wave_incl_scan(nbSum);
        u32 p = pos + incl - nbSum;
#pragma unroll
        for (int t = 0; t < 4; t++) { or_bits_words(hdrw, p, cc[t], nn[t]); p += nn[t]; }
        hdrBits = pos + (u32)__shfl((int)incl, 63, 64);
    }
    __syncthreads();
    u32* hdrOut = reinterpret_cast<u32*>(tmp + (size_t)slot * TMP_STRIDE);
    for (u32 i = lane; i < ((hdrBits + 31) >> 5); i += 64) hdrOut[i] = bswap32(hdrw[i]);

    if (asz <= 1) {
        if (lane == 0) { cd->hdrBits = hdrBits; cd->midLen = 0; cd->trailerLen = 0; cd->nPieces = 0; cd->aux = asz; }
        return;
    }

    // ---- encodeChunk (HuffmanEncoder.cpp:348-421): 4 fragments of n/4 symbols
    const u32 szFrag = n / 4;
    u8* pay = tmp + (size_t)slot * TMP_STRIDE + HDR_BYTES;
    u32 fragBits[4];
    for (int j = 0; j < 4; j++) {
        for (int i = lane; i < (int)HUF_FRAG_WORDS; i += 64) fragw[i] = 0;
        __syncthreads();
        const u8* src = blk + (size_t)j * szFrag;
        u32 bitBase = 0;
        for (u32 base = 0; base < szFrag; base += 1024) {
            // 16 symbols per lane
            const u32 s0 = base + 16 * (u32)lane;
            u64 acc[3] = { 0, 0, 0 };            // up to 192 bits, MSB-first: acc[0] holds the first 64 bits
            u32 nb = 0;
            for (u32 k = 0; k < 16; k++) {
                const u32 idx = s0 + k;
                if (idx >= szFrag) break;
                const u32 c = codes[src[idx]];
                const u32 l = c >> 12;
                const u64 v = c & 0xFFF;
                // append l bits of v at bit offset nb
                const u32 w = nb >> 6, o = nb & 63;
                if (o + l <= 64) acc[w] |= v << (64 - o - l);
                else { acc[w] |= v >> (o + l - 64); acc[w + 1] |= v << (128 - o - l); }
                nb += l;
            }
            const u32 incl = wave_incl_scan(nb);
            const u32 total = (u32)__shfl((int)incl, 63, 64);
            u32 p = bitBase + incl - nb;
            // flush lane bits in <= 32-bit pieces
            u32 done = 0;
            while (done < nb) {
                const u32 take = (nb - done) < 32 ? (nb - done) : 32;
                const u32 w = done >> 6, o = done & 63;
                u64 piece;
                if (o + take <= 64) piece = (acc[w] >> (64 - o - take));
                else piece = (acc[w] << (o + take - 64)) | (acc[w + 1] >> (128 - o - take));
                piece &= (take == 32) ? 0xFFFFFFFFull : ((1ull << take) - 1);
                or_bits_words(fragw, p, (u32)piece, take);
                p += take;
                done += take;
            }
            bitBase += total;
        }
        __syncthreads();
        fragBits[j] = bitBase;
        u32* dstw = reinterpret_cast<u32*>(pay + (size_t)j * HUF_FRAG_STRIDE);
        for (u32 i = lane; i < ((bitBase + 31) >> 5); i += 64) dstw[i] = bswap32(fragw[i]);
        __syncthreads();
    }

    if (lane == 0) {
        u8 mid[24];
        u32 ml = 0;
        for (int j = 0; j < 4; j++) {
            u32 v = fragBits[j];
            while (v >= 128) { mid[ml++] = (u8)(0x80 | (v & 0x7F)); v >>= 7; }
            mid[ml++] = (u8)v;
        }
        u32 mw[6] = { 0, 0, 0, 0, 0, 0 };
        for (u32 i = 0; i < ml; i++) mw[i >> 2] |= (u32)mid[i] << (8 * (i & 3));
        for (int i = 0; i < 6; i++) cd->mid[i] = mw[i];
        cd->midLen = ml;
        cd->hdrBits = hdrBits;
        cd->nPieces = 4;
        for (int j = 0; j < 4; j++) { cd->pieceBits[j] = fragBits[j]; cd->piecePtr[j] = pay + (size_t)j * HUF_FRAG_STRIDE; }
        const u32 rem = n - 4 * szFrag;
        u32 tw[2] = { 0, 0 };
        for (u32 i = 0; i < rem; i++) tw[0] |= (u32)blk[4 * szFrag + i] << (8 * i);
        cd->trailer[0] = tw[0]; cd->trailer[1] = tw[1];
        cd->trailerLen = rem;
        cd->aux = asz;
    }
}

// ------------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------------
struct HufDecChunk {
    u64 fragBit[4];      // first bit of each fragment
    u32 fragBits[4];     // declared bit counts
    u64 tailBit;         // raw bytes after the fragments (or the raw chunk itself for kind 2)
    u16 asz;
    u8 kind;             // 0 coded, 1 single-symbol fill, 2 raw, 3 unused
    u8 sym;
    u8 sizes[256];
};

#ifdef KNZ_EMU
struct hu32x4 { u32 x, y, z, w; };                    // (the CPU emulation of tests/emu is built with g++)
#else
typedef u32 hu32x4 __attribute__((ext_vector_type(4)));
#endif

// n in [1, 32]; rel = bit offset from the start of the LDS window (words in bit order)
__device__ __forceinline__ u32 hwin_bits(const u32* win, u32 rel, u32 n)
{
    const u32 i = rel >> 5;
    const u64 v = ((u64)win[i] << 32) | (u64)win[i + 1];
    return (u32)((v << (rel & 31)) >> (64 - n));
}

constexpr u32 HSCAN_WIN_BITS = 8192;
constexpr u32 HSCAN_NEED_BITS = 6 + 256 + 256 * 8 + 4 * 40 + 64;     // alphabet + 256 deltas (|d| <= 11: 8 bits) + 4 var-ints

// One wave per block walks the chunk headers (readLengths, HuffmanDecoder.cpp:65-108, and the 4 fragment sizes,
// :204-246).  Same scheme as the rANS scan: a 1 KiB window of the stream in LDS, the next window prefetched at a
// guessed position, presence masks counted in parallel; the Exp-Golomb deltas are a short uniform chain
// (one LDS read + count-leading-zeros per code).
// V5: the chunk layout of bitstream versions below 6 (HuffmanDecoder.cpp:349-459): no raw small chunks; behind the lengths a 2-bit
// stream count (0 = one) and ONE var-int bit count, then one code stream for the whole chunk -- kept as "fragment 0" here.
template <bool V5>
__global__ __launch_bounds__(64) void k_huff_scan(BitSrc src, DecBlock* __restrict__ blocks, int nBlocks, int maxChunks,
                                                  HufDecChunk* __restrict__ chunks)
{
    const int b = blockIdx.x;
    const int lane = lane_id();
    DecBlock& db = blocks[b];
    HufDecChunk* cs = chunks + (size_t)b * maxChunks;
    for (int i = lane; i < maxChunks; i += 64) cs[i].kind = 3;
    if (db.error) return;
    __shared__ u32 win[256 + 8];
    __shared__ u32 codeSizeW[64];                      // 256 code lengths, written four at a time
    u8* codeSize = reinterpret_cast<u8*>(codeSizeW);
    const u64 limit = db.payloadBit + ((db.bits + 7) & ~7ull);
    u64 pos = db.entropyBit;
    const u64 entropyBit = pos;
    const u32 preLen = db.preLen;
    if (db.copyBlock) {
        if (lane == 0) {
            cs[0].kind = 2; cs[0].tailBit = pos;
            pos += 8ull * preLen;
            if (pos > limit) db.error = KNZ_ERR_PROCESS_BLOCK;
            db.usedBits = pos - entropyBit;
        }
        return;
    }
    const u64 lastWord = ((src.nBytes + 3) >> 2) - 1;
    auto issue = [&](u64 bit0, hu32x4& dstv) {
        const u64 w = (bit0 >> 5) + 4ull * (u32)lane;
        const u64 w1 = w + 1, w2 = w + 2, w3 = w + 3;
        const u32* p = src.words;
        dstv.x = p[w < lastWord ? w : lastWord];
        dstv.y = p[w1 < lastWord ? w1 : lastWord];
        dstv.z = p[w2 < lastWord ? w2 : lastWord];
        dstv.w = p[w3 < lastWord ? w3 : lastWord];
    };
    hu32x4 wr = { 0, 0, 0, 0 }, guess = { 0, 0, 0, 0 };
    u64 winBit0 = 0, guessBit0 = 0;
    bool winValid = false, guessValid = false;
    auto ensure = [&](u32 need) {
        if (winValid && winBit0 <= pos && pos + need <= winBit0 + HSCAN_WIN_BITS) return;
        if (guessValid && guessBit0 <= pos && pos + need <= guessBit0 + HSCAN_WIN_BITS) { wr = guess; winBit0 = guessBit0; }
        else { winBit0 = pos & ~31ull; issue(winBit0, wr); }
        guessValid = false;
        __syncthreads();
        {
            hu32x4 sw = { bswap32(wr.x), bswap32(wr.y), bswap32(wr.z), bswap32(wr.w) };
            *reinterpret_cast<hu32x4*>(&win[4 * lane]) = sw;
        }
        __syncthreads();
        winValid = true;
    };
    // Signed Exp-Golomb code of a length delta (ExpGolombDecoder.hpp:52-75) by its first 8 bits, for codes that start with a 0
    // (a leading 1 is the one-bit code of delta 0): bits 6-9 = code length (0: no valid code starts like this), bits 0-4 =
    // delta + 16. A valid delta has |d| <= 11, i.e. a code of at most 8 bits (a prefix of more than 3 zeros can only decode to an
    // out-of-range length). 128 entries of 16 bits: lane l holds entries 2l and 2l + 1.
    u32 egTab = 0;
    for (u32 j = 0; j < 2; j++) {
        const u32 v = 2u * (u32)lane + j;                                    // 0xxxxxxx
        const u32 lg = 1u + (u32)__clz((int)(v << 25));                      // zeros behind the leading 0, plus one
        u32 e = 0;
        if ((v << 25) != 0 && lg <= 3) {
            const u32 total = 2 * lg + 2;
            int res = (int)((v >> (8 - total)) & ((1u << (lg + 1)) - 1u));
            const int sgn = res & 1;
            res = (res >> 1) + (1 << lg) - 1;
            const int delta = (int)(int8_t)(u8)((res - sgn) ^ -sgn);
            e = (total << 6) | (u32)(delta + 16);
        }
        egTab |= e << (16 * j);
    }
    const u32 nChunks = (preLen + ENT_CHUNK - 1) / ENT_CHUNK;
    u64 prevPos = pos;
    int err = 0;
    for (u32 ci = 0; ci < nChunks && !err; ci++) {
        HufDecChunk& c = cs[ci];
        const u32 n = (preLen - ci * ENT_CHUNK < ENT_CHUNK) ? (preLen - ci * ENT_CHUNK) : ENT_CHUNK;
        if (!V5 && n < 32) {
            if (lane == 0) { c.kind = 2; c.tailBit = pos; }
            pos += 8ull * n;
            if (pos > limit) err = 1;
            continue;
        }
        ensure(HSCAN_NEED_BITS);
        guessValid = false;
        if (ci >= 1 && ci + 1 < nChunks) {
            const u64 est = pos + (pos - prevPos);
            guessBit0 = (est > 2304 ? est - 2304 : 0) & ~31ull;
            issue(guessBit0, guess);
            guessValid = true;
        }
        prevPos = pos;
        u32 p = (u32)(pos - winBit0);
        // ---- alphabet (EntropyUtils.cpp:91-123): which of my 4 symbols are present
        u32 asz, present, firstSym = 0;
        const u32 hb = hwin_bits(win, p, 7);
        if ((hb >> 6) == 0) { asz = ((hb >> 5) & 1) ? 0u : 256u; present = asz ? 0xFu : 0u; p += 2; }
        else {
            const u32 lastMask = (hb >> 1) & 31;
            p += 6;
            const u32 m = (u32)lane >> 1;
            const u32 byte = (m <= lastMask) ? hwin_bits(win, p + 8u * m, 8) : 0u;
            present = (lane & 1) ? (byte >> 4) : (byte & 0xF);
            p += 8u * (lastMask + 1);
            asz = wave_sum(__popc(present));
            const u64 nz = __ballot(present != 0);
            if (nz) {
                const int fl = __ffsll((long long)nz) - 1;
                const u32 fp = (u32)__builtin_amdgcn_readlane((int)present, fl);
                firstSym = 4u * (u32)fl + (u32)(__ffs((int)fp) - 1);
            }
        }
        asz = (u32)__builtin_amdgcn_readfirstlane((int)asz);
        if (asz == 0) { err = 2; break; }
        // ---- code length deltas (ExpGolombDecoder.hpp:52-75)
        // A valid delta has |d| <= 11, i.e. a code of at most 8 bits (prefix of <= 3 zeros); a longer prefix can only decode to an
        // out-of-range length, so it is rejected right away. The deltas were 89 % of this kernel (measured with the cycle counter: 108
        // cycles per code, a chain of ~ 14 scalar instructions and a readlane). Round 5: 56 bits of header at a time. Lane b < 56
        // decodes the code that WOULD start at bit q + b (its 8 bits from the LDS window, one lookup in the table the wave holds in a
        // register) and knows where the next one starts, next[b] = b + length <= 63; lanes 56-63 point at themselves. What is left of
        // the chain is "which offsets start a code": pos = readlane(next, pos) and a bit set in a scalar mask per code, eight at a time
        // without a branch -- once it has left the 56 bits it sits on the lane whose number IS the offset of the next window's first
        // code. Ranks (popcount of the mask below the lane), the running length (a wave scan over the deltas of the lanes that start a
        // code), the range check and the stores are parallel.
        u32 curSize = 2;
        u32 q = (u32)__builtin_amdgcn_readfirstlane((int)p);
        u32 bad = 0;
        const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        u32 k = 0;
        while (k < asz) {
            if (q > HSCAN_WIN_BITS - 96) { bad = 1; break; }                 // longer than any valid header
            const u32 bp = q + (u32)lane;
            const u32 wi = bp >> 5;                                          // (<= 254: the window has 264 words)
            const u32 top = (u32)(((((u64)win[wi] << 32) | (u64)win[wi + 1]) << (bp & 31)) >> 56);
            const u32 w = (u32)__shfl((int)egTab, (int)((top >> 1) & 63u), 64);
            const u32 e = (top & 0x80u) ? ((1u << 6) | 16u) : ((w >> (16 * (top & 1))) & 0xFFFFu);
            const u32 len = e >> 6;                                          // 0 for an invalid prefix: it ends the window, and is flagged if a code starts there
            const u32 nxt = (lane >= 56) ? (u32)lane : (len ? (u32)lane + len : 63u);
            unsigned long long starts = 0;
            u32 pos = 0;
            do {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    starts |= 1ull << pos;
                    pos = (u32)__builtin_amdgcn_readlane((int)nxt, (int)pos);
                }
            } while (pos < 56);
            starts &= (1ull << 56) - 1ull;
            const u32 want = asz - k;
            const u32 rank = (u32)__popcll(starts & below);
            u32 cnt = (u32)__popcll(starts);
            u32 adv = pos;                                                   // bits of this window that belong to the codes taken
            if (cnt > want) {                                                // the last window: the code with rank `want` is not a length any more
                const unsigned long long nextStart = __ballot(((starts >> lane) & 1ull) && rank == want);
                adv = (u32)__ffsll((long long)nextStart) - 1u;
                cnt = want;
            }
            const bool mine = ((starts >> lane) & 1ull) && rank < want;
            const u32 dsum = wave_incl_scan(mine ? (e & 31u) - 16u : 0u);   // (two's complement: deltas are signed)
            const u32 cs = curSize + dsum;                                   // the length after this code
            if (mine) {
                // (the reference's int8 arithmetic wraps only behind a length that is already out of range)
                if (len == 0 || cs - 1u > (u32)HUF_MAX_LEN - 1u) bad = 1;
                codeSize[k + rank] = (u8)cs;
            }
            if (__ballot(bad != 0) != 0) { bad = 1; break; }
            curSize = (u32)__builtin_amdgcn_readlane((int)cs, 63);
            k += cnt;
            q += adv;
        }
        if (bad) err = 1;
        if (err) break;
        __syncthreads();
        {
            u32 r = wave_incl_scan(__popc(present)) - __popc(present);
            u32 packed = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) if ((present >> k) & 1) { packed |= (u32)codeSize[r] << (8 * k); r++; }
            reinterpret_cast<u32*>(c.sizes)[lane] = packed;
        }
        __syncthreads();
        if (asz == 1) {
            if (lane == 0) { c.asz = 1; c.kind = 1; c.sym = (u8)firstSym; }
            pos = winBit0 + q;
            continue;
        }
        if (V5) {
            // ---- HuffmanDecoder.cpp:374-383: stream count, bit count
            if (hwin_bits(win, q, 2) != 0) { err = 1; break; }
            q += 2;
            u32 value = hwin_bits(win, q, 8); q += 8;
            u32 res = value & 0x7F;
            for (int shift = 7; value >= 128; shift += 7) {
                value = hwin_bits(win, q, 8); q += 8;
                if (shift == 28) { if (value >= 128 || (value & 0x70) != 0) err = 1; res |= (value & 0x0F) << shift; break; }
                res |= (value & 0x7F) << shift;
            }
            if ((int)res < 0 || res > n * (u32)HUF_MAX_LEN) err = 1;
            if (err) break;
            const u64 fp5 = winBit0 + q;
            if (lane == 0 && res != 0) {                         // (:386: a chunk that declares no bits is left as it is)
                c.asz = (u16)asz;
                c.fragBit[0] = fp5; c.fragBits[0] = res;
                for (int j = 1; j < 4; j++) { c.fragBit[j] = fp5 + res; c.fragBits[j] = 0; }
                c.tailBit = fp5 + res;
                c.kind = 0;
            }
            pos = fp5 + res;
            if (pos > limit) err = 1;
            continue;
        }
        // ---- decodeChunk prologue (HuffmanDecoder.cpp:204-246): 4 fragment sizes in bits
        const u32 maxFragBits = 8192u << 3;
        u32 szb[4];
        for (int j = 0; j < 4; j++) {
            u32 value = hwin_bits(win, q, 8); q += 8;
            u32 res = value & 0x7F;
            for (int shift = 7; value >= 128; shift += 7) {
                value = hwin_bits(win, q, 8); q += 8;
                if (shift == 28) { if (value >= 128 || (value & 0x70) != 0) err = 1; res |= (value & 0x0F) << shift; break; }
                res |= (value & 0x7F) << shift;
            }
            szb[j] = res;
            if ((int)res < 0 || res > maxFragBits) err = 1;
        }
        if (err) break;
        u64 fp = winBit0 + q;
        if (lane == 0) {
            c.asz = (u16)asz;
            for (int j = 0; j < 4; j++) { c.fragBit[j] = fp; c.fragBits[j] = szb[j]; fp += szb[j]; }
            c.tailBit = fp;
            c.kind = 0;
        }
        pos = winBit0 + q + (u64)szb[0] + szb[1] + szb[2] + szb[3] + 8ull * (n - 4 * (n / 4));
        if (pos > limit) err = 1;
    }
    if (lane == 0) {
        if (err) db.error = KNZ_ERR_PROCESS_BLOCK;
        db.usedBits = pos - entropyBit;
    }
}

// every load issued so far has arrived (before a burst of stores: one counter for loads and stores, counted in issue order)
// (KNZ_LOADS_DONE: common.hpp)
constexpr int HUF_DEC_CHUNKS = 16;  // chunks per wave (4 lanes = 4 fragments each)
constexpr u32 HUF_PRIM_BITS = 10;   // codes of up to 10 bits are looked up by the top 10 bits of the window
constexpr u32 HUF_OVF = 512;        // table entries of the longer codes: at most 256 codes of 11 or 12 bits, two or one 12-bit slots each

// 16 chunks per wave.  Phase 1: the whole wave builds each chunk's table (canonical order by counting: per code length a wave prefix sum
// of how many of my 4 symbols have it).  The table of the reference has 4096 entries (12 bits, 8 KiB per chunk: 16 chunks per CU, 3.2
// rounds of workgroups over the 12.9 k chunks of a 212 MB job); here it is two: 1024 entries for the top 10 bits (every code of up to 10
// bits covers whole groups of four 12-bit slots, and so does the unassigned tail behind the last code) and up to 512 for the 12-bit slots
// of the codes of 11 and 12 bits, which are consecutive (canonical codes grow with their length) -- 3 KiB per chunk, twice the chunks
// per CU, every lane of the wave a fragment. A step reads both (the second at a clamped index) and picks by one unsigned compare.  Phase 2: one lane per fragment keeps
// 32..64 bits of its stream in a register pair; the next 32 bits are loaded every 2 steps without a branch and
// only committed when there is room (the load is simply repeated otherwise), so no global access sits on the
// table-lookup chain.
// V5 (chunks of bitstream versions below 6): the chunk is one code stream, so one lane of the four decodes all of it.
template <bool V5>
__global__ __launch_bounds__(64) void k_huff_decode(BitSrc src, DecBlock* __restrict__ blocks, int maxChunks, int nSlots,
                                                    const HufDecChunk* __restrict__ chunks, u8* const* __restrict__ outPtr)
{
    __shared__ u16 prim[HUF_DEC_CHUNKS][1u << HUF_PRIM_BITS];   // (sym << 8) | len of 12-bit slot 4 q + 3
    __shared__ u16 ovf[HUF_DEC_CHUNKS][HUF_OVF];                // ... of 12-bit slot longBase + i
    __shared__ u16 longBase[HUF_DEC_CHUNKS], longCnt[HUF_DEC_CHUNKS];
    __shared__ u16 rStart[260];                       // table index where canonical rank r starts
    __shared__ u16 rVal[260];
    __shared__ int chunkErr[HUF_DEC_CHUNKS];
    __shared__ u32 ringAll[64 * 32];                  // the fragments' streams: 32 words per lane
    const int lane = lane_id();
    const int slotBase = blockIdx.x * HUF_DEC_CHUNKS;
    if (lane < HUF_DEC_CHUNKS) chunkErr[lane] = 0;
    __syncthreads();

    for (int gg = 0; gg < HUF_DEC_CHUNKS; gg++) {
        const int slot = slotBase + gg;
        if (slot >= nSlots) break;
        const HufDecChunk& c = chunks[slot];
        const int b = slot / maxChunks;
        const int ci = slot - b * maxChunks;
        if (c.kind == 3 || blocks[b].error) continue;
        u8* dst = outPtr[b] + (size_t)ci * ENT_CHUNK;
        const u32 preLen = blocks[b].preLen;
        const u32 n = (preLen - (u32)ci * ENT_CHUNK < ENT_CHUNK) ? (preLen - (u32)ci * ENT_CHUNK) : ENT_CHUNK;
        if (c.kind == 2) {
            for (u32 i = lane; i < n; i += 64) dst[i] = (u8)peek_bits(src, c.tailBit + 8ull * i, 8);
            continue;
        }
        if (c.kind == 1) {
            for (u32 i = lane; i < n; i += 64) dst[i] = c.sym;
            continue;
        }
        // ---- canonical codes + table (HuffmanCommon.cpp:29-63, HuffmanDecoder.cpp:111-140)
        const u32 packed = reinterpret_cast<const u32*>(c.sizes)[lane];
        u32 sz[4];
#pragma unroll
        for (int k = 0; k < 4; k++) sz[k] = (packed >> (8 * k)) & 0xFF;
        u32 rank[4] = { 0, 0, 0, 0 }, start[4] = { 0, 0, 0, 0 };
        u32 rankBase = 0, startBase = 0, lb = 0;
        for (u32 l = 1; l <= (u32)HUF_MAX_LEN; l++) {
            if (l == HUF_PRIM_BITS + 1) lb = startBase;                  // the first 12-bit slot of a code longer than 10 bits
            u32 mine = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) mine += (sz[k] == l) ? 1u : 0u;
            const u32 incl = wave_incl_scan(mine);
            const u32 totalL = (u32)__shfl((int)incl, 63, 64);
            u32 idx = incl - mine;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (sz[k] == l) { rank[k] = rankBase + idx; start[k] = startBase + (idx << (HUF_MAX_LEN - l)); idx++; }
            }
            rankBase += totalL;
            startBase += totalL << (HUF_MAX_LEN - l);
        }
        const u32 asz = rankBase;
        if (startBase > 4096 || asz < 2) { if (lane == 0) chunkErr[gg] = 1; continue; }     // :130-133 (end > table size)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (sz[k]) { rStart[rank[k]] = (u16)start[k]; rVal[rank[k]] = (u16)(((4u * (u32)lane + (u32)k) << 8) | sz[k]); }
        }
        if (lane == 0) { rStart[asz] = (u16)startBase; rVal[asz] = 0x0707; rStart[asz + 1] = 4097; longBase[gg] = (u16)lb; longCnt[gg] = (u16)(startBase - lb); }
        __syncthreads();
        {
            // lane walks the 12-bit slots [64*lane, 64*lane+64): find the rank covering the first slot, then walk
            const u32 nLongs = startBase - lb;                            // (<= 512: see HUF_OVF)
            const u32 base = (u32)lane * 64;
            u32 lo = 0, hi = asz;                     // rank asz = the unassigned tail (0x0707)
            while (lo < hi) {
                const u32 mid = (lo + hi + 1) >> 1;
                if (rStart[mid] <= base) lo = mid; else hi = mid - 1;
            }
            u32 r = lo;
            for (u32 k = 0; k < 64; k++) {
                const u32 t = base + k;
                while (rStart[r + 1] <= t) r++;
                const u32 v = rVal[r];
                if ((k & 3) == 3) prim[gg][t >> 2] = (u16)v;
                if (t - lb < nLongs) ovf[gg][t - lb] = (u16)v;
            }
        }
        __syncthreads();
    }
    __syncthreads();

    // ---- one lane per fragment
    const int g = lane >> 2;
    const int j = lane & 3;
    bool act = false;
    u32 szFrag = 0, fragBits = 0;
    u64 fbeg = 0;
    u8* dst = nullptr;
    u32 n = 0;
    int b = 0, ci = 0;
    if (g < HUF_DEC_CHUNKS) {
        const int slot = slotBase + g;
        if (slot < nSlots) {
            const HufDecChunk& c = chunks[slot];
            b = slot / maxChunks;
            ci = slot - b * maxChunks;
            if (c.kind == 0 && !blocks[b].error && !chunkErr[g] && (!V5 || j == 0)) {
                act = true;
                const u32 preLen = blocks[b].preLen;
                n = (preLen - (u32)ci * ENT_CHUNK < ENT_CHUNK) ? (preLen - (u32)ci * ENT_CHUNK) : ENT_CHUNK;
                szFrag = V5 ? n : n / 4;
                dst = outPtr[b] + (size_t)ci * ENT_CHUNK + (size_t)j * szFrag;
                fbeg = c.fragBit[j];
                fragBits = c.fragBits[j];
            }
        }
    }
    const u64 lastWord = ((src.nBytes + 3) >> 2) - 1;
    const u64 w0 = fbeg >> 5;
    const u32 sh = (u32)fbeg & 31;
    // Round 5. The stream of a fragment reaches its lane through a ring of 32 words in LDS, as in k_ans0_decode: every 16 steps the lane
    // asks for the 8 words behind what the ring holds (plain loads, issued unconditionally) and puts the 8 it asked for 16 steps earlier
    // into the ring if there is room (else they are simply asked for again). The decode loop itself touches LDS only: the version before
    // loaded the next word two steps before it was needed and stored every 4 symbols from inside the loop, and since loads and stores
    // share one counter that counts in issue order (and a store through a pointer out of an array of pointers is a flat store, which
    // counts as an LDS operation as well), every two steps waited for a load and every four for a store: 2.8 ms per 212 MB for a chain
    // of four table lookups. Output: 64 steps in registers, written after an explicit wait for the loads, before the next are issued.
    constexpr u32 HR = 32, HQ = 8;                     // ring words per lane, words per refill
    u32* ring = ringAll + (u32)lane * HR;
    // words [f, f + 8) of the fragment (32 stream bits each, bits past the fragment's end zero: the reference decodes from a zero-padded copy)
    auto load9 = [&](u32 f, u32 raw[9]) {
#pragma unroll
        for (int t = 0; t < 9; t++) { const u64 w = w0 + f + t; raw[t] = src.words[w < lastWord ? w : lastWord]; }
    };
    auto store8 = [&](u32 f, const u32 raw[9]) {
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const u32 a = bswap32(raw[t]), c = bswap32(raw[t + 1]);
            const u32 v = sh ? ((a << sh) | (c >> (32 - sh))) : a;
            const u32 done = 32u * (f + t);
            const u32 valid = (fragBits > done) ? ((fragBits - done < 32u) ? fragBits - done : 32u) : 0u;
            ring[(f + t) & (HR - 1)] = valid == 32u ? v : (valid ? (v & ~((1u << (32 - valid)) - 1u)) : 0u);
        }
    };
    const u16* tabP = prim[g];
    const u16* tabO = ovf[g];
    const u32 lb12 = longBase[g], nL12 = act ? (u32)longCnt[g] : 0u;
    u32 pend[9];
    load9(0, pend); store8(0, pend);
    load9(HQ, pend); store8(HQ, pend);
    u32 F = 2 * HQ;                                    // fragment words the ring has seen
    u64 bitbuf = ((u64)ring[0] << 32) | ring[1];
    u32 k = 2;                                         // next fragment word to take
    u32 nv = ring[2];
    u32 cnt = 64;
    u32 used = 0;
    bool bad = false;
    const u32 steps = act ? szFrag : 0;
    const u32 maxSteps = wave_max(steps);
    const bool al4 = act && ((reinterpret_cast<uintptr_t>(dst) & 3) == 0);
    load9(F, pend);
    for (u32 i = 0; i < maxSteps; i += 64) {
        u32 out[16];
#pragma unroll
        for (int it = 0; it < 16; it++) {
            if ((it & 3) == 0 && (i | (u32)it)) {
                // (16 steps take at most 6 words; the ring never holds fewer than 10)
                if (F - k <= HR - HQ) { store8(F, pend); F += HQ; }
                load9(F, pend);
            }
            u32 acc = 0;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const bool room = cnt <= 32;
                bitbuf |= room ? ((u64)nv << (32 - cnt)) : 0ull;
                cnt += room ? 32u : 0u;
                k += room ? 1u : 0u;
                nv = ring[k & (HR - 1)];                // (for the next half: off the lookup chain)
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const u32 t12 = (u32)(bitbuf >> 52);
                    const u32 oi = t12 - lb12;
                    const bool isLong = oi < nL12;
                    const u32 pv = tabP[t12 >> 2], ov = tabO[isLong ? oi : 0u];
                    const u32 val = isLong ? ov : pv;
                    const u32 len = val & 0xFF;
                    const bool on = (i + 4 * (u32)it + 2 * h + t) < steps;
                    acc |= (val >> 8) << (8 * (2 * h + t));
                    bitbuf = on ? (bitbuf << len) : bitbuf;
                    cnt -= on ? len : 0u;
                    used += on ? len : 0u;
                }
            }
            out[it] = acc;
        }
        if (used > (V5 ? (u32)(ENT_CHUNK * HUF_MAX_LEN) : (8192u << 3))) bad = true;
        KNZ_LOADS_DONE();
#pragma unroll
        for (int it = 0; it < 16; it++) {
            const u32 o = i + 4 * (u32)it;
            if (o + 4 <= steps) {
                if (al4) stg<u32>(dst + o, out[it]);
                else { stg<u8>(dst + o, (u8)out[it]); stg<u8>(dst + o + 1, (u8)(out[it] >> 8)); stg<u8>(dst + o + 2, (u8)(out[it] >> 16)); stg<u8>(dst + o + 3, (u8)(out[it] >> 24)); }
            } else {
                for (u32 t = 0; o + t < steps; t++) stg<u8>(dst + o + t, (u8)(out[it] >> (8 * t)));
            }
        }
    }
    if (act) {
        if (used != fragBits) bad = true;
        if (!V5 && j == 0) {
            const HufDecChunk& c = chunks[slotBase + g];
            for (u32 i = 4 * szFrag; i < n; i++)
                outPtr[b][(size_t)ci * ENT_CHUNK + i] = (u8)peek_bits(src, c.tailBit + 8ull * (i - 4 * szFrag), 8);
        }
        if (bad) chunkErr[g] = 1;
    }
    __syncthreads();
    if (lane < HUF_DEC_CHUNKS && chunkErr[lane] && slotBase + lane < nSlots)
        blocks[(slotBase + lane) / maxChunks].error = KNZ_ERR_PROCESS_BLOCK;
}

void launch_huffman_encode(hipStream_t s, BlockView view, int nBlocks, int maxChunks, ChunkDesc* desc, u8* tmp)
{
    const int nSlots = nBlocks * maxChunks;
    { KScope ks_("k_huff_encode"); hipLaunchKernelGGL(k_huff_encode, dim3(nSlots), dim3(64), 0, s, view, maxChunks, desc, tmp); }
}

void launch_huffman_decode(hipStream_t s, BitSrc src, DecBlock* blocks, int nBlocks, int maxChunks, void* chunkMeta, u8* const* outPtr, int bsVersion)
{
    HufDecChunk* chunks = reinterpret_cast<HufDecChunk*>(chunkMeta);
    const int nSlots = nBlocks * maxChunks;
    const dim3 gridD((nSlots + HUF_DEC_CHUNKS - 1) / HUF_DEC_CHUNKS);
    if (bsVersion < 6) {
        { KScope ks_("k_huff_scan"); hipLaunchKernelGGL(k_huff_scan<true>, dim3(nBlocks), dim3(64), 0, s, src, blocks, nBlocks, maxChunks, chunks); }
        { KScope ks_("k_huff_decode"); hipLaunchKernelGGL(k_huff_decode<true>, gridD, dim3(64), 0, s, src, blocks, maxChunks, nSlots, chunks, outPtr); }
        return;
    }
    { KScope ks_("k_huff_scan"); hipLaunchKernelGGL(k_huff_scan<false>, dim3(nBlocks), dim3(64), 0, s, src, blocks, nBlocks, maxChunks, chunks); }
    { KScope ks_("k_huff_decode"); hipLaunchKernelGGL(k_huff_decode<false>, gridD, dim3(64), 0, s, src, blocks, maxChunks, nSlots, chunks, outPtr); }
}

size_t huffman_dec_chunk_bytes() { return sizeof(HufDecChunk); }

}  // namespace knz
