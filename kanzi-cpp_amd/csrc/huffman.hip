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

    __shared__ u32 hist[4][256];
    __shared__ u32 hdrw[HDR_WORDS];
    __shared__ u32 keys[256];        // (freq << 8) | sym of present symbols, then sorted
    __shared__ u32 sorted[256];
    __shared__ u32 work[256];        // Moffat-Katajainen working array
    __shared__ u32 alpha[256];
    __shared__ u16 sizes[256];
    __shared__ u16 codes[256];       // (len << 12) | code
    __shared__ u32 fragw[HUF_FRAG_WORDS];
    __shared__ u32 sh_hdrBits;

    for (int i = lane; i < 1024; i += 64) (&hist[0][0])[i] = 0;
    for (int i = lane; i < (int)HDR_WORDS; i += 64) hdrw[i] = 0;
    __syncthreads();

    // ---- histogram
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

    u32 f[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int s = 4 * lane + k;
        f[k] = hist[0][s] + hist[1][s] + hist[2][s] + hist[3][s];
        hist[0][s] = f[k];                         // keep totals for the fallback path
    }
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
            if (maxCodeLen > HUF_MAX_LEN) {
                for (u32 i = 0; i < asz; i++) { codes[alpha[i]] = (u16)i; sizes[alpha[i]] = 8; }
            } else {
                // generateCanonicalCodes (HuffmanCommon.cpp:29-63): symbols ordered by (length, symbol)
                int code = 0, curLen = 0, first = 1;
                for (int l = 1; l <= HUF_MAX_LEN; l++) {
                    for (u32 i = 0; i < asz; i++) {
                        const u32 s = alpha[i];
                        if (sizes[s] != l) continue;
                        if (first) { curLen = l; first = 0; }
                        code <<= (l - curLen);
                        curLen = l;
                        codes[s] = (u16)code;
                        code++;
                    }
                }
            }
        }
        // length deltas (HuffmanEncoder.cpp:111-123)
        u32 p = pos;
        int prevSize = 2;
        for (u32 i = 0; i < asz; i++) {
            const u32 s = alpha[i];
            codes[s] |= (u16)(sizes[s] << 12);
            u32 c, nb;
            eg_signed((int)(int8_t)(u8)(sizes[s] - prevSize), c, nb);
            or_bits_words(hdrw, p, c, nb);
            p += nb;
            prevSize = sizes[s];
        }
        sh_hdrBits = p;
    }
    __syncthreads();
    const u32 hdrBits = sh_hdrBits;
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

__device__ __forceinline__ int eg_decode_signed(const BitSrc& s, u64& pos, int& err)
{
    if (take_bits(s, pos, 1, err) == 1) return 0;
    u32 lg = 1;
    while (take_bits(s, pos, 1, err) == 0) { lg++; if (err) return 0; }
    lg &= 7;
    int res = (int)take_bits(s, pos, lg + 1, err);
    const int sgn = res & 1;
    res = (res >> 1) + (1 << lg) - 1;
    return (int)(int8_t)(u8)((res - sgn) ^ -sgn);
}

__global__ __launch_bounds__(64) void k_huff_scan(BitSrc src, DecBlock* __restrict__ blocks, int nBlocks, int maxChunks,
                                                  HufDecChunk* __restrict__ chunks)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= nBlocks) return;
    DecBlock& db = blocks[b];
    HufDecChunk* cs = chunks + (size_t)b * maxChunks;
    for (int i = 0; i < maxChunks; i++) cs[i].kind = 3;
    if (db.error) return;
    BitSrc s = src;
    s.limitBits = db.payloadBit + ((db.bits + 7) & ~7ull);
    u64 pos = db.entropyBit;
    const u32 preLen = db.preLen;
    int err = 0;
    if (db.copyBlock) {
        cs[0].kind = 2; cs[0].tailBit = pos;
        pos += 8ull * preLen;
        if (pos > s.limitBits) db.error = KNZ_ERR_PROCESS_BLOCK;
        db.usedBits = pos - db.entropyBit;
        return;
    }
    const u32 nChunks = (preLen + ENT_CHUNK - 1) / ENT_CHUNK;
    for (u32 ci = 0; ci < nChunks && !err; ci++) {
        HufDecChunk& c = cs[ci];
        const u32 n = (preLen - ci * ENT_CHUNK < ENT_CHUNK) ? (preLen - ci * ENT_CHUNK) : ENT_CHUNK;
        if (n < 32) {
            c.kind = 2; c.tailBit = pos;
            pos += 8ull * n;
            if (pos > s.limitBits) err = 1;
            continue;
        }
        for (int i = 0; i < 256; i++) c.sizes[i] = 0;
        // readLengths (HuffmanDecoder.cpp:65-108)
        u32 asz = 0;
        u32 firstSym = 0;
        int curSize = 2;
        if (take_bits(s, pos, 1, err) == 0) {
            const u32 full = (take_bits(s, pos, 1, err) == 0) ? 256u : 0u;
            for (u32 sy = 0; sy < full && !err; sy++) {
                curSize = (int)(int8_t)(curSize + eg_decode_signed(s, pos, err));
                if (curSize <= 0 || curSize > HUF_MAX_LEN) { err = 1; break; }
                c.sizes[sy] = (u8)curSize;
            }
            asz = full;
        } else {
            const u32 lastMask = take_bits(s, pos, 5, err);
            u64 mpos = pos;
            pos += 8ull * (lastMask + 1);
            if (pos > s.limitBits) err = 1;
            bool found = false;
            for (u32 m = 0; m <= lastMask && !err; m++) {
                u32 byte = take_bits(s, mpos, 8, err);
                while (byte && !err) {
                    const u32 bit = (u32)(__ffs((int)byte) - 1);
                    byte &= byte - 1;
                    const u32 sy = 8 * m + bit;
                    if (!found) { firstSym = sy; found = true; }
                    curSize = (int)(int8_t)(curSize + eg_decode_signed(s, pos, err));
                    if (curSize <= 0 || curSize > HUF_MAX_LEN) { err = 1; break; }
                    c.sizes[sy] = (u8)curSize;
                    asz++;
                }
            }
        }
        if (err) break;
        if (asz == 0) { err = 2; break; }
        c.asz = (u16)asz;
        if (asz == 1) { c.kind = 1; c.sym = (u8)firstSym; continue; }
        // decodeChunk prologue (HuffmanDecoder.cpp:204-246)
        const u32 maxFragBits = 8192u << 3;
        u32 szb[4];
        for (int j = 0; j < 4; j++) {
            szb[j] = take_varint(s, pos, err);
            if ((int)szb[j] < 0 || szb[j] > maxFragBits) err = 1;
        }
        if (err) break;
        for (int j = 0; j < 4; j++) { c.fragBit[j] = pos; c.fragBits[j] = szb[j]; pos += szb[j]; }
        c.tailBit = pos;
        pos += 8ull * (n - 4 * (n / 4));
        if (pos > s.limitBits) err = 1;
        c.kind = 0;
    }
    if (err) db.error = KNZ_ERR_PROCESS_BLOCK;
    db.usedBits = pos - db.entropyBit;
}

constexpr int HUF_DEC_CHUNKS = 8;   // chunks per wave (4 lanes each)

__global__ __launch_bounds__(64) void k_huff_decode(BitSrc src, DecBlock* __restrict__ blocks, int maxChunks, int nSlots,
                                                    const HufDecChunk* __restrict__ chunks, u8* const* __restrict__ outPtr)
{
    __shared__ u16 tables[HUF_DEC_CHUNKS][4096];      // (sym << 8) | len
    __shared__ u16 symCode[256];
    __shared__ int chunkErr[HUF_DEC_CHUNKS];
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
        u16* tab = tables[gg];
        for (int i = lane; i < 4096; i += 64) tab[i] = 0x0707;
        __syncthreads();
        if (lane == 0) {
            int code = 0, curLen = 0, first = 1, bad = 0;
            for (int l = 1; l <= HUF_MAX_LEN; l++) {
                for (int sy = 0; sy < 256; sy++) {
                    if (c.sizes[sy] != l) continue;
                    if (first) { curLen = l; first = 0; }
                    code <<= (l - curLen);
                    curLen = l;
                    symCode[sy] = (u16)code;
                    if (((code + 1) << (HUF_MAX_LEN - l)) > 4096) bad = 1;
                    code++;
                }
            }
            if (bad) chunkErr[gg] = 1;
        }
        __syncthreads();
        if (chunkErr[gg]) continue;
        // wide ranges cooperatively, narrow ranges per lane
        for (int sy = 0; sy < 256; sy++) {
            const u32 l = c.sizes[sy];
            if (l == 0 || l > 6) continue;
            const u32 w = 1u << (HUF_MAX_LEN - l);
            const u32 idx = (u32)symCode[sy] * w;
            const u16 val = (u16)((sy << 8) | l);
            for (u32 t = lane; t < w; t += 64) tab[idx + t] = val;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int sy = 4 * lane + k;
            const u32 l = c.sizes[sy];
            if (l > 6) {
                const u32 w = 1u << (HUF_MAX_LEN - l);
                const u32 idx = (u32)symCode[sy] * w;
                const u16 val = (u16)((sy << 8) | l);
                for (u32 t = 0; t < w; t++) tab[idx + t] = val;
            }
        }
        __syncthreads();
    }
    __syncthreads();

    // ---- one lane per fragment
    const int g = lane >> 2;
    const int j = lane & 3;
    if (g < HUF_DEC_CHUNKS) {
        const int slot = slotBase + g;
        if (slot < nSlots) {
            const HufDecChunk& c = chunks[slot];
            const int b = slot / maxChunks;
            const int ci = slot - b * maxChunks;
            if (c.kind == 0 && !blocks[b].error && !chunkErr[g]) {
                const u32 preLen = blocks[b].preLen;
                const u32 n = (preLen - (u32)ci * ENT_CHUNK < ENT_CHUNK) ? (preLen - (u32)ci * ENT_CHUNK) : ENT_CHUNK;
                const u32 szFrag = n / 4;
                u8* dst = outPtr[b] + (size_t)ci * ENT_CHUNK + (size_t)j * szFrag;
                const u16* tab = tables[g];
                // private view of the fragment: bits past its end read as zero (guard bytes of the reference)
                BitSrc fs = src;
                const u64 fbeg = c.fragBit[j];
                const u64 fend = fbeg + c.fragBits[j];
                u64 used = 0;
                bool bad = false;
                for (u32 i = 0; i < szFrag; i++) {
                    const u64 p = fbeg + used;
                    u32 win = peek_bits(fs, p, 12);
                    if (p + 12 > fend) {
                        const u32 valid = (p < fend) ? (u32)(fend - p) : 0u;
                        win &= ~((1u << (12 - valid)) - 1u);
                    }
                    const u16 val = tab[win];
                    dst[i] = (u8)(val >> 8);
                    used += (val & 0xFF);
                    if (used > (8192u << 3)) { bad = true; break; }
                }
                if (used != c.fragBits[j]) bad = true;
                if (j == 0) {
                    for (u32 i = 4 * szFrag; i < n; i++)
                        outPtr[b][(size_t)ci * ENT_CHUNK + i] = (u8)peek_bits(src, c.tailBit + 8ull * (i - 4 * szFrag), 8);
                }
                if (bad) chunkErr[g] = 1;
            }
        }
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

void launch_huffman_decode(hipStream_t s, BitSrc src, DecBlock* blocks, int nBlocks, int maxChunks, void* chunkMeta, u8* const* outPtr)
{
    HufDecChunk* chunks = reinterpret_cast<HufDecChunk*>(chunkMeta);
    const int nSlots = nBlocks * maxChunks;
    { KScope ks_("k_huff_scan"); hipLaunchKernelGGL(k_huff_scan, dim3((nBlocks + 63) / 64), dim3(64), 0, s, src, blocks, nBlocks, maxChunks, chunks); }
    { KScope ks_("k_huff_decode"); hipLaunchKernelGGL(k_huff_decode, dim3((nSlots + HUF_DEC_CHUNKS - 1) / HUF_DEC_CHUNKS), dim3(64), 0, s, src, blocks,
                       maxChunks, nSlots, chunks, outPtr); }
}

size_t huffman_dec_chunk_bytes() { return sizeof(HufDecChunk); }

}  // namespace knz
