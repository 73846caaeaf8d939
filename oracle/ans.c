/* TEST INFRASTRUCTURE ONLY (see knz_oracle.h).
 * rANS order-0 / order-1 restatement:
 *   encoder  entropy/ANSRangeEncoder.cpp:83-116 (updateFrequencies), :119-155 (encodeHeader),
 *            :158-192 (encode), :194-261 (encodeChunk), :264-287 (rebuildStatistics),
 *            entropy/ANSRangeEncoder.hpp:92-117 (ANSEncSymbol::reset), :119-131 (encodeSymbol)
 *   decoder  entropy/ANSRangeDecoder.cpp:80-175 (decodeHeader), :177-216 (decode), :218-292 (decodeChunk),
 *            entropy/ANSRangeDecoder.hpp:85-103
 *   helpers  entropy/EntropyUtils.cpp:131-245 (normalizeFrequencies), Global.cpp:170-310 (histograms)
 */
#include "knz_oracle.h"
#include <stdlib.h>
#include <string.h>

#define ANS_TOP (1 << 15)
#define ANS0_CHUNK 16384
#define ANS_LOG_RANGE 12
#define ANS_MAX_CHUNK (1 << 27)

static int ilog2(uint32_t x) { return 31 ^ __builtin_clz(x); }

/* entropy/EntropyUtils.cpp:131-245. Order-sensitive heuristic, control flow kept as is. */
int knzo_normalize_freqs(uint32_t* freqs, uint32_t* alphabet, int length, uint32_t totalFreq, uint32_t scale)
{
    if (length > 256 || scale < 256 || scale > 65536) return -1;
    if (length == 0 || totalFreq == 0) return 0;
    int alphabetSize = 0;

    if (totalFreq == scale) {
        for (int i = 0; i < 256; i++)
            if (freqs[i] != 0) alphabet[alphabetSize++] = (uint32_t)i;
        return alphabetSize;
    }

    uint32_t sumScaledFreq = 0, sumFreq = 0;
    int idxMax = 0;

    for (int i = 0; i < length; i++) {
        alphabet[i] = 0;
        const uint32_t f = freqs[i];
        if (f == 0) continue;
        alphabet[alphabetSize++] = (uint32_t)i;
        const int64_t sf = (int64_t)f * (int64_t)scale;
        const uint32_t scaledFreq = (sf <= (int64_t)totalFreq) ? 1u
            : (uint32_t)((sf + ((int64_t)totalFreq >> 1)) / (int64_t)totalFreq);
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

    if (abs(delta) <= errThr) {
        freqs[idxMax] -= (uint32_t)delta;
        return alphabetSize;
    }

    if (delta < 0) { delta += errThr; freqs[idxMax] += (uint32_t)errThr; }
    else           { delta -= errThr; freqs[idxMax] -= (uint32_t)errThr; }

    const int inc = (delta < 0) ? 1 : -1;
    delta = abs(delta);
    int round = 0;

    while ((++round < 6) && (delta > 0)) {
        int adjustments = 0;
        for (int i = 0; i < alphabetSize; i++) {
            const int idx = (int)alphabet[i];
            if (freqs[idx] <= 2) continue;
            freqs[idx] += (uint32_t)inc;
            adjustments++;
            delta--;
            if (delta == 0) break;
        }
        if (adjustments == 0) break;
    }

    {   /* freqs[idxMax] = max(freqs[idxMax] - delta, uint(1)) in uint32 arithmetic */
        uint32_t v = freqs[idxMax] - (uint32_t)delta;
        freqs[idxMax] = v > 1u ? v : 1u;
    }
    return alphabetSize;
}

typedef struct {
    int xMax, bias, cmplFreq, invShift;
    uint64_t invFreq;
} enc_sym;

/* ANSRangeEncoder.hpp:92-117 */
static void enc_sym_reset(enc_sym* s, int cumFreq, int freq, unsigned lr)
{
    if (freq >= (1 << lr)) freq = (1 << lr) - 1;
    s->xMax = ((ANS_TOP >> lr) << 16) * freq;
    s->cmplFreq = (1 << lr) - freq;
    if (freq < 2) {
        s->invFreq = 0xFFFFFFFFull;
        s->invShift = 32;
        s->bias = cumFreq + (1 << lr) - 1;
    } else {
        int shift = 0;
        while (freq > (1 << shift)) shift++;
        s->invFreq = ((((uint64_t)1 << (shift + 31)) + (uint64_t)freq - 1) / (uint64_t)freq) & 0xFFFFFFFFull;
        s->invShift = 32 + shift - 1;
        s->bias = cumFreq;
    }
}

/* ANSRangeEncoder.hpp:119-131 */
static inline int enc_symbol(uint8_t** pp, int st, const enc_sym* sym)
{
    uint8_t* p = *pp;
    const int x = (st >= sym->xMax) ? 1 : 0;
    *p = (uint8_t)st;
    p -= x;
    *p = (uint8_t)(st >> 8);
    p -= x;
    *pp = p;
    st >>= (-x & 16);
    return st + sym->bias + (int)(((uint64_t)(uint32_t)st * sym->invFreq) >> sym->invShift) * sym->cmplFreq;
}

/* Global.cpp:223-271: order-1 histogram with totals, stride 257 */
static void histo_o1_total(const uint8_t* p, int length, uint32_t* freqs)
{
    if (length <= 0) return;
    const int quarter = length >> 2;
    int n0 = 0, n1 = quarter, n2 = 2 * quarter, n3 = 3 * quarter;
    if (length < 32) {
        uint32_t prv = 0;
        for (int i = 0; i < length; i++) {
            freqs[prv + p[i]]++;
            freqs[prv + 256]++;
            prv = 257u * p[i];
        }
        return;
    }
    uint32_t prv0 = 0, prv1 = 257u * p[n1 - 1], prv2 = 257u * p[n2 - 1], prv3 = 257u * p[n3 - 1];
    for (; n0 < quarter; n0++, n1++, n2++, n3++) {
        const uint32_t c0 = p[n0], c1 = p[n1], c2 = p[n2], c3 = p[n3];
        freqs[prv0 + c0]++; freqs[prv0 + 256]++;
        freqs[prv1 + c1]++; freqs[prv1 + 256]++;
        freqs[prv2 + c2]++; freqs[prv2 + 256]++;
        freqs[prv3 + c3]++; freqs[prv3 + 256]++;
        prv0 = 257u * c0; prv1 = 257u * c1; prv2 = 257u * c2; prv3 = 257u * c3;
    }
    for (; n3 < length; n3++) {
        freqs[prv3 + p[n3]]++;
        freqs[prv3 + 256]++;
        prv3 = 257u * p[n3];
    }
}

/* ANSRangeEncoder.cpp:119-155 */
static void enc_header(knzo_bw* w, int alphabetSize, const uint32_t* alphabet, const uint32_t* f, unsigned lr)
{
    const int encoded = knzo_encode_alphabet(w, alphabet, alphabetSize);
    if (encoded <= 1) return;
    const int chkSize = (alphabetSize >= 64) ? 8 : 6;
    const int llr = ilog2(lr) + 1;
    for (int i = 1; i < alphabetSize; i += chkSize) {
        uint32_t max = f[alphabet[i]] - 1;
        const int endj = (i + chkSize < alphabetSize) ? i + chkSize : alphabetSize;
        for (int j = i + 1; j < endj; j++)
            if (f[alphabet[j]] - 1 > max) max = f[alphabet[j]] - 1;
        const unsigned logMax = (max == 0) ? 0 : (unsigned)ilog2(max) + 1;
        knzo_bw_bits(w, logMax, (unsigned)llr);
        if (logMax == 0) continue;
        for (int j = i; j < endj; j++)
            knzo_bw_bits(w, f[alphabet[j]] - 1, logMax);
    }
}

/* ANSRangeEncoder.cpp:83-116 */
static int update_freqs(knzo_bw* w, uint32_t* frequencies, enc_sym* symbols, int order, unsigned lr)
{
    int res = 0;
    const int endk = 255 * order + 1;
    uint32_t curAlphabet[256];
    knzo_bw_bits(w, lr - 8, 3);
    for (int k = 0; k < endk; k++) {
        uint32_t* f = &frequencies[k * 257];
        const int alphabetSize = knzo_normalize_freqs(f, curAlphabet, 256, f[256], 1u << lr);
        if (alphabetSize > 0) {
            enc_sym* symb = &symbols[k << 8];
            int sum = 0;
            for (int i = 0, count = 0; i < 256; i++) {
                if (f[i] == 0) continue;
                enc_sym_reset(&symb[i], sum, (int)f[i], lr);
                sum += (int)f[i];
                count++;
                if (count >= alphabetSize) break;
            }
        }
        enc_header(w, alphabetSize, curAlphabet, f, lr);
        res += alphabetSize;
    }
    return res;
}

/* ANSRangeEncoder.cpp:158-287 */
int knzo_ans_encode_bw(knzo_bw* w, const uint8_t* block, uint32_t count, int order)
{
    if (count <= 32) {
        knzo_bw_bytes(w, block, 8u * (uint64_t)count);
        return (int)count;
    }
    uint64_t scaled = (uint64_t)ANS0_CHUNK << (8 * order);
    const uint32_t sz = (uint32_t)(scaled < ANS_MAX_CHUNK ? scaled : ANS_MAX_CHUNK);
    const unsigned lr = (order == 0) ? ANS_LOG_RANGE : ANS_LOG_RANGE - 1;
    uint32_t size = sz + (sz >> 3);
    if (size > 2 * count) size = 2 * count;
    if (size < 65536) size = 65536;
    const int dim = 255 * order + 1;
    uint8_t* buffer = (uint8_t*)malloc(size);
    enc_sym* symbols = (enc_sym*)malloc(sizeof(enc_sym) * (size_t)dim * 256);
    uint32_t* freqs = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)dim * 257);
    uint32_t startChunk = 0;

    while (startChunk < count) {
        const uint32_t sizeChunk = (sz < count - startChunk) ? sz : count - startChunk;
        const uint8_t* blk = &block[startChunk];
        const int end = (int)sizeChunk;
        memset(freqs, 0, sizeof(uint32_t) * (size_t)dim * 257);
        if (order == 0) {
            for (int i = 0; i < end; i++) freqs[blk[i]]++;
            freqs[256] = (uint32_t)end;
        } else {
            const int quarter = end >> 2;
            if (quarter == 0) histo_o1_total(blk, end, freqs);
            else {
                histo_o1_total(&blk[0 * quarter], quarter, freqs);
                histo_o1_total(&blk[1 * quarter], quarter, freqs);
                histo_o1_total(&blk[2 * quarter], quarter, freqs);
                histo_o1_total(&blk[3 * quarter], quarter, freqs);
            }
        }
        const int alphabetSize = update_freqs(w, freqs, symbols, order, lr);
        if (alphabetSize <= 1 && order == 0) { startChunk += sizeChunk; continue; }

        /* encodeChunk :194-261 */
        int st0 = ANS_TOP, st1 = ANS_TOP, st2 = ANS_TOP, st3 = ANS_TOP;
        uint8_t* p = &buffer[size - 1];
        const uint8_t* p0 = p;
        const int end4 = end & -4;
        for (int i = end - 1; i >= end4; i--) *p-- = blk[i];
        if (order == 0) {
            for (int i = end4 - 1; i > 0; i -= 4) {
                st0 = enc_symbol(&p, st0, &symbols[blk[i]]);
                st1 = enc_symbol(&p, st1, &symbols[blk[i - 1]]);
                st2 = enc_symbol(&p, st2, &symbols[blk[i - 2]]);
                st3 = enc_symbol(&p, st3, &symbols[blk[i - 3]]);
            }
        } else {
            const int quarter = end4 >> 2;
            int i0 = 1 * quarter - 2, i1 = 2 * quarter - 2, i2 = 3 * quarter - 2, i3 = end4 - 2;
            int prv0 = blk[i0 + 1], prv1 = blk[i1 + 1], prv2 = blk[i2 + 1], prv3 = blk[i3 + 1];
            for (; i0 >= 0; i0--, i1--, i2--, i3--) {
                const int cur0 = blk[i0];
                st0 = enc_symbol(&p, st0, &symbols[(cur0 << 8) | prv0]);
                const int cur1 = blk[i1];
                st1 = enc_symbol(&p, st1, &symbols[(cur1 << 8) | prv1]);
                const int cur2 = blk[i2];
                st2 = enc_symbol(&p, st2, &symbols[(cur2 << 8) | prv2]);
                const int cur3 = blk[i3];
                st3 = enc_symbol(&p, st3, &symbols[(cur3 << 8) | prv3]);
                prv0 = cur0; prv1 = cur1; prv2 = cur2; prv3 = cur3;
            }
            st0 = enc_symbol(&p, st0, &symbols[prv0]);
            st1 = enc_symbol(&p, st1, &symbols[prv1]);
            st2 = enc_symbol(&p, st2, &symbols[prv2]);
            st3 = enc_symbol(&p, st3, &symbols[prv3]);
        }
        knzo_write_varint(w, (uint32_t)(p0 - p));
        knzo_bw_bits(w, (uint32_t)st0, 32);
        knzo_bw_bits(w, (uint32_t)st1, 32);
        knzo_bw_bits(w, (uint32_t)st2, 32);
        knzo_bw_bits(w, (uint32_t)st3, 32);
        if (p != p0) knzo_bw_bytes(w, &p[1], 8u * (uint64_t)(p0 - p));
        startChunk += sizeChunk;
    }
    free(buffer); free(symbols); free(freqs);
    return w->overflow ? -1 : (int)count;
}

typedef struct { uint16_t cumFreq, freq; } dec_sym;

/* ANSRangeDecoder.cpp:80-175. Returns alphabet size sum, or -1 on an invalid stream. */
static int dec_header(knzo_br* r, uint32_t* frequencies, uint32_t* alphabet, dec_sym* symbols,
                      uint8_t** f2s, size_t* f2sSize, int order, unsigned* logRange)
{
    const unsigned lr = 8 + (unsigned)knzo_br_bits(r, 3);
    *logRange = lr;
    if (lr > 15 || r->error) return -1;
    int res = 0;
    const int dim = 255 * order + 1;
    if (*f2sSize < ((size_t)dim << lr)) {
        free(*f2s);
        *f2sSize = (size_t)dim << lr;
        *f2s = (uint8_t*)malloc(*f2sSize);
    }
    const uint32_t scale = 1u << lr;
    const int llr = ilog2(lr) + 1;
    for (int k = 0; k < dim; k++) {
        const int alphabetSize = knzo_decode_alphabet(r, alphabet);
        if (r->error) return -1;
        if (alphabetSize == 0) continue;
        uint32_t* f = &frequencies[k << 8];
        if (alphabetSize != 256) memset(f, 0, sizeof(uint32_t) * 256);
        const int chkSize = (alphabetSize >= 64) ? 8 : 6;
        uint32_t sum = 0;
        for (int i = 1; i < alphabetSize; i += chkSize) {
            const unsigned logMax = (unsigned)knzo_br_bits(r, (unsigned)llr);
            if (logMax > lr || r->error) return -1;
            const int endj = (i + chkSize < alphabetSize) ? i + chkSize : alphabetSize;
            for (int j = i; j < endj; j++) {
                const uint32_t freq = (logMax == 0) ? 1u : (uint32_t)(knzo_br_bits(r, logMax) + 1);
                if (freq >= scale || r->error) return -1;
                f[alphabet[j]] = freq;
                sum += freq;
            }
        }
        if (scale <= sum) return -1;
        f[alphabet[0]] = scale - sum;
        sum = 0;
        dec_sym* symb = &symbols[k << 8];
        uint8_t* freq2sym = &(*f2s)[(size_t)k << lr];
        for (int i = 0; i < 256; i++) {
            if (f[i] == 0) continue;
            memset(&freq2sym[sum], i, f[i]);
            symb[i].cumFreq = (uint16_t)sum;
            symb[i].freq = (f[i] >= scale) ? (uint16_t)(scale - 1) : (uint16_t)f[i];
            sum += f[i];
        }
        res += alphabetSize;
    }
    return res;
}

/* ANSRangeDecoder.hpp:92-103 */
static inline uint32_t dec_symbol(const uint8_t** pp, uint32_t st, const dec_sym* sym, uint32_t mask, unsigned lr)
{
    const uint8_t* p = *pp;
    st = (uint32_t)sym->freq * (st >> lr) + (st & mask) - (uint32_t)sym->cumFreq;
    const int x = (st < ANS_TOP) ? -1 : 0;
    st = (st << (x & 16)) | ((uint32_t)x & (((uint32_t)p[0] << 8) | (uint32_t)p[1]));
    p -= (x + x);
    *pp = p;
    return st;
}

/* ANSRangeDecoder.cpp:177-292 */
int knzo_ans_decode_br(knzo_br* r, uint8_t* block, uint32_t count, int order)
{
    if (count <= 32) {
        knzo_br_bytes(r, block, 8u * (uint64_t)count);
        return r->error ? -1 : (int)count;
    }
    uint64_t scaled = (uint64_t)ANS0_CHUNK << (8 * order);
    const uint32_t chunkSize = (uint32_t)(scaled < ANS_MAX_CHUNK ? scaled : ANS_MAX_CHUNK);
    const uint32_t bufferSize = 2 * chunkSize;
    const int dim = 255 * order + 1;
    uint8_t* buffer = (uint8_t*)malloc((size_t)bufferSize + 8);
    uint32_t* freqs = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)dim * 256);
    dec_sym* symbols = (dec_sym*)malloc(sizeof(dec_sym) * (size_t)dim * 256);
    uint8_t* f2s = NULL;
    size_t f2sSize = 0;
    uint32_t alphabet[256];
    uint32_t startChunk = 0;
    int ret = (int)count;

    while (startChunk < count) {
        const uint32_t sizeChunk = (chunkSize < count - startChunk) ? chunkSize : count - startChunk;
        unsigned lr;
        const int alphabetSize = dec_header(r, freqs, alphabet, symbols, &f2s, &f2sSize, order, &lr);
        if (alphabetSize < 0) { ret = -2; break; }           /* reference throws BitStreamException */
        if (alphabetSize == 0) { ret = (int)startChunk; break; }
        uint8_t* blk = &block[startChunk];
        if (order == 0 && alphabetSize == 1) {
            memset(blk, (int)alphabet[0], sizeChunk);
        } else {
            const uint32_t sz = knzo_read_varint(r);
            if (r->error) { ret = -2; break; }
            if (sz >= ANS_MAX_CHUNK || sz > bufferSize - 2) { ret = -1; break; }
            uint32_t st0 = (uint32_t)knzo_br_bits(r, 32);
            uint32_t st1 = (uint32_t)knzo_br_bits(r, 32);
            uint32_t st2 = (uint32_t)knzo_br_bits(r, 32);
            uint32_t st3 = (uint32_t)knzo_br_bits(r, 32);
            memset(buffer, 0, (size_t)bufferSize + 8);
            knzo_br_bytes(r, buffer, 8u * (uint64_t)sz);
            if (r->error) { ret = -2; break; }
            const uint8_t* p = buffer;
            const uint8_t* endPayload = &buffer[sz];
            const uint32_t mask = (1u << lr) - 1;
            const int count4 = (int)sizeChunk & -4;
            if (order == 0) {
                for (int i = 0; i < count4; i += 4) {
                    const uint8_t cur3 = f2s[st3 & mask];
                    blk[i] = cur3;
                    st3 = dec_symbol(&p, st3, &symbols[cur3], mask, lr);
                    const uint8_t cur2 = f2s[st2 & mask];
                    blk[i + 1] = cur2;
                    st2 = dec_symbol(&p, st2, &symbols[cur2], mask, lr);
                    const uint8_t cur1 = f2s[st1 & mask];
                    blk[i + 2] = cur1;
                    st1 = dec_symbol(&p, st1, &symbols[cur1], mask, lr);
                    const uint8_t cur0 = f2s[st0 & mask];
                    blk[i + 3] = cur0;
                    st0 = dec_symbol(&p, st0, &symbols[cur0], mask, lr);
                }
            } else {
                const int quarter = count4 >> 2;
                int i0 = 0, i1 = quarter, i2 = 2 * quarter, i3 = 3 * quarter;
                int prv0 = 0, prv1 = 0, prv2 = 0, prv3 = 0;
                for (; i0 < quarter; i0++, i1++, i2++, i3++) {
                    const uint8_t cur3 = f2s[((size_t)prv3 << lr) + (st3 & mask)];
                    const uint8_t cur2 = f2s[((size_t)prv2 << lr) + (st2 & mask)];
                    const uint8_t cur1 = f2s[((size_t)prv1 << lr) + (st1 & mask)];
                    const uint8_t cur0 = f2s[((size_t)prv0 << lr) + (st0 & mask)];
                    st3 = dec_symbol(&p, st3, &symbols[(prv3 << 8) | cur3], mask, lr);
                    st2 = dec_symbol(&p, st2, &symbols[(prv2 << 8) | cur2], mask, lr);
                    st1 = dec_symbol(&p, st1, &symbols[(prv1 << 8) | cur1], mask, lr);
                    st0 = dec_symbol(&p, st0, &symbols[(prv0 << 8) | cur0], mask, lr);
                    blk[i3] = cur3; blk[i2] = cur2; blk[i1] = cur1; blk[i0] = cur0;
                    prv3 = cur3; prv2 = cur2; prv1 = cur1; prv0 = cur0;
                }
            }
            for (uint32_t i = (uint32_t)count4; i < sizeChunk; i++) blk[i] = *p++;
            if (p != endPayload) { ret = -1; break; }
        }
        startChunk += sizeChunk;
    }
    free(buffer); free(freqs); free(symbols); free(f2s);
    return ret;
}
