/* TEST INFRASTRUCTURE ONLY (see knz_oracle.h).
 * Canonical Huffman (<= 12 bit codes, 16 KiB chunks, 4 fragments per chunk) restatement:
 *   encoder  entropy/HuffmanEncoder.cpp:58-126 (updateFrequencies), :129-215 (limitCodeLengths),
 *            :219-300 (computeCodeLengths + Moffat-Katajainen phases), :304-344 (encode),
 *            :348-421 (encodeChunk)
 *   common   entropy/HuffmanCommon.cpp:29-63 (generateCanonicalCodes)
 *   lengths  entropy/ExpGolombEncoder.hpp:51-62 / ExpGolombDecoder.hpp:52-75 (signed Exp-Golomb)
 *   decoder  entropy/HuffmanDecoder.cpp:65-108 (readLengths), :111-140 (buildDecodingTable),
 *            :156-201 (decodeV6), :204-347 (decodeChunk), :349-459 (decodeV5: bitstream versions below 6, one code stream per
 *            chunk behind a 2-bit stream count and a var-int bit count; no raw small chunks)
 * The reference only WRITES version 6. knzo_huffman_encode_bw under knzo_set_bs_version(< 6) is this file's own writer of the old
 * layout (what decodeV5 reads), there to hand-build old streams for the tests; the reference decoding them is what pins it.
 */
#include "knz_oracle.h"
#include <stdlib.h>
#include <string.h>

#define HUF_CHUNK 16384
#define HUF_MAX_SYMBOL_SIZE 12
#define HUF_MAX_CHUNK_SIZE (1 << 14)
#define HUF_TABLE_BITS 12

static int ilog2(uint32_t x) { return 31 ^ __builtin_clz(x); }

/* Signed Exp-Golomb of an int8 delta: ExpGolombEncoder.hpp:51-62 (the reference uses a
 * precomputed table whose entries equal: prefix of log2(|v|+1) zeros, |v|+1, sign bit). */
static void eg_encode_signed(knzo_bw* w, uint8_t val)
{
    if (val == 0) { knzo_bw_bits(w, 1, 1); return; }
    const int v = (int8_t)val;
    const uint32_t m = (uint32_t)(v < 0 ? -v : v);
    const uint32_t e = m + 1;
    const int L = ilog2(e);
    knzo_bw_bits(w, ((uint64_t)e << 1) | (v < 0 ? 1u : 0u), (unsigned)(2 * L + 2));
}

/* ExpGolombDecoder.hpp:52-75 */
static uint8_t eg_decode_signed(knzo_br* r)
{
    if (knzo_br_bits(r, 1) == 1) return 0;
    unsigned lg = 1;
    while (knzo_br_bits(r, 1) == 0) {
        lg++;
        if (r->error) return 0;
    }
    lg &= 7;
    int res = (int)knzo_br_bits(r, lg + 1);
    const int sgn = res & 1;
    res = (res >> 1) + (1 << lg) - 1;
    return (uint8_t)((res - sgn) ^ -sgn);
}

static int cmp_u32(const void* a, const void* b)
{
    const uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return (x > y) - (x < y);
}

/* HuffmanEncoder.cpp:246-270 */
static void phase1(uint32_t* data, int n)
{
    for (int s = 0, r = 0, t = 0; t < n - 1; t++) {
        uint32_t sum = 0;
        for (int i = 0; i < 2; i++) {
            if ((s >= n) || ((r < t) && (data[r] < data[s]))) {
                sum += data[r];
                data[r] = (uint32_t)t;
                r++;
                continue;
            }
            sum += data[s];
            if (s > t) data[s] = 0;
            s++;
        }
        data[t] = sum;
    }
}

/* HuffmanEncoder.cpp:274-300 */
static uint32_t phase2(uint32_t* data, int n)
{
    if (n < 2) return 0;
    uint32_t topLevel = (uint32_t)n - 2;
    uint32_t depth = 1;
    uint32_t totalNodesAtLevel = 2;
    while (n > 0) {
        uint32_t k = topLevel;
        while ((k != 0) && (data[k - 1] >= topLevel)) k--;
        const int internalNodesAtLevel = (int)(topLevel - k);
        const int leavesAtLevel = (int)totalNodesAtLevel - internalNodesAtLevel;
        for (int j = 0; j < leavesAtLevel; j++) data[--n] = depth;
        totalNodesAtLevel = (uint32_t)internalNodesAtLevel << 1;
        topLevel = k;
        depth++;
    }
    return depth - 1;
}

/* HuffmanEncoder.cpp:219-244 */
static int compute_code_lengths(uint16_t* sizes, uint32_t* ranks, int count)
{
    qsort(ranks, (size_t)count, sizeof(uint32_t), cmp_u32);
    uint32_t freqs[256];
    memset(freqs, 0, sizeof(freqs));
    int valid = 1;
    for (int i = 0; i < count; i++) {
        freqs[i] = ranks[i] >> 8;
        ranks[i] &= 0xFF;
        valid &= (freqs[i] != 0);
    }
    if (!valid) return 0;
    phase1(freqs, count);
    const int maxCodeLen = (int)phase2(freqs, count);
    for (int i = 0; i < count; i++) sizes[ranks[i]] = (uint16_t)freqs[i];
    return maxCodeLen;
}

/* HuffmanEncoder.cpp:129-215 */
static int limit_code_lengths(const uint32_t* alphabet, uint32_t* freqs, uint16_t* sizes, uint32_t* ranks, int count)
{
    int n = 0, debt = 0;
    while (sizes[ranks[n]] >= HUF_MAX_SYMBOL_SIZE) {
        debt += (sizes[ranks[n]] - HUF_MAX_SYMBOL_SIZE);
        sizes[ranks[n]] = HUF_MAX_SYMBOL_SIZE;
        n++;
    }
    if (debt == 0) return HUF_MAX_SYMBOL_SIZE;

    int v[6][256];
    int vLen[6] = { 0, 0, 0, 0, 0, 0 };
    int vHead[6] = { 0, 0, 0, 0, 0, 0 };

    while (n < count) {
        const int idx = HUF_MAX_SYMBOL_SIZE - 1 - sizes[ranks[n]];
        if ((idx > 5) || (debt < (1 << idx))) break;
        v[idx][vLen[idx]++] = n;
        n++;
    }

    int idx = 5;
    while ((debt > 0) && (idx >= 0)) {
        if ((vHead[idx] >= vLen[idx]) || (debt < (1 << idx))) { idx--; continue; }
        sizes[ranks[v[idx][vHead[idx]]]]++;
        debt -= (1 << idx);
        vHead[idx]++;
    }

    idx = 0;
    while ((debt > 0) && (idx < 6)) {
        if (vHead[idx] >= vLen[idx]) { idx++; continue; }
        sizes[ranks[v[idx][vHead[idx]]]]++;
        debt -= (1 << idx);
        vHead[idx]++;
    }

    if (debt > 0) {
        uint32_t alpha[256];
        uint32_t f[256];
        uint32_t totalFreq = 0;
        memset(alpha, 0, sizeof(alpha));
        for (int i = 0; i < count; i++) { f[i] = freqs[alphabet[i]]; totalFreq += f[i]; }
        knzo_normalize_freqs(f, alpha, count, totalFreq, HUF_MAX_CHUNK_SIZE >> 3);
        for (int i = 0; i < count; i++) {
            freqs[alphabet[i]] = f[i];
            ranks[i] = (f[i] << 8) | alphabet[i];
        }
        return compute_code_lengths(sizes, ranks, count);
    }
    return HUF_MAX_SYMBOL_SIZE;
}

/* HuffmanCommon.cpp:29-63 */
static int gen_canonical(const uint16_t* sizes, uint16_t* codes, uint32_t* symbols, int count)
{
    if (count == 0) return 0;
    if (count > 1) {
        int8_t buf[(HUF_MAX_SYMBOL_SIZE << 8) + 256];
        memset(buf, 0, sizeof(buf));
        for (int i = 0; i < count; i++) {
            const uint32_t s = symbols[i];
            if ((s > 255) || (sizes[s] > HUF_MAX_SYMBOL_SIZE)) return -1;
            buf[((sizes[s] - 1) << 8) | s] = 1;
        }
        for (int i = 0, n = 0; n < count; i++) {
            symbols[n] = (uint32_t)(i & 0xFF);
            n += buf[i];
        }
    }
    int curLen = sizes[symbols[0]];
    for (int i = 0, code = 0; i < count; i++) {
        const int s = (int)symbols[i];
        code <<= (sizes[s] - curLen);
        curLen = sizes[s];
        codes[s] = (uint16_t)code;
        code++;
    }
    return count;
}

/* HuffmanEncoder.cpp:58-126. codes[s] = (len << 12) | code. Returns alphabet size, -1 on error. */
static int update_frequencies(knzo_bw* w, uint32_t* freqs, uint16_t* codes)
{
    int count = 0;
    uint16_t sizes[256];
    uint32_t alphabet[256];
    memset(sizes, 0, sizeof(sizes));
    memset(alphabet, 0, sizeof(alphabet));
    for (int i = 0; i < 256; i++) {
        codes[i] = 0;
        if (freqs[i] > 0) alphabet[count++] = (uint32_t)i;
    }
    knzo_encode_alphabet(w, alphabet, count);
    if (count == 0) return 0;
    if (count == 1) {
        codes[alphabet[0]] = 1 << 12;
        sizes[alphabet[0]] = 1;
    } else {
        uint32_t ranks[256];
        for (int i = 0; i < count; i++) ranks[i] = (freqs[alphabet[i]] << 8) | alphabet[i];
        int maxCodeLen = compute_code_lengths(sizes, ranks, count);
        if (maxCodeLen == 0) return -1;
        if (maxCodeLen > HUF_MAX_SYMBOL_SIZE) {
            maxCodeLen = limit_code_lengths(alphabet, freqs, sizes, ranks, count);
            if (maxCodeLen == 0) return -1;
        }
        if (maxCodeLen > HUF_MAX_SYMBOL_SIZE) {
            uint16_t n = 0;
            for (int i = 0; i < count; i++) {
                codes[alphabet[i]] = n;
                sizes[alphabet[i]] = 8;
                n++;
            }
        } else {
            gen_canonical(sizes, codes, ranks, count);
        }
    }
    uint16_t prevSize = 2;
    for (int i = 0; i < count; i++) {
        const int s = (int)alphabet[i];
        codes[s] |= (uint16_t)(sizes[s] << 12);
        eg_encode_signed(w, (uint8_t)(sizes[s] - prevSize));
        prevSize = sizes[s];
    }
    return count;
}

/* HuffmanEncoder.cpp:304-421 */
static int huffman_encode_v5(knzo_bw* w, const uint8_t* block, uint32_t count)
{
    uint32_t startChunk = 0;
    uint16_t codes[256];
    const size_t cap = (size_t)HUF_CHUNK * 2 + 64;
    uint8_t* tmp = (uint8_t*)malloc(cap);
    while (startChunk < count) {
        const uint32_t sizeChunk = (HUF_CHUNK < count - startChunk) ? HUF_CHUNK : count - startChunk;
        const uint8_t* blk = &block[startChunk];
        uint32_t freqs[256];
        memset(freqs, 0, sizeof(freqs));
        for (uint32_t i = 0; i < sizeChunk; i++) freqs[blk[i]]++;
        const int asz = update_frequencies(w, freqs, codes);
        if (asz < 0) { free(tmp); return -1; }
        if (asz > 1) {
            knzo_bw cw;
            knzo_bw_init(&cw, tmp, cap);
            for (uint32_t i = 0; i < sizeChunk; i++) {
                const uint16_t c = codes[blk[i]];
                knzo_bw_bits(&cw, c & 0x0FFF, c >> 12);
            }
            knzo_bw_bits(w, 0, 2);                          /* number of streams - 1 */
            knzo_write_varint(w, (uint32_t)cw.bits);
            knzo_bw_bytes(w, cw.buf, cw.bits);
        }
        startChunk += sizeChunk;
    }
    free(tmp);
    return w->overflow ? -1 : (int)count;
}

int knzo_huffman_encode_bw(knzo_bw* w, const uint8_t* block, uint32_t count)
{
    if (count == 0) return 0;
    if (knzo_get_bs_version() < 6) return huffman_encode_v5(w, block, count);
    uint32_t startChunk = 0;
    uint16_t codes[256];
    uint8_t* frag = (uint8_t*)malloc(4 * (HUF_CHUNK / 4) * 2 + 64);
    while (startChunk < count) {
        const uint32_t sizeChunk = (HUF_CHUNK < count - startChunk) ? HUF_CHUNK : count - startChunk;
        const uint8_t* blk = &block[startChunk];
        if (sizeChunk < 32) {
            knzo_bw_bytes(w, blk, 8u * (uint64_t)sizeChunk);
        } else {
            uint32_t freqs[256];
            memset(freqs, 0, sizeof(freqs));
            for (uint32_t i = 0; i < sizeChunk; i++) freqs[blk[i]]++;
            const int asz = update_frequencies(w, freqs, codes);
            if (asz < 0) { free(frag); return -1; }
            if (asz > 1) {
                /* encodeChunk: 4 fragments of count/4 symbols, codes concatenated MSB-first */
                const uint32_t szFrag = sizeChunk / 4;
                uint32_t nbBits[4];
                knzo_bw fw[4];
                const size_t fcap = (size_t)(HUF_CHUNK / 4) * 2 + 16;
                for (int j = 0; j < 4; j++) {
                    knzo_bw_init(&fw[j], frag + (size_t)j * fcap, fcap);
                    const uint8_t* src = &blk[(uint32_t)j * szFrag];
                    for (uint32_t i = 0; i < szFrag; i++) {
                        const uint16_t c = codes[src[i]];
                        knzo_bw_bits(&fw[j], c & 0x0FFF, c >> 12);
                    }
                    nbBits[j] = (uint32_t)fw[j].bits;
                }
                for (int j = 0; j < 4; j++) knzo_write_varint(w, nbBits[j]);
                for (int j = 0; j < 4; j++) knzo_bw_bytes(w, fw[j].buf, nbBits[j]);
                for (uint32_t i = 4 * szFrag; i < sizeChunk; i++) knzo_bw_bits(w, blk[i], 8);
            }
        }
        startChunk += sizeChunk;
    }
    free(frag);
    return w->overflow ? -1 : (int)count;
}

/* HuffmanDecoder.cpp:349-459 */
static int huffman_decode_v5(knzo_br* r, uint8_t* block, uint32_t count)
{
    uint16_t codes[256], sizes[256];
    uint32_t alphabet[256];
    uint16_t* table = (uint16_t*)malloc(sizeof(uint16_t) * (1 << HUF_TABLE_BITS));
    uint8_t* buf = NULL;
    for (int i = 0; i < 256; i++) { codes[i] = (uint16_t)i; sizes[i] = 8; }
    memset(alphabet, 0, sizeof(alphabet));
    uint32_t startChunk = 0;
    int ret = (int)count;
    while (startChunk < count) {
        const uint32_t sizeChunk = (HUF_CHUNK < count - startChunk) ? HUF_CHUNK : count - startChunk;
        uint8_t* blk = &block[startChunk];
        /* readLengths :65-108 */
        const int asz = knzo_decode_alphabet(r, alphabet);
        if (r->error) { ret = -2; break; }
        if (asz <= 0) { ret = (int)startChunk; break; }
        int8_t curSize = 2;
        int bad = 0;
        for (int i = 0; i < asz; i++) {
            const uint32_t s = alphabet[i];
            codes[s] = 0;
            curSize = (int8_t)(curSize + (int8_t)eg_decode_signed(r));
            if (r->error || curSize <= 0 || curSize > HUF_MAX_SYMBOL_SIZE) { bad = 1; break; }
            sizes[s] = (uint16_t)curSize;
        }
        if (bad || gen_canonical(sizes, codes, alphabet, asz) < 0) { ret = -2; break; }
        if (asz == 1) { memset(blk, (int)alphabet[0], sizeChunk); startChunk += sizeChunk; continue; }
        /* buildDecodingTable :111-140 */
        for (int i = 0; i < (1 << HUF_TABLE_BITS); i++) table[i] = 0x0707;
        uint16_t length = 0;
        for (int i = 0; i < asz; i++) {
            const uint32_t s = alphabet[i];
            if (sizes[s] > length) length = sizes[s];
            const int wdt = 1 << (HUF_TABLE_BITS - length);
            int idx = (int)codes[s] * wdt;
            const int end = idx + wdt;
            if (end > (1 << HUF_TABLE_BITS)) { bad = 1; break; }
            const uint16_t val = (uint16_t)((s << 8) | sizes[s]);
            while (idx < end) table[idx++] = val;
        }
        if (bad) { ret = -1; break; }
        if (knzo_br_bits(r, 2) != 0 || r->error) { ret = r->error ? -2 : -1; break; }      /* one stream only, :375-377 */
        const int szBits = (int)knzo_read_varint(r);
        if (r->error) { ret = -2; break; }
        if (szBits < 0 || szBits > (int)sizeChunk * HUF_MAX_SYMBOL_SIZE) { ret = -1; break; }
        if (szBits != 0) {
            const size_t sz = ((size_t)szBits + 7) >> 3;
            buf = (uint8_t*)realloc(buf, sz + 8);
            memset(buf, 0, sz + 8);
            knzo_br_bytes(r, buf, (uint64_t)szBits);
            if (r->error) { ret = -2; break; }
            /* every symbol is one table look-up on the next 12 bits (zeros behind the end), :396-450; the chunk is good when
             * exactly szBits bits were used, :452-453 */
            uint64_t used = 0;
            for (uint32_t i = 0; i < sizeChunk; i++) {
                if (used > (uint64_t)szBits) { bad = 1; break; }
                const size_t b = (size_t)(used >> 3);
                uint32_t win = ((uint32_t)buf[b] << 16) | ((uint32_t)buf[b + 1] << 8) | buf[b + 2];
                win = (win >> (12 - (used & 7))) & 0xFFF;
                const uint16_t val = table[win];
                blk[i] = (uint8_t)(val >> 8);
                used += (val & 0xFF);
            }
            if (bad || used != (uint64_t)szBits) { ret = -1; break; }
        }
        startChunk += sizeChunk;
    }
    free(table); free(buf);
    return ret;
}

/* HuffmanDecoder.cpp:156-347 (v6 path) */
int knzo_huffman_decode_br(knzo_br* r, uint8_t* block, uint32_t count)
{
    if (count == 0) return 0;
    if (knzo_get_bs_version() < 6) return huffman_decode_v5(r, block, count);
    uint16_t codes[256], sizes[256];
    uint32_t alphabet[256];
    uint16_t* table = (uint16_t*)malloc(sizeof(uint16_t) * (1 << HUF_TABLE_BITS));
    const uint32_t bufferSize = 2 * HUF_CHUNK;     /* + guard bytes, :158 */
    const uint32_t fragCapacity = bufferSize >> 2;
    uint8_t* fbuf = (uint8_t*)malloc((size_t)fragCapacity + 32);
    for (int i = 0; i < 256; i++) { codes[i] = (uint16_t)i; sizes[i] = 8; }
    memset(alphabet, 0, sizeof(alphabet));
    uint32_t startChunk = 0;
    int ret = (int)count;

    while (startChunk < count) {
        const uint32_t sizeChunk = (HUF_CHUNK < count - startChunk) ? HUF_CHUNK : count - startChunk;
        uint8_t* blk = &block[startChunk];
        if (sizeChunk < 32) {
            knzo_br_bytes(r, blk, 8u * (uint64_t)sizeChunk);
            if (r->error) { ret = -2; break; }
        } else {
            /* readLengths :65-108 */
            const int asz = knzo_decode_alphabet(r, alphabet);
            if (r->error) { ret = -2; break; }
            if (asz <= 0) { ret = (int)startChunk; break; }
            int8_t curSize = 2;
            int bad = 0;
            for (int i = 0; i < asz; i++) {
                const uint32_t s = alphabet[i];
                codes[s] = 0;
                curSize = (int8_t)(curSize + (int8_t)eg_decode_signed(r));
                if (r->error || curSize <= 0 || curSize > HUF_MAX_SYMBOL_SIZE) { bad = 1; break; }
                sizes[s] = (uint16_t)curSize;
            }
            if (bad || gen_canonical(sizes, codes, alphabet, asz) < 0) { ret = -2; break; }
            if (asz == 1) {
                memset(blk, (int)alphabet[0], sizeChunk);
            } else {
                /* buildDecodingTable :111-140 */
                for (int i = 0; i < (1 << HUF_TABLE_BITS); i++) table[i] = 0x0707;
                uint16_t length = 0;
                for (int i = 0; i < asz; i++) {
                    const uint32_t s = alphabet[i];
                    if (sizes[s] > length) length = sizes[s];
                    const int wdt = 1 << (HUF_TABLE_BITS - length);
                    int idx = (int)codes[s] * wdt;
                    const int end = idx + wdt;
                    if (end > (1 << HUF_TABLE_BITS)) { bad = 1; break; }
                    const uint16_t val = (uint16_t)((s << 8) | sizes[s]);
                    while (idx < end) table[idx++] = val;
                }
                if (bad) { ret = -1; break; }
                /* decodeChunk :204-347 */
                int szBits[4];
                for (int j = 0; j < 4; j++) szBits[j] = (int)knzo_read_varint(r);
                if (r->error) { ret = -2; break; }
                const int maxFragBits = (int)(fragCapacity << 3);
                for (int j = 0; j < 4; j++)
                    if (szBits[j] < 0 || szBits[j] > maxFragBits) bad = 1;
                if (bad) { ret = -1; break; }
                const uint32_t szFrag = sizeChunk / 4;
                /* the reference reads all four fragments first, then decodes */
                uint64_t fragPos[4];
                for (int j = 0; j < 4; j++) {
                    fragPos[j] = r->pos;
                    if (r->pos + (uint64_t)szBits[j] > r->nbits) { r->error = 1; break; }
                    r->pos += (uint64_t)szBits[j];
                }
                if (r->error) { ret = -2; break; }
                for (int j = 0; j < 4 && !bad; j++) {
                    knzo_br fr;
                    knzo_br_init(&fr, r->buf, r->nbits);
                    fr.pos = fragPos[j];
                    memset(fbuf, 0, (size_t)fragCapacity + 32);
                    knzo_br_bytes(&fr, fbuf, (uint64_t)szBits[j]);
                    uint64_t used = 0;
                    uint8_t* dst = &blk[(uint32_t)j * szFrag];
                    for (uint32_t i = 0; i < szFrag; i++) {
                        /* peek 12 bits at 'used' (zero guard past the end) */
                        const size_t b = (size_t)(used >> 3);
                        uint32_t win = ((uint32_t)fbuf[b] << 16) | ((uint32_t)fbuf[b + 1] << 8) | fbuf[b + 2];
                        win = (win >> (12 - (used & 7))) & 0xFFF;
                        const uint16_t val = table[win];
                        dst[i] = (uint8_t)(val >> 8);
                        used += (val & 0xFF);
                        if (used > (uint64_t)maxFragBits) { bad = 1; break; }
                    }
                    if (used != (uint64_t)szBits[j]) bad = 1;
                }
                for (uint32_t i = 4 * szFrag; i < sizeChunk; i++) blk[i] = (uint8_t)knzo_br_bits(r, 8);
                if (r->error) { ret = -2; break; }
                if (bad) { ret = -1; break; }
            }
        }
        startChunk += sizeChunk;
    }
    free(table); free(fbuf);
    return ret;
}
