/* TEST INFRASTRUCTURE ONLY (see knz_oracle.h).
 * FPAQ adaptive order-0 binary arithmetic coder restatement:
 *   encoder entropy/FPAQEncoder.cpp:41-110, entropy/FPAQEncoder.hpp:72-94
 *   decoder entropy/FPAQDecoder.cpp:41-120, entropy/FPAQDecoder.hpp:74-117
 * State (low, high, probs) carries over the 4 MiB sub-chunks; only the table pointer resets.
 * Also the NONE entropy codec (entropy/NullEntropyEncoder.hpp / NullEntropyDecoder.hpp): raw bytes.
 */
#include "knz_oracle.h"
#include <stdlib.h>
#include <string.h>

#define FPAQ_TOP 0x00FFFFFFFFFFFFFFull
#define MASK_0_24 0x0000000000FFFFFFull
#define MASK_0_32 0x00000000FFFFFFFFull
#define MASK_0_56 0x00FFFFFFFFFFFFFFull
#ifndef FPAQ_CHUNK            /* tests/test_emu_kernels.py builds a copy with a small value to cross sub-chunk borders quickly */
#define FPAQ_CHUNK (4u * 1024 * 1024)
#endif
#define PSCALE 65536

typedef struct {
    uint64_t low, high;
    uint8_t* buf;
    uint32_t index;
    uint16_t probs[4][256];
} fpaq_enc;

static inline void enc_bit(fpaq_enc* e, int bit, uint16_t* prob)
{
    if (bit == 0) {
        e->low = e->low + ((((e->high - e->low) >> 8) * (uint64_t)*prob) >> 8) + 1;
        *prob = (uint16_t)(*prob - (uint16_t)(*prob >> 6));
    } else {
        e->high = e->low + ((((e->high - e->low) >> 8) * (uint64_t)*prob) >> 8);
        *prob = (uint16_t)(*prob - (uint16_t)(((int)*prob - PSCALE + 64) >> 6));
    }
    if (((e->low ^ e->high) >> 24) == 0) {
        const uint32_t v = (uint32_t)(e->high >> 24);
        e->buf[e->index] = (uint8_t)(v >> 24);
        e->buf[e->index + 1] = (uint8_t)(v >> 16);
        e->buf[e->index + 2] = (uint8_t)(v >> 8);
        e->buf[e->index + 3] = (uint8_t)v;
        e->index += 4;
        e->low <<= 32;
        e->high = (e->high << 32) | MASK_0_32;
    }
}

int knzo_fpaq_encode_bw(knzo_bw* w, const uint8_t* block, uint32_t count)
{
    if (count >= (1u << 30)) return -1;
    fpaq_enc e;
    e.low = 0; e.high = FPAQ_TOP; e.index = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 256; j++) e.probs[i][j] = PSCALE >> 1;
    e.buf = (uint8_t*)malloc(FPAQ_CHUNK + (FPAQ_CHUNK >> 3) + 8);
    uint32_t startChunk = 0;
    while (startChunk < count) {
        const uint32_t chunkSize = (FPAQ_CHUNK < count - startChunk) ? FPAQ_CHUNK : count - startChunk;
        e.index = 0;
        const uint32_t endChunk = startChunk + chunkSize;
        uint16_t* p = e.probs[0];
        for (uint32_t i = startChunk; i < endChunk; i++) {
            const int val = block[i];
            const int bits = val + 256;
            enc_bit(&e, val & 0x80, &p[1]);
            enc_bit(&e, val & 0x40, &p[bits >> 7]);
            enc_bit(&e, val & 0x20, &p[bits >> 6]);
            enc_bit(&e, val & 0x10, &p[bits >> 5]);
            enc_bit(&e, val & 0x08, &p[bits >> 4]);
            enc_bit(&e, val & 0x04, &p[bits >> 3]);
            enc_bit(&e, val & 0x02, &p[bits >> 2]);
            enc_bit(&e, val & 0x01, &p[bits >> 1]);
            p = e.probs[val >> 6];
        }
        knzo_write_varint(w, e.index);
        knzo_bw_bytes(w, e.buf, 8u * (uint64_t)e.index);
        startChunk += chunkSize;
        if (startChunk < count) knzo_bw_bits(w, (e.low | MASK_0_24) & MASK_0_56, 56);
    }
    /* dispose(): FPAQEncoder.cpp:103-110 */
    knzo_bw_bits(w, (e.low | MASK_0_24) & MASK_0_56, 56);
    free(e.buf);
    return w->overflow ? -1 : (int)count;
}

int knzo_fpaq_decode_br(knzo_br* r, uint8_t* block, uint32_t count)
{
    if (count >= (1u << 30)) return -1;
    uint64_t low = 0, high = FPAQ_TOP, current = 0;
    uint16_t probs[4][256];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 256; j++) probs[i][j] = PSCALE >> 1;
    uint8_t* buf = NULL;
    size_t bufCap = 0;
    uint32_t startChunk = 0;
    int ret = (int)count;

    while (startChunk < count) {
        const uint32_t szBytes = knzo_read_varint(r);
        if (r->error) { ret = -2; break; }
        if (szBytes >= 2 * count) { ret = 0; break; }
        size_t bufSize = (size_t)szBytes + (szBytes >> 3);
        if (bufSize < 8192) bufSize = 8192;
        if (bufCap < bufSize) { free(buf); buf = (uint8_t*)malloc(bufSize); bufCap = bufSize; }
        current = knzo_br_bits(r, 56);
        memset(buf, 0, bufSize);
        knzo_br_bytes(r, buf, 8u * (uint64_t)szBytes);
        if (r->error) { ret = -2; break; }
        uint32_t index = 0;
        const uint32_t chunkSize = (FPAQ_CHUNK < count - startChunk) ? FPAQ_CHUNK : count - startChunk;
        const uint32_t endChunk = startChunk + chunkSize;
        uint16_t* p = probs[0];
        int fail = 0;
        for (uint32_t i = startChunk; i < endChunk; i++) {
            int ctx = 1;
            for (int k = 0; k < 8; k++) {
                const uint64_t split = ((((high - low) >> 8) * (uint64_t)p[ctx]) >> 8) + low;
                if (split >= current) {
                    high = split;
                    p[ctx] = (uint16_t)(p[ctx] - (uint16_t)(((int)p[ctx] - PSCALE + 64) >> 6));
                    ctx += ctx + 1;
                } else {
                    low = split + 1;
                    p[ctx] = (uint16_t)(p[ctx] - (uint16_t)(p[ctx] >> 6));
                    ctx += ctx;
                }
                if (((low ^ high) >> 24) == 0) {
                    low = (low << 32) & MASK_0_56;
                    high = ((high << 32) | MASK_0_32) & MASK_0_56;
                    if (index + 4 > szBytes) {
                        current = (current << 32) & MASK_0_56;
                        index = szBytes + 1;
                    } else {
                        const uint64_t val = ((uint64_t)buf[index] << 24) | ((uint64_t)buf[index + 1] << 16) |
                                             ((uint64_t)buf[index + 2] << 8) | (uint64_t)buf[index + 3];
                        current = ((current << 32) | val) & MASK_0_56;
                        index += 4;
                    }
                }
            }
            block[i] = (uint8_t)ctx;
            if (index > szBytes) { fail = 1; break; }
            p = probs[(ctx & 0xFF) >> 6];
        }
        if (fail || index > szBytes) { ret = 0; break; }
        startChunk = endChunk;
    }
    free(buf);
    return ret;
}

int knzo_none_encode_bw(knzo_bw* w, const uint8_t* in, uint32_t n)
{
    knzo_bw_bytes(w, in, 8u * (uint64_t)n);
    return w->overflow ? -1 : (int)n;
}

int knzo_none_decode_br(knzo_br* r, uint8_t* out, uint32_t n)
{
    knzo_br_bytes(r, out, 8u * (uint64_t)n);
    return r->error ? -1 : (int)n;
}

int64_t knzo_entropy_encode(int etype, const uint8_t* in, uint32_t n, uint8_t* out, size_t cap)
{
    knzo_bw w;
    knzo_bw_init(&w, out, cap);
    int r;
    switch (etype) {
    case 0: r = knzo_none_encode_bw(&w, in, n); break;
    case 1: r = knzo_huffman_encode_bw(&w, in, n); break;
    case 2: r = knzo_fpaq_encode_bw(&w, in, n); break;
    case 5: r = knzo_ans_encode_bw(&w, in, n, 0); break;
    case 8: r = knzo_ans_encode_bw(&w, in, n, 1); break;
    default: return -1;
    }
    if (r != (int)n || w.overflow) return -1;
    return (int64_t)w.bits;
}

int knzo_entropy_decode(int etype, const uint8_t* in, size_t inBytes, uint8_t* out, uint32_t n)
{
    knzo_br r;
    knzo_br_init(&r, in, 8u * (uint64_t)inBytes);
    switch (etype) {
    case 0: return knzo_none_decode_br(&r, out, n);
    case 1: return knzo_huffman_decode_br(&r, out, n);
    case 2: return knzo_fpaq_decode_br(&r, out, n);
    case 5: return knzo_ans_decode_br(&r, out, n, 0);
    case 8: return knzo_ans_decode_br(&r, out, n, 1);
    default: return -1;
    }
}
