/* TEST INFRASTRUCTURE ONLY (see knz_oracle.h).
 * MSB-first bit writer/reader; restates the observable behaviour of
 * bitstream/DefaultOutputBitStream.hpp:83-131 (.cpp:42-128 for byte arrays) and
 * bitstream/DefaultInputBitStream.hpp:88-150: bits are packed MSB-first into bytes,
 * a trailing partial byte is zero padded, written() is the exact bit count.
 * Also EntropyUtils var-ints and alphabet coding (entropy/EntropyUtils.cpp:57-123,247-285). */
#include "knz_oracle.h"
#include <string.h>

void knzo_bw_init(knzo_bw* w, uint8_t* buf, size_t cap)
{
    w->buf = buf; w->cap = cap; w->bits = 0; w->overflow = 0;
}

void knzo_bw_bits(knzo_bw* w, uint64_t v, unsigned n)
{
    if (n == 0 || n > 64) return;
    if ((w->bits + n + 7) / 8 > w->cap) { w->overflow = 1; return; }
    /* write bit by byte pieces */
    unsigned left = n;
    while (left > 0) {
        size_t byte = (size_t)(w->bits >> 3);
        unsigned used = (unsigned)(w->bits & 7);
        unsigned room = 8 - used;
        unsigned take = left < room ? left : room;
        unsigned piece = (unsigned)((v >> (left - take)) & ((1u << take) - 1u));
        if (used == 0) w->buf[byte] = 0;
        w->buf[byte] |= (uint8_t)(piece << (room - take));
        w->bits += take;
        left -= take;
    }
}

void knzo_bw_bytes(knzo_bw* w, const uint8_t* p, uint64_t nbits)
{
    if ((w->bits + nbits + 7) / 8 > w->cap) { w->overflow = 1; return; }
    uint64_t full = nbits >> 3;
    if ((w->bits & 7) == 0) {
        memcpy(w->buf + (w->bits >> 3), p, (size_t)full);
        w->bits += full * 8;
    } else {
        for (uint64_t i = 0; i < full; i++) knzo_bw_bits(w, p[i], 8);
    }
    unsigned rem = (unsigned)(nbits & 7);
    if (rem) knzo_bw_bits(w, (uint64_t)(p[full] >> (8 - rem)), rem);
}

void knzo_br_init(knzo_br* r, const uint8_t* buf, uint64_t nbits)
{
    r->buf = buf; r->nbits = nbits; r->pos = 0; r->error = 0;
}

uint64_t knzo_br_bits(knzo_br* r, unsigned n)
{
    if (n == 0 || n > 64) return 0;
    if (r->pos + n > r->nbits) { r->error = 1; r->pos = r->nbits; return 0; }
    uint64_t v = 0;
    unsigned left = n;
    while (left > 0) {
        size_t byte = (size_t)(r->pos >> 3);
        unsigned used = (unsigned)(r->pos & 7);
        unsigned room = 8 - used;
        unsigned take = left < room ? left : room;
        unsigned piece = (r->buf[byte] >> (room - take)) & ((1u << take) - 1u);
        v = (v << take) | piece;
        r->pos += take;
        left -= take;
    }
    return v;
}

void knzo_br_bytes(knzo_br* r, uint8_t* p, uint64_t nbits)
{
    if (r->pos + nbits > r->nbits) { r->error = 1; r->pos = r->nbits; return; }
    uint64_t full = nbits >> 3;
    if ((r->pos & 7) == 0) {
        memcpy(p, r->buf + (r->pos >> 3), (size_t)full);
        r->pos += full * 8;
    } else {
        for (uint64_t i = 0; i < full; i++) p[i] = (uint8_t)knzo_br_bits(r, 8);
    }
    unsigned rem = (unsigned)(nbits & 7);
    /* DefaultInputBitStream.cpp readBits(byte[],n): trailing bits land in the top of the last byte */
    if (rem) p[full] = (uint8_t)(knzo_br_bits(r, rem) << (8 - rem));
}

/* entropy/EntropyUtils.cpp:247-259 */
void knzo_write_varint(knzo_bw* w, uint32_t value)
{
    while (value >= 128) {
        knzo_bw_bits(w, 0x80 | (value & 0x7F), 8);
        value >>= 7;
    }
    knzo_bw_bits(w, value, 8);
}

/* entropy/EntropyUtils.cpp:261-285 ; malformed => r->error */
uint32_t knzo_read_varint(knzo_br* r)
{
    uint32_t value = (uint32_t)knzo_br_bits(r, 8);
    uint32_t res = value & 0x7F;
    for (int shift = 7; value >= 128; shift += 7) {
        value = (uint32_t)knzo_br_bits(r, 8);
        if (shift == 28) {
            if (value >= 128 || (value & 0x70) != 0) { r->error = 1; return 0; }
            res |= (value & 0x0F) << shift;
            return res;
        }
        res |= (value & 0x7F) << shift;
    }
    return res;
}

/* entropy/EntropyUtils.cpp:57-89 (length fixed at 256) */
int knzo_encode_alphabet(knzo_bw* w, const uint32_t* alphabet, int count)
{
    if (count > 256) return -1;
    if (count == 0) {
        knzo_bw_bits(w, 0, 1); /* FULL_ALPHABET */
        knzo_bw_bits(w, 1, 1); /* ALPHABET_0 */
    } else if (count == 256) {
        knzo_bw_bits(w, 0, 1);
        knzo_bw_bits(w, 0, 1); /* ALPHABET_256 */
    } else {
        uint8_t masks[32];
        memset(masks, 0, sizeof(masks));
        knzo_bw_bits(w, 1, 1); /* PARTIAL_ALPHABET */
        for (int i = 0; i < count; i++)
            masks[alphabet[i] >> 3] |= (uint8_t)(1u << (alphabet[i] & 7));
        const int lastMask = (int)(alphabet[count - 1] >> 3);
        knzo_bw_bits(w, (uint64_t)lastMask, 5);
        knzo_bw_bytes(w, masks, 8u * (uint64_t)(lastMask + 1));
    }
    return count;
}

/* entropy/EntropyUtils.cpp:91-123 */
int knzo_decode_alphabet(knzo_br* r, uint32_t* alphabet)
{
    if (knzo_br_bits(r, 1) == 0) {
        const int size = (knzo_br_bits(r, 1) == 0) ? 256 : 0;
        for (int i = 0; i < size; i++) alphabet[i] = (uint32_t)i;
        return size;
    }
    const int lastMask = (int)knzo_br_bits(r, 5);
    uint8_t masks[32];
    memset(masks, 0, sizeof(masks));
    knzo_br_bytes(r, masks, 8u * (uint64_t)(lastMask + 1));
    int count = 0;
    for (int i = 0; i <= lastMask; i++)
        for (int j = 0; j < 8; j++)
            if ((masks[i] >> j) & 1) alphabet[count++] = (uint32_t)(8 * i + j);
    return count;
}

static __thread int g_bs_version = 6;
void knzo_set_bs_version(int v) { g_bs_version = (v >= 0 && v <= 6) ? v : 6; }
int knzo_get_bs_version(void) { return g_bs_version; }
