/* TEST INFRASTRUCTURE ONLY (see knz_oracle.h).
 * Stream framing restated with jobs=1 semantics (output is independent of the job count):
 *   header        io/CompressedOutputStream.cpp:277-342, io/CompressedInputStream.cpp:511-663
 *   block encode  io/CompressedOutputStream.cpp:651-898 (EncodingTask::run)
 *   block decode  io/CompressedInputStream.cpp:790-1041 (DecodingTask::run)
 *   sequencing    transform/TransformSequence.hpp:88-162 (forward), :165-247 (inverse), :250-265
 *   buffers       io/CompressedOutputStream.cpp:140-145,447-473,720-739 ; CompressedInputStream.cpp:273-282
 *   type ids      transform/TransformFactory.hpp:49-73,100-137 ; entropy/EntropyEncoderFactory.hpp:37-52
 *   checksums     util/XXHash.hpp:61-115,153-230 (kanzi's XXHash64 merge step uses 32-bit style shifts)
 */
#include "knz_oracle.h"
#include <ctype.h>
#include <stdlib.h>
#include <string.h>

#define KNZ_MAGIC 0x4B414E5Au
#define KNZ_VERSION 6
#define DEFAULT_BUFFER_SIZE (256 * 1024)
/* src/Error.hpp:26-48 */
#define ERR_BLOCK_SIZE 2
#define ERR_INVALID_CODEC 3
#define ERR_READ_FILE 11
#define ERR_WRITE_FILE 12
#define ERR_PROCESS_BLOCK 13
#define ERR_INVALID_FILE 15
#define ERR_STREAM_VERSION 16
#define ERR_INVALID_PARAM 18
#define ERR_CRC_CHECK 19

static int ilog2_64(uint64_t x) { return 63 ^ __builtin_clzll(x); }
static int ilog2(uint32_t x) { return 31 ^ __builtin_clz(x); }

static uint32_t rd32le(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64le(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

uint32_t knzo_xxhash32(const uint8_t* data, size_t len, uint32_t seed)
{
    const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    const int length = (int)len;
    uint32_t h32;
    int idx = 0;
    if (length >= 16) {
        const int end16 = length - 16;
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        do {
            v1 += rd32le(&data[idx]) * P2;      v1 = ((v1 << 13) | (v1 >> 19)) * P1;
            v2 += rd32le(&data[idx + 4]) * P2;  v2 = ((v2 << 13) | (v2 >> 19)) * P1;
            v3 += rd32le(&data[idx + 8]) * P2;  v3 = ((v3 << 13) | (v3 >> 19)) * P1;
            v4 += rd32le(&data[idx + 12]) * P2; v4 = ((v4 << 13) | (v4 >> 19)) * P1;
            idx += 16;
        } while (idx <= end16);
        h32 = ((v1 << 1) | (v1 >> 31)) + ((v2 << 7) | (v2 >> 25)) + ((v3 << 12) | (v3 >> 20)) + ((v4 << 18) | (v4 >> 14));
    } else {
        h32 = seed + P5;
    }
    h32 += (uint32_t)length;
    while (idx <= length - 4) {
        h32 += rd32le(&data[idx]) * P3;
        h32 = ((h32 << 17) | (h32 >> 15)) * P4;
        idx += 4;
    }
    while (idx < length) {
        h32 += (uint32_t)data[idx] * P5;
        h32 = ((h32 << 11) | (h32 >> 21)) * P1;
        idx++;
    }
    h32 ^= h32 >> 15; h32 *= P2; h32 ^= h32 >> 13; h32 *= P3;
    return h32 ^ (h32 >> 16);
}

static uint64_t xx64_round(uint64_t acc, uint64_t val)
{
    acc += val * 0xC2B2AE3D27D4EB4Full;
    return ((acc << 31) | (acc >> 33)) * 0x9E3779B185EBCA87ull;
}

uint64_t knzo_xxhash64(const uint8_t* data, size_t len, uint64_t seed)
{
    const uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull,
                   P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
    const int length = (int)len;
    uint64_t h64;
    int idx = 0;
    if (length >= 32) {
        const int length32 = length - 32;
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        do {
            v1 = xx64_round(v1, rd64le(&data[idx]));
            v2 = xx64_round(v2, rd64le(&data[idx + 8]));
            v3 = xx64_round(v3, rd64le(&data[idx + 16]));
            v4 = xx64_round(v4, rd64le(&data[idx + 24]));
            idx += 32;
        } while (idx <= length32);
        /* XXHash.hpp:186-187: shifts as written in the reference (not 64-bit rotations) */
        h64 = ((v1 << 1) | (v1 >> 31)) + ((v2 << 7) | (v2 >> 25)) + ((v3 << 12) | (v3 >> 20)) + ((v4 << 18) | (v4 >> 14));
        h64 = (h64 ^ xx64_round(0, v1)) * P1 + P4;
        h64 = (h64 ^ xx64_round(0, v2)) * P1 + P4;
        h64 = (h64 ^ xx64_round(0, v3)) * P1 + P4;
        h64 = (h64 ^ xx64_round(0, v4)) * P1 + P4;
    } else {
        h64 = seed + P5;
    }
    h64 += (uint64_t)(int64_t)length;
    while (idx + 8 <= length) {
        h64 ^= xx64_round(0, rd64le(&data[idx]));
        h64 = ((h64 << 27) | (h64 >> 37)) * P1 + P4;
        idx += 8;
    }
    while (idx + 4 <= length) {
        h64 ^= (uint64_t)rd32le(&data[idx]) * P1;
        h64 = ((h64 << 23) | (h64 >> 41)) * P2 + P3;
        idx += 4;
    }
    while (idx < length) {
        h64 ^= (uint64_t)data[idx] * P5;
        h64 = ((h64 << 11) | (h64 >> 53)) * P1;
        idx++;
    }
    h64 ^= h64 >> 33; h64 *= P2; h64 ^= h64 >> 29; h64 *= P3;
    return h64 ^ (h64 >> 32);
}

static int token_type(const char* s, size_t len)
{
    static const struct { const char* n; int t; } T[] = {
        {"NONE", 0}, {"BWT", 1}, {"BWTS", 2}, {"LZ", 3}, {"RLT", 5}, {"ZRLT", 6}, {"MTFT", 7}, {"RANK", 8},
        {"EXE", 9}, {"TEXT", 10}, {"ROLZ", 11}, {"ROLZX", 12}, {"SRT", 13}, {"LZP", 14}, {"MM", 15},
        {"LZX", 16}, {"UTF", 17}, {"PACK", 18}, {"DNA", 19} };
    char up[16];
    if (len >= sizeof(up)) return -1;
    for (size_t i = 0; i < len; i++) up[i] = (char)toupper((unsigned char)s[i]);
    up[len] = 0;
    for (size_t i = 0; i < sizeof(T) / sizeof(T[0]); i++)
        if (strcmp(T[i].n, up) == 0) return T[i].t;
    return -1;
}

/* TransformFactory.hpp:100-137. Returns ~0 on error. */
uint64_t knzo_transform_type(const char* names)
{
    uint64_t res = 0;
    int shift = 42, n = 0;
    const char* p = names;
    if (strchr(names, '+') == NULL) {
        const int t = token_type(names, strlen(names));
        return t < 0 ? ~0ull : ((uint64_t)t << 42);
    }
    while (1) {
        const char* q = strchr(p, '+');
        const size_t len = q ? (size_t)(q - p) : strlen(p);
        if (++n > 8) return ~0ull;
        const int t = token_type(p, len);
        if (t < 0) return ~0ull;
        if (t != 0) { res |= ((uint64_t)t << shift); shift -= 6; }
        if (!q) break;
        p = q + 1;
    }
    return res;
}

int knzo_entropy_type(const char* name)
{
    static const struct { const char* n; int t; } T[] = {
        {"NONE", 0}, {"HUFFMAN", 1}, {"FPAQ", 2}, {"RANGE", 4}, {"ANS0", 5}, {"CM", 6}, {"TPAQ", 7}, {"ANS1", 8}, {"TPAQX", 9} };
    char up[16];
    const size_t len = strlen(name);
    if (len >= sizeof(up)) return -1;
    for (size_t i = 0; i < len; i++) up[i] = (char)toupper((unsigned char)name[i]);
    up[len] = 0;
    for (size_t i = 0; i < sizeof(T) / sizeof(T[0]); i++)
        if (strcmp(T[i].n, up) == 0) return T[i].t;
    return -1;
}

/* Sequence of transform ids as TransformFactory::newTransform builds it (:208-222). */
static int seq_tokens(uint64_t ttype, int* tok)
{
    int nb = 0;
    for (int i = 0; i < 8; i++) {
        const int t = (int)((ttype >> (42 - 6 * i)) & 63);
        if (t != 0 || i == 0) tok[nb++] = t;
    }
    return nb;
}

static int max_encoded_len(int t, int n)
{
    switch (t) {
    case 1: return n + 33;
    case 13: return n + 1024;
    case 5: return (n <= 512) ? n + 32 : n;
    case 3: case 16: return knzo_lz_max_encoded(n);
    case 17: return n + 8192;          /* UTFCodec.hpp:54 (only its share of a chain's buffer size: the stage itself is not restated here) */
    default: return n;
    }
}

static int seq_required(const int* tok, int nb, int n)
{
    int req = n;
    for (int i = 0; i < nb; i++) {
        const int nx = max_encoded_len(tok[i], req);
        if (nx > req) req = nx;
    }
    return req;
}

static int supported_transform(int t) { return t == 0 || t == 1 || t == 3 || t == 5 || t == 6 || t == 7 || t == 8 || t == 13 || t == 16; }

/* Stages a caller has run itself on a block before handing it over (the product's TEXT and UTF, ids 10 and 17, which this restatement
 * does not contain): the stand-in for knz_hip_encode_block_hosted / knz_hip_decode_block_hosted in tests/stub. While `hosted` is
 * non-zero the first `hosted` tokens of a chain are taken as done (encode: bit i of `applied` says whether stage i succeeded, the
 * checksum and the original length come from the caller) or left to the caller (decode: the block's skip flags and stored checksum
 * are handed back, nothing is verified). */
static __thread struct { int hosted; unsigned applied; int origLen; uint64_t checksum; int skipOut; uint64_t checksumOut; } g_hosted;
void knzo_set_hosted(int hosted, unsigned applied, int origLen, uint64_t checksum) { g_hosted.hosted = hosted; g_hosted.applied = applied; g_hosted.origLen = origLen; g_hosted.checksum = checksum; }
void knzo_get_hosted(int* skipFlags, uint64_t* checksum) { *skipFlags = g_hosted.skipOut; *checksum = g_hosted.checksumOut; }
static int supported_entropy(int e) { return e == 0 || e == 1 || e == 2 || e == 5 || e == 8; }

/* TransformSequence::forward with explicit capacities. data = block input (capacity dataCap),
 * result written to 'outbuf' (capacity bufCap). Returns post-transform length. */
static int seq_forward(const uint8_t* in, int count, const int* tok, int nb, int etype,
                       int dataCap, int bufCap, uint8_t* outbuf, int* skipFlagsOut)
{
    const int hosted = g_hosted.hosted;
    const int blockSize = hosted ? g_hosted.origLen : count;
    const int requiredSize = seq_required(tok, nb, blockSize);
    int skip = 0xFF;
    /* physical buffers: 0 = input(data), 1 = output(buffer), 2 = temp */
    int capA = dataCap, capB = bufCap;
    uint8_t* A = (uint8_t*)malloc((size_t)(capA > requiredSize ? capA : requiredSize) + 16);
    uint8_t* B = (uint8_t*)malloc((size_t)(capB > requiredSize ? capB : requiredSize) + 16);
    memcpy(A, in, (size_t)count);
    uint8_t* pin = A; uint8_t* pout = B;
    int capIn = capA, capOut = capB;
    int swaps = 0;
    for (int i = 0; i < hosted && i < nb; i++) {
        /* a stage the caller has run: a swap of the buffers and a cleared flag when it succeeded */
        if (capOut < requiredSize) capOut = requiredSize;
        if (!((g_hosted.applied >> i) & 1u)) continue;
        skip &= ~(1 << (7 - i));
        { uint8_t* tp = pin; pin = pout; pout = tp; const int tc = capIn; capIn = capOut; capOut = tc; }
        swaps++;
    }
    if (swaps & 1) memcpy(pin, in, (size_t)count);          /* (the data sits where the last host stage left it) */
    for (int i = hosted; i < nb; i++) {
        if (capOut < requiredSize) capOut = requiredSize;   /* reallocation path, :104-115 */
        int outLen = 0;
        if (!knzo_transform_forward(tok[i], pin, count, pout, capOut, etype, &outLen)) continue;
        skip &= ~(1 << (7 - i));
        count = outLen;
        { uint8_t* tp = pin; pin = pout; pout = tp; const int tc = capIn; capIn = capOut; capOut = tc; }
        swaps++;
    }
    if ((swaps & 1) == 0) {
        if (count > bufCap || count > capIn) skip = 0xFF;
        else memcpy(outbuf, pin, (size_t)count);
    } else {
        memcpy(outbuf, pin, (size_t)count);   /* pin is the 'output' physical buffer after an odd swap count */
    }
    free(A); free(B);
    *skipFlagsOut = skip;
    return count;
}

int64_t knzo_encode_block(const uint8_t* in, int n, uint64_t ttype, int etype, int checksumBits,
                          int dataCap, int bufCap, uint8_t* out, size_t cap, int* skipFlagsOut, int* postLenOut)
{
    int mode = 0;
    uint64_t checksum = 0;
    if (g_hosted.hosted && checksumBits) checksum = g_hosted.checksum;
    else if (checksumBits == 32) checksum = knzo_xxhash32(in, (size_t)n, KNZ_MAGIC);
    else if (checksumBits == 64) checksum = knzo_xxhash64(in, (size_t)n, KNZ_MAGIC);
    if (n <= 15) { ttype = 0; etype = 0; mode |= 0x80; }
    int tok[8];
    const int nb = seq_tokens(ttype, tok);
    for (int i = 0; i < nb; i++) if (!(i < g_hosted.hosted && n > 15) && !supported_transform(tok[i])) return -2;
    if (!supported_entropy(etype)) return -2;
    const int requiredSize = seq_required(tok, nb, (g_hosted.hosted && n > 15) ? g_hosted.origLen : n);
    if (bufCap < requiredSize) bufCap = requiredSize;
    if (dataCap < n) dataCap = n;
    uint8_t* buffer = (uint8_t*)malloc((size_t)bufCap + 16);
    int skipFlags = 0xFF;
    const int postLen = seq_forward(in, n, tok, nb, etype, dataCap, bufCap, buffer, &skipFlags);
    const int dataSize = (postLen < 256) ? 1 : (ilog2((uint32_t)postLen) >> 3) + 1;
    if (dataSize > 4) { free(buffer); return -1; }
    mode |= ((dataSize - 1) & 3) << 5;
    knzo_bw w;
    knzo_bw_init(&w, out, cap);
    if ((mode & 0x80) || nb <= 4) {
        mode |= (skipFlags >> 4);
        knzo_bw_bits(&w, (uint64_t)mode, 8);
    } else {
        mode |= 0x10;
        knzo_bw_bits(&w, (uint64_t)mode, 8);
        knzo_bw_bits(&w, (uint64_t)skipFlags, 8);
    }
    knzo_bw_bits(&w, (uint64_t)postLen, 8u * (unsigned)dataSize);
    if (checksumBits == 32) knzo_bw_bits(&w, checksum & 0xFFFFFFFFull, 32);
    else if (checksumBits == 64) knzo_bw_bits(&w, checksum, 64);
    int r;
    switch (etype) {
    case 0: r = knzo_none_encode_bw(&w, buffer, (uint32_t)postLen); break;
    case 1: r = knzo_huffman_encode_bw(&w, buffer, (uint32_t)postLen); break;
    case 2: r = knzo_fpaq_encode_bw(&w, buffer, (uint32_t)postLen); break;
    case 5: r = knzo_ans_encode_bw(&w, buffer, (uint32_t)postLen, 0); break;
    default: r = knzo_ans_encode_bw(&w, buffer, (uint32_t)postLen, 1); break;
    }
    free(buffer);
    if (r != postLen || w.overflow) return -1;
    if (skipFlagsOut) *skipFlagsOut = skipFlags;
    if (postLenOut) *postLenOut = postLen;
    return (int64_t)w.bits;
}

/* DecodingTask::run for one block's private bits. Returns 0 or an Error code. */
int knzo_decode_block(const uint8_t* in, uint64_t nbits, uint64_t ttype, int etype, int checksumBits,
                      int blockSize, uint8_t* out, int outCap, int* outLen)
{
    *outLen = 0;
    knzo_br r;
    knzo_br_init(&r, in, nbits);
    const int mode = (int)knzo_br_bits(&r, 8);
    int skipFlags = 0;
    if (mode & 0x80) { ttype = 0; etype = 0; }
    else if (mode & 0x10) skipFlags = (int)knzo_br_bits(&r, 8);
    else skipFlags = ((mode << 4) | 0x0F) & 0xFF;
    const int dataSize = 1 + ((mode >> 5) & 3);
    const int pre = (int)knzo_br_bits(&r, 8u * (unsigned)dataSize);
    const uint32_t blkLen = (uint32_t)((blockSize + 512 > blockSize + (blockSize >> 4)) ? blockSize + 512 : blockSize + (blockSize >> 4));
    uint32_t mts = blkLen + blkLen / 2;
    if (mts < 2048) mts = 2048;
    if (mts > (1u << 30)) mts = 1u << 30;
    if (r.error || pre <= 0 || (uint32_t)pre > mts) return ERR_READ_FILE;
    uint64_t checksum1 = 0;
    if (checksumBits == 32) checksum1 = knzo_br_bits(&r, 32);
    else if (checksumBits == 64) checksum1 = knzo_br_bits(&r, 64);
    int tok[8];
    const int nb = seq_tokens(ttype, tok);
    const int hosted = (mode & 0x80) ? 0 : g_hosted.hosted;
    for (int i = hosted; i < nb; i++) if (!supported_transform(tok[i])) return ERR_INVALID_CODEC;
    if (!supported_entropy(etype)) return ERR_INVALID_CODEC;
    g_hosted.skipOut = (mode & 0x80) ? 0xFF : skipFlags;
    g_hosted.checksumOut = checksum1;
    const int rbytes = (int)((nbits + 7) >> 3);
    int dataCap = (int)blkLen > rbytes ? (int)blkLen : rbytes;
    int bufCap = (int)blkLen > pre + 512 ? (int)blkLen : pre + 512;
    const int big = dataCap > bufCap ? dataCap : bufCap;
    uint8_t* A = (uint8_t*)malloc((size_t)big + 16);
    uint8_t* B = (uint8_t*)malloc((size_t)big + 16);
    int d;
    switch (etype) {
    case 0: d = knzo_none_decode_br(&r, A, (uint32_t)pre); break;
    case 1: d = knzo_huffman_decode_br(&r, A, (uint32_t)pre); break;
    case 2: d = knzo_fpaq_decode_br(&r, A, (uint32_t)pre); break;
    case 5: d = knzo_ans_decode_br(&r, A, (uint32_t)pre, 0); break;
    default: d = knzo_ans_decode_br(&r, A, (uint32_t)pre, 1); break;
    }
    if (d != pre) { free(A); free(B); return ERR_PROCESS_BLOCK; }
    /* TransformSequence::inverse :165-247 */
    int count = pre;
    int res = 1;
    uint8_t* pin = A; uint8_t* pout = B;
    if (count > dataCap) res = 0;
    if (res && skipFlags != 0xFF) {
        for (int i = nb - 1; i >= hosted; i--) {
            if (skipFlags & (1 << (7 - i))) continue;
            int ol = 0;
            res = knzo_transform_inverse(tok[i], pin, count, pout, dataCap, &ol);
            if (!res) break;
            count = ol;
            { uint8_t* tp = pin; pin = pout; pout = tp; }
        }
    }
    if (!res) { free(A); free(B); return ERR_PROCESS_BLOCK; }
    if (count > outCap) { free(A); free(B); return ERR_PROCESS_BLOCK; }
    memcpy(out, pin, (size_t)count);
    free(A); free(B);
    if (g_hosted.hosted) { *outLen = count; return 0; }      /* (the caller undoes its stages and verifies the checksum) */
    if (checksumBits == 32) {
        if (knzo_xxhash32(out, (size_t)count, KNZ_MAGIC) != (uint32_t)checksum1) return ERR_CRC_CHECK;
    } else if (checksumBits == 64) {
        if (knzo_xxhash64(out, (size_t)count, KNZ_MAGIC) != checksum1) return ERR_CRC_CHECK;
    }
    *outLen = count;
    return 0;
}

/* CompressedInputStream.cpp:622-645: versions below 6 seed with the version alone, leave the checksum size out and keep 16 bits */
static uint32_t header_checksum_v(int ver, uint32_t ckSize, uint32_t etype, uint64_t ttype, uint32_t blockSize, int szMask, uint64_t size)
{
    const uint32_t HASH = 0x1E35A7BDu;
    uint32_t c = HASH * ((ver >= 6 ? 0x01030507u : 1u) * (uint32_t)ver);
    if (ver >= 6) c ^= HASH * (uint32_t)~ckSize;
    c ^= HASH * (uint32_t)~etype;
    c ^= HASH * (uint32_t)((~ttype) >> 32);
    c ^= HASH * (uint32_t)~ttype;
    c ^= HASH * (uint32_t)~blockSize;
    if (szMask != 0) {
        c ^= HASH * (uint32_t)((~size) >> 32);
        c ^= HASH * (uint32_t)~size;
    }
    return ((c >> 23) ^ (c >> 3)) & (ver >= 6 ? 0xFFFFFFu : 0xFFFFu);
}

static uint32_t header_checksum(uint32_t ckSize, uint32_t etype, uint64_t ttype, uint32_t blockSize, int szMask, uint64_t size)
{
    const uint32_t HASH = 0x1E35A7BDu;
    uint32_t c = HASH * (0x01030507u * KNZ_VERSION);
    c ^= HASH * (uint32_t)~ckSize;
    c ^= HASH * (uint32_t)~etype;
    c ^= HASH * (uint32_t)((~ttype) >> 32);
    c ^= HASH * (uint32_t)~ttype;
    c ^= HASH * (uint32_t)~blockSize;
    if (szMask != 0) {
        c ^= HASH * (uint32_t)((~size) >> 32);
        c ^= HASH * (uint32_t)~size;
    }
    return ((c >> 23) ^ (c >> 3)) & 0xFFFFFFu;
}

int knzo_compress(const uint8_t* in, size_t n, const char* transform, const char* entropy,
                  int blockSize, int checksum, uint64_t origSize, int headerless,
                  uint8_t* out, size_t cap, size_t* outLen)
{
    return knzo_compress_jobs(in, n, transform, entropy, blockSize, checksum, origSize, headerless, 1, out, cap, outLen);
}

/* `jobs` only selects which buffer slot (and therefore which capacities) a block sees:
 * block i runs on slot i % jobs (io/CompressedOutputStream.cpp:447-473); SURVEY.md App. C #1. */
int knzo_compress_jobs(const uint8_t* in, size_t n, const char* transform, const char* entropy,
                       int blockSize, int checksum, uint64_t origSize, int headerless, int jobs,
                       uint8_t* out, size_t cap, size_t* outLen)
{
    uint64_t bits = 0;
    return knzo_compress_run(in, n, transform, entropy, blockSize, checksum, origSize, headerless, jobs, 0, 1, out, cap, outLen, &bits);
}

/* The blocks of a run (length prefixes + private streams, optional end marker) appended to an open bit writer; codecs by id. */
int knzo_compress_run_ids(const uint8_t* in, size_t n, uint64_t ttype, int etype, int blockSize, int checksum, int jobs,
                          uint64_t firstBlock, int finish, knzo_bw* w)
{
    if (jobs < 1 || jobs > 64) return ERR_INVALID_PARAM;
    int dataCaps[64], bufCaps[64];
    for (int j = 0; j < jobs; j++) {
        if (j == 0) {
            dataCaps[j] = blockSize + (blockSize >> 3);
            if (dataCaps[j] < DEFAULT_BUFFER_SIZE) dataCaps[j] = DEFAULT_BUFFER_SIZE;
        } else {
            dataCaps[j] = blockSize + (blockSize >> 6);
            if (dataCaps[j] < 65536) dataCaps[j] = 65536;
        }
        bufCaps[j] = 0;
    }
    size_t blockIdx = (size_t)firstBlock;
    if (firstBlock > 0) {
        /* slots already used by earlier (full-size) blocks have grown their buffers */
        int tok0[8];
        const int nb0 = seq_tokens(ttype, tok0);
        const int req0 = seq_required(tok0, nb0, blockSize);
        for (int j = 0; j < jobs; j++) if ((uint64_t)j < firstBlock) bufCaps[j] = req0;
    }
    const size_t tmpCap = (size_t)blockSize + ((size_t)blockSize >> 1) + 65536;
    uint8_t* tmp = (uint8_t*)malloc(tmpCap);
    size_t off = 0;
    while (off < n) {
        const int len = (n - off < (size_t)blockSize) ? (int)(n - off) : blockSize;
        const int slot = (int)(blockIdx % (size_t)jobs);
        int dataCap = dataCaps[slot], bufCap = bufCaps[slot];
        blockIdx++;
        int tok[8];
        const int nb = seq_tokens(len <= 15 ? 0 : ttype, tok);
        const int req = seq_required(tok, nb, (g_hosted.hosted && len > 15) ? g_hosted.origLen : len);
        if (bufCap < req) bufCap = req;
        int skipFlags, postLen;
        const int64_t bits = knzo_encode_block(in + off, len, ttype, etype, checksum, dataCap, bufCap, tmp, tmpCap, &skipFlags, &postLen);
        if (bits < 0) { free(tmp); return ERR_PROCESS_BLOCK; }
        /* _data may grow after the transform: CompressedOutputStream.cpp:774-783 */
        {
            int bs2 = len + (len >> 3);
            if (bs2 < postLen) bs2 = postLen;
            if (bs2 < DEFAULT_BUFFER_SIZE) bs2 = DEFAULT_BUFFER_SIZE;
            if (dataCap < bs2) dataCap = bs2;
        }
        dataCaps[slot] = dataCap; bufCaps[slot] = bufCap;
        const uint64_t written = (uint64_t)bits;
        const unsigned lw = (written < 8) ? 3u : (unsigned)ilog2((uint32_t)(written >> 3)) + 4u;
        knzo_bw_bits(w, lw - 3, 5);
        knzo_bw_bits(w, written, lw);
        knzo_bw_bytes(w, tmp, written);
        off += (size_t)len;
    }
    free(tmp);
    if (finish) {
        knzo_bw_bits(w, 0, 5);
        knzo_bw_bits(w, 0, 3);
    }
    return 0;
}

/* A run of consecutive blocks of a larger stream: firstBlock = index of the first block (selects the
 * buffer slots), finish = append the end marker. *outBits = exact bit count (multi-GPU sharding tests). */
int knzo_compress_run(const uint8_t* in, size_t n, const char* transform, const char* entropy,
                      int blockSize, int checksum, uint64_t origSize, int headerless, int jobs,
                      uint64_t firstBlock, int finish, uint8_t* out, size_t cap, size_t* outLen, uint64_t* outBits)
{
    *outLen = 0;
    const uint64_t ttype = knzo_transform_type(transform);
    const int etype = knzo_entropy_type(entropy);
    if (ttype == ~0ull || etype < 0) return ERR_INVALID_PARAM;
    if (blockSize < 1024 || blockSize > (1 << 30) || (blockSize & -16) != blockSize) return ERR_INVALID_PARAM;
    if (checksum != 0 && checksum != 32 && checksum != 64) return ERR_INVALID_PARAM;
    knzo_bw w;
    knzo_bw_init(&w, out, cap);
    const int ver = knzo_get_bs_version();
    if (!headerless && ver < 6) {
        /* the header as versions below 6 had it (what CompressedInputStream.cpp:541-558,606-645 reads): one checksum bit, no padding,
         * 16 checksum bits. Test writer: the reference writes version 6 only. */
        if (checksum == 64) return ERR_INVALID_PARAM;
        knzo_bw_bits(&w, KNZ_MAGIC, 32);
        knzo_bw_bits(&w, (uint64_t)ver, 4);
        knzo_bw_bits(&w, checksum == 32 ? 1 : 0, 1);
        knzo_bw_bits(&w, (uint64_t)etype, 5);
        knzo_bw_bits(&w, ttype, 48);
        knzo_bw_bits(&w, (uint64_t)(blockSize >> 4), 28);
        const int szMask = (origSize == 0 || origSize >= (1ull << 48)) ? 0 : (ilog2_64(origSize) >> 4) + 1;
        knzo_bw_bits(&w, (uint64_t)szMask, 2);
        if (szMask) knzo_bw_bits(&w, origSize, 16u * (unsigned)szMask);
        knzo_bw_bits(&w, header_checksum_v(ver, 0, (uint32_t)etype, ttype, (uint32_t)blockSize, szMask, origSize), 16);
    } else if (!headerless) {
        const uint32_t ckSize = checksum == 32 ? 1 : (checksum == 64 ? 2 : 0);
        knzo_bw_bits(&w, KNZ_MAGIC, 32);
        knzo_bw_bits(&w, KNZ_VERSION, 4);
        knzo_bw_bits(&w, ckSize, 2);
        knzo_bw_bits(&w, (uint64_t)etype, 5);
        knzo_bw_bits(&w, ttype, 48);
        knzo_bw_bits(&w, (uint64_t)(blockSize >> 4), 28);
        const int szMask = (origSize == 0 || origSize >= (1ull << 48)) ? 0 : (ilog2_64(origSize) >> 4) + 1;
        knzo_bw_bits(&w, (uint64_t)szMask, 2);
        if (szMask) knzo_bw_bits(&w, origSize, 16u * (unsigned)szMask);
        knzo_bw_bits(&w, 0, 15);
        knzo_bw_bits(&w, header_checksum(ckSize, (uint32_t)etype, ttype, (uint32_t)blockSize, szMask, origSize), 24);
    }
    {
        const int rc = knzo_compress_run_ids(in, n, ttype, etype, blockSize, checksum, jobs, firstBlock, finish, &w);
        if (rc) return rc;
    }
    if (w.overflow) return ERR_WRITE_FILE;
    *outLen = (size_t)((w.bits + 7) >> 3);
    *outBits = w.bits;
    return 0;
}

/* Blocks from bit `startBit` on: until the end marker, `maxBlocks` blocks (< 0: no limit) or the end of the data. */
int knzo_decode_run(const uint8_t* in, uint64_t inBits, uint64_t startBit, uint64_t ttype, int etype, int checksumBits, int blockSize,
                    int64_t maxBlocks, uint8_t* out, size_t cap, size_t* outLen, uint64_t* endBit, int64_t* blocksDone)
{
    knzo_br r;
    knzo_br_init(&r, in, inBits);
    r.pos = startBit;
    uint8_t* tmp = NULL;
    size_t tmpCap = 0;
    size_t off = 0;
    int err = 0;
    int64_t done = 0;
    *endBit = startBit;
    while (maxBlocks < 0 || done < maxBlocks) {
        /* a run (maxBlocks >= 0) may end with the data; a whole stream must end with the end marker, like the reference, which
         * throws at the end of the input (io/CompressedInputStream.cpp:823-856) */
        if (r.pos + 8 > inBits) { if (maxBlocks < 0) err = ERR_READ_FILE; break; }
        const unsigned lr = 3 + (unsigned)knzo_br_bits(&r, 5);
        const uint64_t bits = knzo_br_bits(&r, lr);
        if (r.error) { err = ERR_READ_FILE; break; }
        if (bits == 0) { *endBit = r.pos; break; }
        if (bits > (1ull << 34)) { err = ERR_BLOCK_SIZE; break; }
        const size_t nb = (size_t)((bits + 7) >> 3);
        if (tmpCap < nb + 8) { free(tmp); tmpCap = nb + 8; tmp = (uint8_t*)malloc(tmpCap); }
        memset(tmp, 0, nb + 8);
        knzo_br_bytes(&r, tmp, bits);
        if (r.error) { err = ERR_READ_FILE; break; }
        int ol = 0;
        const size_t room = cap - off;
        const int outCap = room > (size_t)0x7FFFFFFF ? 0x7FFFFFFF : (int)room;
        err = knzo_decode_block(tmp, bits, ttype, etype, checksumBits, blockSize, out + off, outCap, &ol);
        if (err) break;
        off += (size_t)ol;
        done++;
        *endBit = r.pos;
    }
    free(tmp);
    *outLen = off;
    *blocksDone = done;
    return err;
}

int knzo_decompress(const uint8_t* in, size_t inLen, uint8_t* out, size_t cap, size_t* outLen)
{
    *outLen = 0;
    knzo_br r;
    knzo_br_init(&r, in, 8ull * inLen);
    if ((uint32_t)knzo_br_bits(&r, 32) != KNZ_MAGIC) return ERR_INVALID_FILE;
    const int ver = (int)knzo_br_bits(&r, 4);
    if (ver > KNZ_VERSION) return ERR_STREAM_VERSION;
    uint32_t ckSize;
    if (ver >= 6) {
        ckSize = (uint32_t)knzo_br_bits(&r, 2);
        if (ckSize == 3) return ERR_INVALID_FILE;
    } else {
        ckSize = (uint32_t)knzo_br_bits(&r, 1);         /* CompressedInputStream.cpp:555-558 */
    }
    const int etype = (int)knzo_br_bits(&r, 5);
    const uint64_t ttype = knzo_br_bits(&r, 48);
    const int blockSize = (int)(knzo_br_bits(&r, 28) << 4);
    if (blockSize < 1024 || blockSize > (1 << 30)) return ERR_BLOCK_SIZE;
    const int szMask = (int)knzo_br_bits(&r, 2);
    uint64_t size = 0;
    if (szMask) size = knzo_br_bits(&r, 16u * (unsigned)szMask);
    if (ver >= 6) knzo_br_bits(&r, 15);
    const uint32_t ck1 = (uint32_t)knzo_br_bits(&r, ver >= 6 ? 24 : 16);
    if (r.error) return ERR_INVALID_FILE;
    if (ck1 != header_checksum_v(ver, ckSize, (uint32_t)etype, ttype, (uint32_t)blockSize, szMask, size)) return ERR_CRC_CHECK;
    const int checksumBits = ckSize == 1 ? 32 : (ckSize == 2 ? 64 : 0);
    uint64_t endBit = 0;
    int64_t done = 0;
    const int before = knzo_get_bs_version();
    knzo_set_bs_version(ver);                              /* the codecs' old layouts, where they have one */
    const int rc = knzo_decode_run(in, 8ull * inLen, r.pos, ttype, etype, checksumBits, blockSize, -1, out, cap, outLen, &endBit, &done);
    knzo_set_bs_version(before);
    return rc;
}

