/*
 * knz_oracle.h -- CPU restatement of the kanzi per-block transform + entropy path.
 *
 * TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (kanzi-cpp_amd/) must never
 * call it. Every function cites the reference file:line (relative to /root/reference/src)
 * it restates. Parity is PINNED: tests/test_oracle_vs_ref.py checks every function
 * against the unmodified reference compiled into oracle/_ref/ (see oracle/Makefile) and
 * tests/golden/ holds vectors generated from that reference (tests/golden/make_golden.py).
 */
#ifndef KNZ_ORACLE_H
#define KNZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Bitstream version the calling thread's codec calls work in (6 = current; 3..5 = the old layouts the reference still DECODES:
 * Huffman chunks, HuffmanDecoder.cpp:349-459; BWT block header, BWTBlockCodec.cpp:140-164; stream header,
 * CompressedInputStream.cpp:541-558,623-625). The writers follow it too, which is how the tests get old streams: the reference
 * has no writer for them. knzo_decompress switches by the header it reads. */
void knzo_set_bs_version(int v);
int knzo_get_bs_version(void);

/* ---- MSB-first bit I/O (bitstream/DefaultOutputBitStream.hpp:83-131, DefaultInputBitStream.hpp:88-150) */
typedef struct {
    uint8_t* buf;
    size_t cap;      /* bytes */
    uint64_t bits;   /* bits written so far */
    int overflow;
} knzo_bw;

typedef struct {
    const uint8_t* buf;
    uint64_t nbits;  /* bits available */
    uint64_t pos;    /* bits consumed */
    int error;       /* set when reading past the end */
} knzo_br;

void knzo_bw_init(knzo_bw* w, uint8_t* buf, size_t cap);
void knzo_bw_bits(knzo_bw* w, uint64_t v, unsigned n);                 /* n in [0,64], v must be clean */
void knzo_bw_bytes(knzo_bw* w, const uint8_t* p, uint64_t nbits);     /* first nbits of p, MSB-first */
void knzo_br_init(knzo_br* r, const uint8_t* buf, uint64_t nbits);
uint64_t knzo_br_bits(knzo_br* r, unsigned n);
void knzo_br_bytes(knzo_br* r, uint8_t* p, uint64_t nbits);

/* ---- EntropyUtils (entropy/EntropyUtils.cpp:57-89,91-123,131-245,247-285) */
int knzo_encode_alphabet(knzo_bw* w, const uint32_t* alphabet, int count);
int knzo_decode_alphabet(knzo_br* r, uint32_t* alphabet);
int knzo_normalize_freqs(uint32_t* freqs, uint32_t* alphabet, int length, uint32_t total, uint32_t scale);
void knzo_write_varint(knzo_bw* w, uint32_t v);
uint32_t knzo_read_varint(knzo_br* r);

/* ---- Entropy codecs writing into / reading from a bit stream. Return n on success. */
int knzo_ans_encode_bw(knzo_bw* w, const uint8_t* in, uint32_t n, int order);
int knzo_ans_decode_br(knzo_br* r, uint8_t* out, uint32_t n, int order);
int knzo_huffman_encode_bw(knzo_bw* w, const uint8_t* in, uint32_t n);
int knzo_huffman_decode_br(knzo_br* r, uint8_t* out, uint32_t n);
int knzo_fpaq_encode_bw(knzo_bw* w, const uint8_t* in, uint32_t n);
int knzo_fpaq_decode_br(knzo_br* r, uint8_t* out, uint32_t n);
int knzo_none_encode_bw(knzo_bw* w, const uint8_t* in, uint32_t n);
int knzo_none_decode_br(knzo_br* r, uint8_t* out, uint32_t n);

/* Convenience: standalone buffers. etype = kanzi entropy id (0 NONE,1 HUFFMAN,2 FPAQ,5 ANS0,8 ANS1).
 * encode returns bits written (or -1); decode returns the decoder result (n on success). */
int64_t knzo_entropy_encode(int etype, const uint8_t* in, uint32_t n, uint8_t* out, size_t cap);
int knzo_entropy_decode(int etype, const uint8_t* in, size_t inBytes, uint8_t* out, uint32_t n);

/* ---- Transforms. Return 1 on success (0 = "does not apply"/failed). *outLen bytes written.
 * ttype = kanzi transform id (1 BWT(block codec), 3 LZ, 5 RLT, 6 ZRLT, 7 MTFT, 8 RANK, 13 SRT, 16 LZX, 0 NONE;
 * 64 selects SBRT's TIMESTAMP mode, which has no kanzi id).
 * dstCap mirrors SliceArray::_length - _index of the destination. etype is the stream's entropy
 * id (RLT picks its escape from it, transform/RLT.cpp:59-106); pass -1 when "entropy" is absent. */
int knzo_transform_forward(int ttype, const uint8_t* src, int n, uint8_t* dst, int dstCap, int etype, int* outLen);
int knzo_transform_inverse(int ttype, const uint8_t* src, int n, uint8_t* dst, int dstCap, int* outLen);

/* LZ / LZX (transform/LZCodec.cpp:119-456,470-640). extra = 1 for LZX. */
int knzo_lz_forward(const uint8_t* src, int n, uint8_t* dst, int dstCap, int extra, int* outLen);
int knzo_lz_inverse(const uint8_t* src, int n, uint8_t* dst, int dstCap, int* outLen);
int knzo_lz_max_encoded(int n);

/* Raw BWT without the block-codec header (transform/BWT.cpp:92-134). primary[8]. */
int knzo_bwt_forward_raw(const uint8_t* src, int n, uint8_t* dst, int* primary);
int knzo_bwt_inverse_raw(const uint8_t* src, int n, uint8_t* dst, const int* primary);
int knzo_bwt_chunks(int n);

/* ---- Stream level (io/CompressedOutputStream.cpp, io/CompressedInputStream.cpp) */
uint64_t knzo_transform_type(const char* names);   /* TransformFactory.hpp:100-137 */
int knzo_entropy_type(const char* name);           /* EntropyEncoderFactory.hpp:37-52 */

/* Encode one block payload exactly as EncodingTask::run builds its private stream
 * (mode byte, length, optional checksum, entropy bits). Returns bits written or <0. */
int64_t knzo_encode_block(const uint8_t* in, int n, uint64_t ttype, int etype, int checksumBits,
                          int dataCap, int bufCap, uint8_t* out, size_t cap, int* skipFlags, int* postLen);
int knzo_decode_block(const uint8_t* in, uint64_t nbits, uint64_t ttype, int etype, int checksumBits,
                      int blockSize, uint8_t* out, int outCap, int* outLen);

/* Whole stream, jobs=1 semantics (output does not depend on jobs). 0 on success else kanzi Error code. */
int knzo_compress(const uint8_t* in, size_t n, const char* transform, const char* entropy,
                  int blockSize, int checksum, uint64_t origSize, int headerless,
                  uint8_t* out, size_t cap, size_t* outLen);
int knzo_compress_jobs(const uint8_t* in, size_t n, const char* transform, const char* entropy,
                       int blockSize, int checksum, uint64_t origSize, int headerless, int jobs,
                       uint8_t* out, size_t cap, size_t* outLen);
int knzo_compress_run(const uint8_t* in, size_t n, const char* transform, const char* entropy,
                      int blockSize, int checksum, uint64_t origSize, int headerless, int jobs,
                      uint64_t firstBlock, int finish, uint8_t* out, size_t cap, size_t* outLen, uint64_t* outBits);
int knzo_decompress(const uint8_t* in, size_t inLen, uint8_t* out, size_t cap, size_t* outLen);
/* the same with codec ids and an open bit writer / a start bit (tests/stub: a CPU stand-in for the device library) */
int knzo_compress_run_ids(const uint8_t* in, size_t n, uint64_t ttype, int etype, int blockSize, int checksum, int jobs,
                          uint64_t firstBlock, int finish, knzo_bw* w);
int knzo_decode_run(const uint8_t* in, uint64_t inBits, uint64_t startBit, uint64_t ttype, int etype, int checksumBits, int blockSize,
                    int64_t maxBlocks, uint8_t* out, size_t cap, size_t* outLen, uint64_t* endBit, int64_t* blocksDone);

/* XXHash32/64 as used for block checksums (util/XXHash.hpp:61-115,153-230), seed 0x4B414E5A */
uint32_t knzo_xxhash32(const uint8_t* p, size_t n, uint32_t seed);
uint64_t knzo_xxhash64(const uint8_t* p, size_t n, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
