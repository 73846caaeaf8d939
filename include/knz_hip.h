/*
 * knz_hip.h -- C ABI of the MI355X (gfx950) kanzi block pipeline.
 *
 * This is the thin device layer SURVEY.md section 8(b) calls "New thin C-ABI to HIP": one call
 * processes a batch of independent blocks that are already resident in HBM and returns the block
 * payloads exactly as the reference's EncodingTask would have put them in its private bitstream
 * (io/CompressedOutputStream.cpp:651-898), assembled MSB-first at bit granularity
 * (bitstream/DefaultOutputBitStream.hpp:97-131). Plain pointers and sizes only; all pointers
 * named d_* are device pointers, everything else is host memory. Every function returns 0 on
 * success, a negative value for a device/runtime failure (see knz_hip_last_error) or a positive
 * kanzi Error code (src/Error.hpp:26-48) for data errors.
 *
 * The C++ host mirror of the reference interfaces (Transform<byte>, EntropyEncoder/Decoder,
 * CompressedOutputStream/InputStream, include/kanzi_amd/) and the reference C API replacement
 * (include/kanzi_api.h) are built on top of these entry points.
 */
#ifndef KNZ_HIP_H
#define KNZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KNZ_API __attribute__((visibility("default")))

/* kanzi ids: entropy (entropy/EntropyEncoderFactory.hpp:37-52) and transforms (transform/TransformFactory.hpp:49-73) */
enum { KNZ_E_NONE = 0, KNZ_E_HUFFMAN = 1, KNZ_E_FPAQ = 2, KNZ_E_ANS0 = 5, KNZ_E_ANS1 = 8 };
enum { KNZ_T_NONE = 0, KNZ_T_BWT = 1, KNZ_T_LZ = 3, KNZ_T_RLT = 5, KNZ_T_ZRLT = 6, KNZ_T_MTFT = 7, KNZ_T_RANK = 8, KNZ_T_SRT = 13, KNZ_T_LZX = 16,
       KNZ_T_TIMESTAMP = 64 /* SBRT's third mode: no kanzi id, never part of a chain; per-stage entry points only */,
       KNZ_T_TEXT = 10, KNZ_T_UTF = 17 /* stages that run on the HOST in front of the device chain: knz_hip_encode_block_hosted / _decode_ */ };

/* kanzi error codes surfaced for data errors (src/Error.hpp:26-48) */
enum { KNZ_ERR_BLOCK_SIZE = 2, KNZ_ERR_INVALID_CODEC = 3, KNZ_ERR_READ_FILE = 11, KNZ_ERR_WRITE_FILE = 12,
       KNZ_ERR_PROCESS_BLOCK = 13, KNZ_ERR_INVALID_FILE = 15, KNZ_ERR_STREAM_VERSION = 16,
       KNZ_ERR_INVALID_PARAM = 18, KNZ_ERR_CRC_CHECK = 19 };

typedef struct knz_ctx knz_ctx;

/* Number of visible HIP devices. */
KNZ_API int knz_hip_device_count(int* count);

/* Create a context on `device`. `stream` is a hipStream_t (as void*) to enqueue on, or NULL to
 * let the context create its own. Workspaces are grown on demand and reused across calls. */
KNZ_API int knz_hip_create(int device, void* stream, knz_ctx** out);
KNZ_API void knz_hip_destroy(knz_ctx* ctx);
KNZ_API const char* knz_hip_last_error(knz_ctx* ctx);

/* Parameters of one batch of blocks: what CompressedOutputStream puts in the per-task Context
 * (io/CompressedOutputStream.cpp:491-496): transform ids (48 bits, 6 per stage, first stage in
 * bits 47..42), entropy id, block size, checksum width (0/32/64). */
typedef struct {
    uint64_t transform_type;
    int32_t entropy_type;
    int32_t block_size;
    int32_t checksum_bits;
    int32_t jobs;            /* reference job count being reproduced (0/1 = single). It only selects the
                                buffer slot, hence the capacities, a block sees. */
    int32_t bs_version;      /* decode only: bitstream version of the stream the blocks come from, as
                                CompressedInputStream puts it in the Context (io/CompressedInputStream.cpp:528-537).
                                0 (unset, the ABI default) or 6 = current. 1..5 select the old layouts the reference still reads (it treats every
                                version below 6 alike; a caller that parsed or was given version 0 passes 1): Huffman chunks
                                (entropy/HuffmanDecoder.cpp:349-459), the BWT block header
                                (transform/BWTBlockCodec.cpp:140-164) and LZ / LZX blocks (transform/LZCodec.cpp:614-760).
                                The encoder writes version 6 only, like the reference. */
} knz_params;

/* Upper bound, in bytes, of the bit-packed output of knz_hip_encode_blocks for n input bytes. */
KNZ_API size_t knz_hip_encode_bound(const knz_params* p, size_t n);

/*
 * Encode n bytes at d_in, cut into blocks of p->block_size (the last one may be short), each block
 * through TransformSequence::forward + block header + EntropyEncoder::encode, and append the
 * results in block order as CompressedOutputStream does (5-bit lw-3, lw-bit length, payload bits;
 * io/CompressedOutputStream.cpp:852-864).
 *
 *   prologue/prologue_bits : host bytes placed first (the stream header, or NULL/0)
 *   first_block_id         : 0-based id of the first block in this batch (only for buffer-capacity
 *                            modelling of the reference)
 *   finish                 : append the end-of-stream marker (5+3 zero bits, :415-417)
 *   d_out/out_cap          : device output, zero-filled by the call, out_cap >= knz_hip_encode_bound
 *   out_bits               : total bits written (host). Output bytes = (out_bits+7)/8.
 *
 * The call is synchronous with respect to the host when out_bits != NULL (it has to read the
 * length back); all device work is enqueued on the context's stream.
 */
KNZ_API int knz_hip_encode_blocks(knz_ctx* ctx, const knz_params* p, const uint8_t* d_in, size_t n,
                                  const uint8_t* prologue, uint32_t prologue_bits, int64_t first_block_id,
                                  int finish, uint8_t* d_out, size_t out_cap, uint64_t* out_bits);

/*
 * Decode a run of blocks: d_in holds the bit stream, the first block's 5-bit length prefix is at
 * bit `start_bit`. Blocks are decoded until the end marker, `max_blocks` blocks or `in_bits` is
 * exhausted. Decoded blocks are written back to back at d_out.
 *   out_bytes  : decoded byte count (host)
 *   end_bit    : bit position after the last consumed block / end marker (host, may be NULL)
 * Mirrors DecodingTask::run (io/CompressedInputStream.cpp:790-1041) incl. its accept/reject rules.
 */
KNZ_API int knz_hip_decode_blocks(knz_ctx* ctx, const knz_params* p, const uint8_t* d_in, uint64_t in_bits,
                                  uint64_t start_bit, int64_t max_blocks, uint8_t* d_out, size_t out_cap,
                                  uint64_t* out_bytes, uint64_t* end_bit, int64_t* blocks_done);

/*
 * Chains whose first stages run on the host (the reference's level presets 5 and 6: TEXT + UTF in front of BWT + RANK / SRT + ZRLT,
 * app/BlockCompressor.cpp:583-591). The host applies those stages to ONE block (they change its length by a different amount per block)
 * and hands over what TransformSequence::forward would have left for the next stage (transform/TransformSequence.hpp:88-162):
 *   stages        how many leading stages of p->transform_type the host owns (each KNZ_T_TEXT or KNZ_T_UTF)
 *   applied_mask  bit i set = stage i succeeded (its skip flag is clear, and it counts as a buffer swap for the capacities the
 *                 device stages see); clear = the stage refused the block and the block went on unchanged
 *   orig_len      length of the block before any stage (decides copy blocks and the buffer sizes of the reference)
 *   checksum      XXHash32 / 64 of the ORIGINAL block when p->checksum_bits != 0 (the device only ever sees the transformed bytes)
 * d_in holds the n bytes the host stages left. Everything else as knz_hip_encode_blocks with exactly one block; the block header
 * carries the skip flags of all stages (a separate byte when the chain has more than four, io/CompressedOutputStream.cpp:791-799).
 */
typedef struct { int32_t stages; uint32_t applied_mask; uint32_t orig_len; uint32_t reserved; uint64_t checksum; } knz_host_stages;
KNZ_API int knz_hip_encode_block_hosted(knz_ctx* ctx, const knz_params* p, const knz_host_stages* hs, const uint8_t* d_in, size_t n,
                                        const uint8_t* prologue, uint32_t prologue_bits, int64_t first_block_id, int finish,
                                        uint8_t* d_out, size_t out_cap, uint64_t* out_bits);
/* The way back: ONE block is entropy-decoded and taken through the inverse of the device stages; the host stages are left to the
 * caller, who gets the block's skip flags (bit 7 = stage 0; 0xFF for a copy block) and its stored checksum (not verified here: it is
 * the checksum of the original block). *done = 0 when the end marker was found instead of a block. */
KNZ_API int knz_hip_decode_block_hosted(knz_ctx* ctx, const knz_params* p, int32_t host_stages, const uint8_t* d_in, uint64_t in_bits,
                                        uint64_t start_bit, uint8_t* d_out, size_t out_cap, uint64_t* out_bytes, uint64_t* end_bit,
                                        uint32_t* skip_flags, uint64_t* checksum, int32_t* done);

/* ---- per-stage entry points (host buffers in/out; used by the host mirror classes and tests) ---- */

/* EntropyEncoder::encode for one buffer (entropy/EntropyEncoder.hpp:30-37). out receives the bits
 * MSB-first from bit 0; *out_bits the exact bit count (dispose() bits included for FPAQ). */
KNZ_API int knz_hip_entropy_encode(knz_ctx* ctx, int entropy_type, const uint8_t* in, uint32_t n,
                                   uint8_t* out, size_t out_cap, uint64_t* out_bits);
/* EntropyDecoder::decode: reads from bit `start_bit` of `in`; *used_bits = bits consumed.
 * Returns 0 and *decoded == n on success; *decoded mirrors the reference's return value otherwise. */
KNZ_API int knz_hip_entropy_decode(knz_ctx* ctx, int entropy_type, const uint8_t* in, uint64_t in_bits,
                                   uint64_t start_bit, uint8_t* out, uint32_t n, int32_t* decoded,
                                   uint64_t* used_bits);
/* The same with the bitstream version the bits come from (knz_params.bs_version: 0 / 6 current, 1..5 the old Huffman chunk
 * layout): what HuffmanDecoder takes from its Context (entropy/HuffmanDecoder.hpp:32, HuffmanDecoder.cpp:349-352). */
KNZ_API int knz_hip_entropy_decode_v(knz_ctx* ctx, int entropy_type, int bs_version, const uint8_t* in, uint64_t in_bits,
                                     uint64_t start_bit, uint8_t* out, uint32_t n, int32_t* decoded,
                                     uint64_t* used_bits);

/* Transform<byte>::forward / inverse for one buffer (src/Transform.hpp:38-45). dst_cap mirrors
 * SliceArray::_length - _index of the destination (it changes results for ZRLT/RLT; LZ/LZX refuse a
 * destination below getMaxEncodedLength). *ok = 1 when the reference would return true. entropy_type: the
 * stream's entropy id (RLT escape choice), -1 if unset. Classes replaced, by transform_type:
 *   KNZ_T_BWT   BWTBlockCodec        transform/BWTBlockCodec.cpp:32-87,89-168 (+ BWT.cpp, DivSufSort.cpp)
 *   KNZ_T_MTFT / KNZ_T_RANK / KNZ_T_TIMESTAMP   SBRT modes 1 / 2 / 3   transform/SBRT.cpp:46-97,99-145
 *   KNZ_T_SRT   SRT                  transform/SRT.cpp:22-109,111-204
 *   KNZ_T_ZRLT  ZRLT                 transform/ZRLT.cpp:27-117,119-215
 *   KNZ_T_RLT   RLT                  transform/RLT.cpp:39-221,247-369
 *   KNZ_T_LZ / KNZ_T_LZX   LZCodec -> LZXCodec<false> / LZXCodec<true>   transform/LZCodec.cpp:119-456,470-640
 * (the inverse of LZ/LZX expects what the reference expects: two readable bytes behind `in + n`,
 * LZCodec.cpp:486-490; the library stages the input itself, so callers need not pad). */
KNZ_API int knz_hip_transform_forward(knz_ctx* ctx, int transform_type, const uint8_t* in, int32_t n,
                                      uint8_t* out, int32_t dst_cap, int entropy_type, int32_t* out_len, int32_t* ok);
KNZ_API int knz_hip_transform_inverse(knz_ctx* ctx, int transform_type, const uint8_t* in, int32_t n,
                                      uint8_t* out, int32_t dst_cap, int32_t* out_len, int32_t* ok);
/* The same for a block of an older stream (bs_version as in knz_params: the BWT block header of versions below 6,
 * transform/BWTBlockCodec.cpp:140-164, and the LZ / LZX token layout, transform/LZCodec.cpp:614-760): what the transform
 * classes take from the "bsVersion" entry of their Context. */
KNZ_API int knz_hip_transform_inverse_v(knz_ctx* ctx, int transform_type, int bs_version, const uint8_t* in, int32_t n,
                                        uint8_t* out, int32_t dst_cap, int32_t* out_len, int32_t* ok);

/* Device-memory helpers so that non-HIP hosts (ctypes, cgo, JNI) can stage data. */
KNZ_API int knz_hip_malloc(knz_ctx* ctx, size_t bytes, void** d_ptr);
KNZ_API int knz_hip_free(knz_ctx* ctx, void* d_ptr);
KNZ_API int knz_hip_memcpy_h2d(knz_ctx* ctx, void* d_dst, const void* src, size_t bytes);
KNZ_API int knz_hip_memcpy_d2h(knz_ctx* ctx, void* dst, const void* d_src, size_t bytes);
KNZ_API int knz_hip_sync(knz_ctx* ctx);
/* The same copies, queued on the context's copy streams (one per direction) and returning at once: they run beside the kernels
 * of a knz_hip_encode_blocks / knz_hip_decode_blocks call that another host thread has in flight (no context lock is taken), which
 * is how the stream classes overlap PCIe with compute. Host memory must be page-locked (knz_hip_host_alloc); the device buffer
 * must not be in use by a call in flight. knz_hip_copy_wait blocks until the copy behind `ticket` is complete (a ticket can be
 * waited for once; 0 and unknown tickets return at once). */
KNZ_API int knz_hip_memcpy_h2d_async(knz_ctx* ctx, void* d_dst, const void* src, size_t bytes, uint64_t* ticket);
KNZ_API int knz_hip_memcpy_d2h_async(knz_ctx* ctx, void* dst, const void* d_src, size_t bytes, uint64_t* ticket);
KNZ_API int knz_hip_copy_wait(knz_ctx* ctx, uint64_t ticket);
/* Page-locked host memory for staging buffers (what knz_hip_memcpy_h2d / _d2h move at full PCIe rate). */
KNZ_API int knz_hip_host_alloc(size_t bytes, void** ptr);
KNZ_API int knz_hip_host_free(void* ptr);

/* d_out = r zero bits (1..7) followed by the nbits bits of d_in (MSB-first byte streams, (r + nbits + 7) / 8 bytes written): what a
 * host that appends bit runs of independent batches in order needs (io/CompressedOutputStream.cpp:835-868 appends at bit granularity).
 * Queued on the context's stream. */
KNZ_API int knz_hip_shift_bits(knz_ctx* ctx, const uint8_t* d_in, uint64_t nbits, uint32_t r, uint8_t* d_out);

/* Timing of the kernels of the last encode/decode call, measured with HIP events on the context's
 * stream: name/ms pairs. Returns the number of entries written (<= cap). */
typedef struct { char name[48]; float ms; uint64_t launches; } knz_kernel_time;
KNZ_API int knz_hip_set_profiling(knz_ctx* ctx, int enabled);
KNZ_API int knz_hip_get_kernel_times(knz_ctx* ctx, knz_kernel_time* out, int cap);

/* Developer / test knobs, process-wide (the same names in upper case with a KNZ_ prefix are read from the environment once, when
 * the library first needs them): "bwt_nsym" (symbols of the suffix sort's first round, 0 = four or five by the data's entropy),
 * "bwt_no_run_round", "bwt_run_fallback" (1: force the general path for runs), "bwt_no_super", "bwt_no_text_round", "bwt_no_run_offsets",
 * "bwt_no_probe", "bwt_link" (0: the suffix sort's link step for groups inside long repeats off), "bwt_stats", "bwt_split" (parts of a batch the BWT stages run in, 1..4), "rs_onesweep" (0: radix passes with a
 * counting kernel of their own), "lz_serial_decode" (1: LZ / LZX blocks decoded by one wave each), "mtf_tile" (0: MTFT tile size by
 * batch size, 1024 / 4096 force it), "mtf_chain" (1: MTFT forward ranks by the byte-serial kernel of rounds 2-4).
 * Returns 0, or -1 for an unknown name. No knob changes a result. */
KNZ_API int knz_hip_tune(const char* name, int value);

#ifdef __cplusplus
}
#endif
#endif
