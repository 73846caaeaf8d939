/* TEST INFRASTRUCTURE ONLY -- a CPU stand-in for libknz_hip.so (include/knz_hip.h) built on the oracle, so that the HOST layer
 * (kanzi-cpp_amd/host/kanzi_amd.cpp: stream classes, batching, staging slots and worker thread, header parsing, seek, the C API)
 * can be exercised by the CPU-only test suite. "Device" pointers are host pointers. Never shipped, never loaded by the product:
 * tests/test_host_stub.py links it into a private copy of the host library. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/knz_hip.h"
#include "../../oracle/knz_oracle.h"

#include <pthread.h>
#include <time.h>

struct knz_ctx { char err[256]; int device; };

/* A model of SEVERAL devices (tests/test_host_stub.py: the 8-lane stream): KNZ_STUB_DEVICES devices, each of which runs one block
 * call at a time (a mutex per device) and takes KNZ_STUB_NS_PER_BYTE nanoseconds per input byte of the call, slept, so that what the
 * clock sees is how the host layer spreads its batches over the devices and not how fast the oracle is on the cores of the test box. */
#define STUB_MAX_DEV 16
static pthread_mutex_t g_dev_mu[STUB_MAX_DEV];
static pthread_once_t g_dev_once = PTHREAD_ONCE_INIT;
static int g_ndev = 1;
static long g_ns_per_byte = 0;
static void stub_init(void)
{
    const char* e = getenv("KNZ_STUB_DEVICES");
    g_ndev = e ? atoi(e) : 1;
    if (g_ndev < 1) g_ndev = 1;
    if (g_ndev > STUB_MAX_DEV) g_ndev = STUB_MAX_DEV;
    e = getenv("KNZ_STUB_NS_PER_BYTE");
    g_ns_per_byte = e ? atol(e) : 0;
    for (int i = 0; i < STUB_MAX_DEV; i++) pthread_mutex_init(&g_dev_mu[i], NULL);
}
static void stub_device_busy(const knz_ctx* c, size_t bytes)
{
    pthread_once(&g_dev_once, stub_init);
    if (g_ns_per_byte <= 0) return;
    const long long ns = (long long)bytes * g_ns_per_byte;
    struct timespec ts; ts.tv_sec = (time_t)(ns / 1000000000LL); ts.tv_nsec = (long)(ns % 1000000000LL);
    pthread_mutex_lock(&g_dev_mu[c->device]);
    nanosleep(&ts, NULL);
    pthread_mutex_unlock(&g_dev_mu[c->device]);
}

int knz_hip_device_count(int* count) { pthread_once(&g_dev_once, stub_init); *count = g_ndev; return 0; }
int knz_hip_create(int device, void* stream, knz_ctx** out)
{
    (void)stream;
    pthread_once(&g_dev_once, stub_init);
    if (device < 0 || device >= g_ndev) { *out = NULL; return -1; }
    *out = (knz_ctx*)calloc(1, sizeof(knz_ctx));
    if (*out) (*out)->device = device;
    return *out ? 0 : -1;
}
void knz_hip_destroy(knz_ctx* c) { free(c); }
const char* knz_hip_last_error(knz_ctx* c) { return c ? c->err : "null context"; }
size_t knz_hip_encode_bound(const knz_params* p, size_t n)
{
    const size_t bs = (size_t)p->block_size;
    const size_t nb = (n + bs - 1) / bs + 1;
    size_t bound = 2 * n + nb * 64 + ((n / 16384) + nb) * 608 + 4096;
    if (p->entropy_type == KNZ_E_ANS1) bound += (n / (4u << 20) + nb) * 256 * 576;
    return bound;
}

static int fail(knz_ctx* c, int code, const char* msg) { snprintf(c->err, sizeof(c->err), "%s", msg); return code; }

int knz_hip_encode_blocks(knz_ctx* c, const knz_params* p, const uint8_t* d_in, size_t n, const uint8_t* prologue, uint32_t prologue_bits,
                          int64_t first_block_id, int finish, uint8_t* d_out, size_t out_cap, uint64_t* out_bits)
{
    if (p->block_size < 1024 || (p->block_size & 15)) return fail(c, KNZ_ERR_INVALID_PARAM, "invalid block size");
    stub_device_busy(c, n);
    knzo_bw w;
    knzo_bw_init(&w, d_out, out_cap);
    for (uint32_t i = 0; i < prologue_bits; i += 8) {
        const unsigned nb = prologue_bits - i < 8 ? prologue_bits - i : 8;
        knzo_bw_bits(&w, (uint64_t)(prologue[i >> 3] >> (8 - nb)), nb);
    }
    const int jobs = p->jobs <= 0 ? 1 : (p->jobs > 64 ? 64 : p->jobs);
    const int rc = knzo_compress_run_ids(d_in, n, p->transform_type, p->entropy_type, p->block_size, p->checksum_bits, jobs,
                                         (uint64_t)first_block_id, finish, &w);
    if (rc) return fail(c, rc, "encode failed");
    if (w.overflow) return fail(c, KNZ_ERR_WRITE_FILE, "output buffer too small");
    *out_bits = w.bits;
    return 0;
}

void knzo_set_hosted(int hosted, unsigned applied, int origLen, uint64_t checksum);
void knzo_get_hosted(int* skipFlags, uint64_t* checksum);

int knz_hip_encode_block_hosted(knz_ctx* c, const knz_params* p, const knz_host_stages* hs, const uint8_t* d_in, size_t n, const uint8_t* prologue,
                                uint32_t prologue_bits, int64_t first_block_id, int finish, uint8_t* d_out, size_t out_cap, uint64_t* out_bits)
{
    if (!hs || n == 0 || n > (size_t)p->block_size) return fail(c, KNZ_ERR_INVALID_PARAM, "a hosted call takes exactly one block");
    knzo_set_hosted(hs->stages, hs->applied_mask, (int)hs->orig_len, hs->checksum);
    /* (a block size no smaller than the data makes the run encoder cut exactly one block; the slot capacities follow the stream's) */
    const int rc = knz_hip_encode_blocks(c, p, d_in, n, prologue, prologue_bits, first_block_id, finish, d_out, out_cap, out_bits);
    knzo_set_hosted(0, 0, 0, 0);
    return rc;
}

int knz_hip_decode_blocks(knz_ctx* c, const knz_params* p, const uint8_t* d_in, uint64_t in_bits, uint64_t start_bit, int64_t max_blocks,
                          uint8_t* d_out, size_t out_cap, uint64_t* out_bytes, uint64_t* end_bit, int64_t* blocks_done);
int knz_hip_decode_block_hosted(knz_ctx* c, const knz_params* p, int32_t host_stages, const uint8_t* d_in, uint64_t in_bits, uint64_t start_bit,
                                uint8_t* d_out, size_t out_cap, uint64_t* out_bytes, uint64_t* end_bit, uint32_t* skip_flags, uint64_t* checksum, int32_t* done)
{
    int64_t nb = 0;
    knzo_set_hosted(host_stages, 0, 0, 0);
    const int rc = knz_hip_decode_blocks(c, p, d_in, in_bits, start_bit, 1, d_out, out_cap, out_bytes, end_bit, &nb);
    int sk = 0xFF; uint64_t ck = 0;
    knzo_get_hosted(&sk, &ck);
    knzo_set_hosted(0, 0, 0, 0);
    if (skip_flags) *skip_flags = (uint32_t)sk;
    if (checksum) *checksum = ck;
    if (done) *done = (int32_t)nb;
    return rc;
}

int knz_hip_decode_blocks(knz_ctx* c, const knz_params* p, const uint8_t* d_in, uint64_t in_bits, uint64_t start_bit, int64_t max_blocks,
                          uint8_t* d_out, size_t out_cap, uint64_t* out_bytes, uint64_t* end_bit, int64_t* blocks_done)
{
    size_t ol = 0;
    uint64_t eb = 0;
    int64_t done = 0;
    const int ver = p->bs_version == 0 ? 6 : p->bs_version;
    if (ver < 0 || ver > 6) return fail(c, KNZ_ERR_STREAM_VERSION, "unknown bitstream version");
    knzo_set_bs_version(ver);
    const int rc = knzo_decode_run(d_in, in_bits, start_bit, p->transform_type, p->entropy_type, p->checksum_bits, p->block_size,
                                   max_blocks > 0 ? max_blocks : -1, d_out, out_cap, &ol, &eb, &done);
    knzo_set_bs_version(6);
    if (rc) return fail(c, rc, "decode failed");
    stub_device_busy(c, ol);
    if (out_bytes) *out_bytes = ol;
    if (end_bit) *end_bit = eb;
    if (blocks_done) *blocks_done = done;
    return 0;
}

int knz_hip_entropy_encode(knz_ctx* c, int entropy_type, const uint8_t* in, uint32_t n, uint8_t* out, size_t out_cap, uint64_t* out_bits)
{
    const int64_t bits = knzo_entropy_encode(entropy_type, in, n, out, out_cap);
    if (bits < 0) return fail(c, KNZ_ERR_PROCESS_BLOCK, "entropy encode failed");
    *out_bits = (uint64_t)bits;
    return 0;
}

int knz_hip_entropy_decode_v(knz_ctx* c, int entropy_type, int bs_version, const uint8_t* in, uint64_t in_bits, uint64_t start_bit, uint8_t* out,
                             uint32_t n, int32_t* decoded, uint64_t* used_bits)
{
    const int ver = bs_version == 0 ? 6 : bs_version;
    if (ver < 0 || ver > 6) return fail(c, KNZ_ERR_STREAM_VERSION, "unknown bitstream version");
    knzo_set_bs_version(ver);
    const int rc = knz_hip_entropy_decode(c, entropy_type, in, in_bits, start_bit, out, n, decoded, used_bits);
    knzo_set_bs_version(6);
    return rc;
}

int knz_hip_entropy_decode(knz_ctx* c, int entropy_type, const uint8_t* in, uint64_t in_bits, uint64_t start_bit, uint8_t* out, uint32_t n,
                           int32_t* decoded, uint64_t* used_bits)
{
    (void)c;
    knzo_br r;
    knzo_br_init(&r, in, in_bits);
    r.pos = start_bit;
    int res;
    switch (entropy_type) {
    case KNZ_E_NONE: res = knzo_none_decode_br(&r, out, n); break;
    case KNZ_E_HUFFMAN: res = knzo_huffman_decode_br(&r, out, n); break;
    case KNZ_E_FPAQ: res = knzo_fpaq_decode_br(&r, out, n); break;
    case KNZ_E_ANS0: res = knzo_ans_decode_br(&r, out, n, 0); break;
    case KNZ_E_ANS1: res = knzo_ans_decode_br(&r, out, n, 1); break;
    default: return KNZ_ERR_INVALID_CODEC;
    }
    *decoded = r.error ? -1 : res;
    if (used_bits) *used_bits = r.pos - start_bit;
    return 0;
}

int knz_hip_transform_forward(knz_ctx* c, int t, const uint8_t* in, int32_t n, uint8_t* out, int32_t dst_cap, int etype, int32_t* out_len, int32_t* ok)
{
    (void)c;
    int ol = 0;
    *ok = knzo_transform_forward(t, in, n, out, dst_cap, etype, &ol) == 1;
    *out_len = *ok ? ol : 0;
    return 0;
}

int knz_hip_transform_inverse(knz_ctx* c, int t, const uint8_t* in, int32_t n, uint8_t* out, int32_t dst_cap, int32_t* out_len, int32_t* ok)
{
    (void)c;
    int ol = 0;
    *ok = knzo_transform_inverse(t, in, n, out, dst_cap, &ol) == 1;
    *out_len = *ok ? ol : 0;
    return 0;
}

int knz_hip_transform_inverse_v(knz_ctx* c, int t, int bs_version, const uint8_t* in, int32_t n, uint8_t* out, int32_t dst_cap, int32_t* out_len,
                                int32_t* ok)
{
    const int ver = bs_version == 0 ? 6 : bs_version;
    if (ver < 0 || ver > 6) return fail(c, KNZ_ERR_STREAM_VERSION, "unknown bitstream version");
    knzo_set_bs_version(ver);
    const int rc = knz_hip_transform_inverse(c, t, in, n, out, dst_cap, out_len, ok);
    knzo_set_bs_version(6);
    return rc;
}

int knz_hip_malloc(knz_ctx* c, size_t bytes, void** p) { (void)c; *p = malloc(bytes ? bytes : 1); return *p ? 0 : -1; }
int knz_hip_free(knz_ctx* c, void* p) { (void)c; free(p); return 0; }
int knz_hip_memcpy_h2d(knz_ctx* c, void* d, const void* s, size_t n) { (void)c; memcpy(d, s, n); return 0; }
int knz_hip_memcpy_d2h(knz_ctx* c, void* d, const void* s, size_t n) { (void)c; memcpy(d, s, n); return 0; }
int knz_hip_sync(knz_ctx* c) { (void)c; return 0; }
int knz_hip_memcpy_h2d_async(knz_ctx* c, void* d, const void* s, size_t n, uint64_t* t) { (void)c; memcpy(d, s, n); *t = 1; return 0; }
int knz_hip_memcpy_d2h_async(knz_ctx* c, void* d, const void* s, size_t n, uint64_t* t) { (void)c; memcpy(d, s, n); *t = 1; return 0; }
int knz_hip_copy_wait(knz_ctx* c, uint64_t t) { (void)c; (void)t; return 0; }
int knz_hip_host_alloc(size_t bytes, void** p) { *p = malloc(bytes ? bytes : 1); return *p ? 0 : -1; }
int knz_hip_host_free(void* p) { free(p); return 0; }
int knz_hip_set_profiling(knz_ctx* c, int on) { (void)c; (void)on; return 0; }
int knz_hip_get_kernel_times(knz_ctx* c, knz_kernel_time* out, int cap) { (void)c; (void)out; (void)cap; return 0; }
int knz_hip_tune(const char* name, int value) { (void)name; (void)value; return 0; }
int knz_hip_shift_bits(knz_ctx* c, const uint8_t* in, uint64_t nbits, uint32_t r, uint8_t* out)
{
    (void)c;
    const uint64_t inBytes = (nbits + 7) >> 3, outBytes = (nbits + r + 7) >> 3;
    for (uint64_t q = 0; q < outBytes; q++) {
        const unsigned a = (q > 0 && q - 1 < inBytes) ? in[q - 1] : 0u, b = (q < inBytes) ? in[q] : 0u;
        out[q] = (uint8_t)((a << (8 - r)) | (b >> r));
    }
    return 0;
}
