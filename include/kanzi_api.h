/*
 * kanzi_api.h -- C API of the MI355X kanzi library (libkanzi_amd.so).
 *
 * Same names, signatures, parameter structs and error codes as the reference's C API so that C
 * callers and the ctypes shim (src/api/kanzi_c_api.py:88-137) bind to this library unchanged:
 *   compressor    src/api/Compressor.hpp:65-73 (cData), :80-116 (functions)
 *   decompressor  src/api/Decompressor.hpp:63-77 (dData), :83-117 (functions)
 * Behaviour mirrored from src/api/Compressor.cpp:183-358 and src/api/Decompressor.cpp:108-313:
 * names are validated and rewritten canonically, blockSize is rounded up to 16, compress() rejects
 * inSize > blockSize with ERR_INVALID_PARAM (18), *outSize is the number of bytes that reached the
 * sink during the call, the header's original size comes from fstat(dst), version = 0x010000.
 * All compute runs on the GPU through knz_hip.h; there is no CPU fallback.
 */
#ifndef KANZI_AMD_API_H
#define KANZI_AMD_API_H

#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KANZI_API __attribute__((visibility("default")))

struct cContext;
struct dContext;

struct cData {
    char transform[64];
    char entropy[16];
    size_t blockSize;
    unsigned int jobs;
    int checksum;
    int headerless;
};

struct dData {
    size_t bufferSize;
    unsigned int jobs;
    int headerless;
    char transform[64];
    char entropy[16];
    unsigned int blockSize;
    size_t originalSize;
    int checksum;
    int bsVersion;
};

KANZI_API unsigned int getCompressorVersion(void);
KANZI_API int initCompressor(struct cData* cParam, FILE* dst, struct cContext** ctx);
KANZI_API int compress(struct cContext* ctx, const unsigned char* src, size_t inSize, size_t* outSize);
KANZI_API int disposeCompressor(struct cContext** ctx, size_t* outSize);

KANZI_API unsigned int getDecompressorVersion(void);
KANZI_API int initDecompressor(struct dData* dParam, FILE* src, struct dContext** ctx);
KANZI_API int decompress(struct dContext* ctx, unsigned char* dst, size_t* inSize, size_t* outSize);
KANZI_API int disposeDecompressor(struct dContext** ctx);

#ifdef __cplusplus
}
#endif
#endif
