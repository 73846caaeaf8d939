// Stages of a transform chain that run on the host, in front of the device chain (text_codec.cpp): the reference's TEXT and UTF transforms.
#pragma once
#include <cstdint>

namespace kanzi_amd {
namespace hoststage {

// Global::DataType (Global.hpp:29): what the first stages learn about a block and later ones ask for
enum { DT_UNDEFINED = 0, DT_TEXT, DT_MULTIMEDIA, DT_EXE, DT_NUMERIC, DT_BASE64, DT_DNA, DT_BIN, DT_UTF8, DT_SMALL_ALPHABET };

uint32_t magicOf(const uint8_t* p);                        // Magic::getType: 4 readable bytes
int presetDataType(const uint8_t* block, int n);          // io/CompressedOutputStream.cpp:724-733

// variant 1 / 2 = TextCodec1 / TextCodec2. forward: false = the stage is skipped (the block goes on unchanged), *dataType is updated either way.
// blockSize is the STREAM's block size (it sizes the hash map, hence decides which words collide), not the block's length.
bool textForward(int variant, const uint8_t* src, int count, uint8_t* dst, int dstCap, int blockSize, int bsVersion, int* dataType, int* outLen);
bool textInverse(int variant, const uint8_t* src, int count, uint8_t* dst, int dstCap, int blockSize, int bsVersion, int* outLen);
int textVariantFor(const char* entropy);                   // transform/TransformFactory.hpp:225-242
bool utfForward(const uint8_t* src, int count, uint8_t* dst, int dstCap, int* dataType, int* outLen);
bool utfInverse(const uint8_t* src, int count, uint8_t* dst, int dstCap, int* outLen);

}  // namespace hoststage
}  // namespace kanzi_amd
