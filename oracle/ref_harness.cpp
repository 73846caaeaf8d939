// TEST INFRASTRUCTURE ONLY -- never linked into the product path.
//
// Thin extern "C" harness over the *unmodified* reference classes compiled from
// /root/reference/src (see oracle/Makefile, target _ref). It exposes per-stage and
// whole-stream entry points so that tests (ctypes) can pin the C restatement in
// oracle/*.c and the HIP path against the real reference:
//   * per-stage transforms   -> TransformFactory<byte>::newTransform (src/transform/TransformFactory.hpp:208)
//   * per-stage entropy      -> EntropyEncoderFactory::newEncoder / EntropyDecoderFactory::newDecoder
//   * whole stream           -> CompressedOutputStream / CompressedInputStream (src/io/*.hpp)
// Built only where /root/reference exists (this container); the resulting .so lives in
// oracle/_ref/ (git-ignored, travels with gpurun) and is used as cpu_baseline "reference".
#include <chrono>
#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>
#include <streambuf>

#include "types.hpp"
#include "Context.hpp"
#include "SliceArray.hpp"
#include "transform/TransformFactory.hpp"
#include "transform/SBRT.hpp"
#include "entropy/EntropyEncoderFactory.hpp"
#include "entropy/EntropyDecoderFactory.hpp"
#include "bitstream/DefaultOutputBitStream.hpp"
#include "bitstream/DefaultInputBitStream.hpp"
#include "io/CompressedOutputStream.hpp"
#include "io/CompressedInputStream.hpp"
#include "io/IOException.hpp"

using namespace kanzi;

namespace {
// Fixed-capacity output streambuf over caller memory.
class MemOutBuf : public std::streambuf {
public:
    MemOutBuf(char* p, size_t cap) : _base(p), _cap(cap), _pos(0), _overflow(false) {}
    size_t size() const { return _pos; }
    bool overflowed() const { return _overflow; }
protected:
    std::streamsize xsputn(const char* s, std::streamsize n) override {
        if (_pos + size_t(n) > _cap) { _overflow = true; return 0; }
        memcpy(_base + _pos, s, size_t(n));
        _pos += size_t(n);
        return n;
    }
    int_type overflow(int_type c) override {
        if (c == traits_type::eof()) return traits_type::not_eof(c);
        char ch = char(c);
        return xsputn(&ch, 1) == 1 ? c : traits_type::eof();
    }
    pos_type seekoff(off_type off, std::ios_base::seekdir dir, std::ios_base::openmode) override {
        if (dir == std::ios_base::cur && off == 0) return pos_type(off_type(_pos));
        return pos_type(off_type(-1));
    }
private:
    char* _base; size_t _cap; size_t _pos; bool _overflow;
};

class MemInBuf : public std::streambuf {
public:
    MemInBuf(const char* p, size_t n) {
        char* b = const_cast<char*>(p);
        setg(b, b, b + n);
    }
};
}

extern "C" {

// Returns 1 if the transform sequence reported success, 0 otherwise. *outLen = bytes produced.
// dstCap is the SliceArray::_length of the destination (capacity affects ZRLT/RLT results).
int ref_transform(const char* name, int forward, const uint8_t* in, int n, int srcCap,
                  uint8_t* out, int dstCap, const char* entropy, int* outLen, int* skipFlags)
{
    try {
        Context ctx;
        ctx.putInt("bsVersion", 6);
        ctx.putInt("size", n);
        if (entropy && entropy[0]) ctx.putString("entropy", entropy);
        uint64 t = TransformFactory<byte>::getType(name);
        TransformSequence<byte>* seq = TransformFactory<byte>::newTransform(ctx, t);
        std::vector<byte> src(size_t(srcCap > n ? srcCap : n));
        memcpy(src.data(), in, size_t(n));
        SliceArray<byte> sa1(src.data(), int(src.size()), 0);
        SliceArray<byte> sa2(reinterpret_cast<byte*>(out), dstCap, 0);
        bool res;
        if (forward) {
            res = seq->forward(sa1, sa2, n);
            if (skipFlags) *skipFlags = int(seq->getSkipFlags());
        } else {
            if (skipFlags) seq->setSkipFlags(byte(*skipFlags));
            res = seq->inverse(sa1, sa2, n);
        }
        *outLen = sa2._index;
        delete seq;
        return res ? 1 : 0;
    } catch (const std::exception&) {
        return -1;
    }
}

// SBRT with an explicit mode (1 MTF, 2 RANK, 3 TIMESTAMP): TIMESTAMP has no factory id.
int ref_sbrt(int mode, int forward, const uint8_t* in, int n, uint8_t* out, int dstCap, int* outLen)
{
    try {
        SBRT t(mode);
        std::vector<byte> src(size_t(n > 0 ? n : 1));
        memcpy(src.data(), in, size_t(n));
        SliceArray<byte> sa1(src.data(), n, 0);
        SliceArray<byte> sa2(reinterpret_cast<byte*>(out), dstCap, 0);
        const bool res = forward ? t.forward(sa1, sa2, n) : t.inverse(sa1, sa2, n);
        *outLen = sa2._index;
        return res ? 1 : 0;
    } catch (const std::exception&) {
        return -1;
    }
}

static int entropyType(const char* name) { return int(EntropyEncoderFactory::getType(name)); }

// Encodes n bytes with the named entropy codec. Returns number of bits written or -1.
long long ref_entropy_encode(const char* name, const uint8_t* in, int n, uint8_t* out, size_t outCap)
{
    try {
        MemOutBuf buf(reinterpret_cast<char*>(out), outCap);
        std::ostream os(&buf);
        DefaultOutputBitStream obs(os, 65536);
        Context ctx;
        ctx.putInt("bsVersion", 6);
        ctx.putInt("size", n);
        EntropyEncoder* ee = EntropyEncoderFactory::newEncoder(obs, ctx, short(entropyType(name)));
        int r = ee->encode(reinterpret_cast<const byte*>(in), 0, uint(n));
        ee->dispose();
        delete ee;
        obs.close();
        if (r != n || buf.overflowed()) return -1;
        return (long long)obs.written();
    } catch (const std::exception&) {
        return -1;
    }
}

// Decodes n bytes. Returns the decoder's return value (n on success), -2 on exception.
int ref_entropy_decode(const char* name, const uint8_t* in, size_t inBytes, uint8_t* out, int n)
{
    try {
        MemInBuf buf(reinterpret_cast<const char*>(in), inBytes);
        std::istream is(&buf);
        DefaultInputBitStream ibs(is, 65536);
        Context ctx;
        ctx.putInt("bsVersion", 6);
        ctx.putInt("size", n);
        EntropyDecoder* ed = EntropyDecoderFactory::newDecoder(ibs, ctx, short(entropyType(name)));
        int r = ed->decode(reinterpret_cast<byte*>(out), 0, uint(n));
        ed->dispose();
        delete ed;
        return r;
    } catch (const std::exception&) {
        return -2;
    }
}

// Whole-stream compress through CompressedOutputStream. Returns 0 or an error code (<0 on exception).
int ref_compress_stream(const uint8_t* in, size_t n, const char* transform, const char* entropy,
                        int blockSize, int jobs, int checksum, unsigned long long origSize,
                        int headerless, uint8_t* out, size_t outCap, size_t* outLen)
{
    try {
        MemOutBuf buf(reinterpret_cast<char*>(out), outCap);
        std::ostream os(&buf);
        {
            CompressedOutputStream cos(os, jobs, entropy, transform, blockSize, checksum,
                                       uint64(origSize), nullptr, headerless != 0);
            size_t off = 0;
            while (off < n) {
                size_t c = n - off < (size_t(1) << 26) ? n - off : (size_t(1) << 26);
                cos.write(reinterpret_cast<const char*>(in) + off, std::streamsize(c));
                off += c;
            }
            cos.close();
        }
        *outLen = buf.size();
        return buf.overflowed() ? -3 : 0;
    } catch (const IOException& e) {
        return e.error();
    } catch (const std::exception&) {
        return -1;
    }
}

int ref_decompress_stream(const uint8_t* in, size_t inLen, int jobs, uint8_t* out, size_t outCap,
                          size_t* outLen)
{
    try {
        MemInBuf buf(reinterpret_cast<const char*>(in), inLen);
        std::istream is(&buf);
        CompressedInputStream cis(is, jobs);
        size_t off = 0;
        while (off < outCap) {
            size_t c = outCap - off < (size_t(1) << 26) ? outCap - off : (size_t(1) << 26);
            cis.read(reinterpret_cast<char*>(out) + off, std::streamsize(c));
            size_t got = size_t(cis.gcount());
            off += got;
            if (got == 0) break;
        }
        cis.close();
        *outLen = off;
        return 0;
    } catch (const IOException& e) {
        return e.error();
    } catch (const std::exception&) {
        return -1;
    }
}

// Compress + decompress one buffer, timed inside C with a steady clock around the stream objects only
// (construction .. close), so that no caller-side marshalling is inside the figures bench.py reports as
// cpu_baseline. Buffers are the caller's; `back` must hold n bytes. Returns 0, an error code, or -4 when
// the round trip does not reproduce the input.
int ref_time_roundtrip(const uint8_t* in, size_t n, const char* transform, const char* entropy, int blockSize, int jobs,
                       uint8_t* comp, size_t compCap, size_t* compLen, uint8_t* back, double* encSec, double* decSec)
{
    const auto t0 = std::chrono::steady_clock::now();
    int rc = ref_compress_stream(in, n, transform, entropy, blockSize, jobs, 0, 0ull, 0, comp, compCap, compLen);
    const auto t1 = std::chrono::steady_clock::now();
    if (rc != 0) return rc;
    size_t got = 0;
    rc = ref_decompress_stream(comp, *compLen, jobs, back, n, &got);
    const auto t2 = std::chrono::steady_clock::now();
    if (rc != 0) return rc;
    *encSec = std::chrono::duration<double>(t1 - t0).count();
    *decSec = std::chrono::duration<double>(t2 - t1).count();
    return (got == n && memcmp(in, back, n) == 0) ? 0 : -4;
}

// Decompress with a block range ("from" / "to" in the Context, io/CompressedInputStream.cpp:836-868).
int ref_decompress_range(const uint8_t* in, size_t inLen, int jobs, int from, int to, uint8_t* out, size_t outCap, size_t* outLen)
{
    try {
        MemInBuf buf(reinterpret_cast<const char*>(in), inLen);
        std::istream is(&buf);
        Context ctx;
        ctx.putInt("jobs", jobs);
        ctx.putInt("from", from);
        ctx.putInt("to", to);
        CompressedInputStream cis(is, ctx);
        size_t off = 0;
        while (off < outCap) {
            size_t c = outCap - off < (size_t(1) << 26) ? outCap - off : (size_t(1) << 26);
            cis.read(reinterpret_cast<char*>(out) + off, std::streamsize(c));
            size_t got = size_t(cis.gcount());
            off += got;
            if (got == 0) break;
        }
        cis.close();
        *outLen = off;
        return 0;
    } catch (const IOException& e) {
        return e.error();
    } catch (const std::exception&) {
        return -1;
    }
}

}
