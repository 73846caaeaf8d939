// kanzi_amd.hpp -- C++ host mirror of the reference interfaces for the accelerated path.
//
// Same class names, method signatures, argument meaning and error behaviour as the reference so
// that code written against kanzi-cpp's block pipeline compiles against this header with a
// namespace change (kanzi -> kanzi_amd):
//   SliceArray<T>                 src/SliceArray.hpp:24-68
//   Transform<T>                  src/Transform.hpp:38-45
//   EntropyEncoder / Decoder      src/EntropyEncoder.hpp:30-39, src/EntropyDecoder.hpp:30-39
//   OutputBitStream / Input...    src/OutputBitStream.hpp:30-49, src/InputBitStream.hpp:29-50
//   Default{Output,Input}BitStream src/bitstream/DefaultOutputBitStream.hpp, DefaultInputBitStream.hpp
//   TransformFactory / Sequence   src/transform/TransformFactory.hpp:49-137,208-308, TransformSequence.hpp:88-265
//   Entropy{En,De}coderFactory    src/entropy/EntropyEncoderFactory.hpp:37-175, EntropyDecoderFactory.hpp
//   CompressedOutputStream        src/io/CompressedOutputStream.hpp:140-172 (both constructors incl. the ThreadPool*
//                                 positional of the concurrent build, listeners, flush/tellp/seekp :228-243)
//   CompressedInputStream         src/io/CompressedInputStream.hpp:180-230 (same; tellg/seekg/putback/unget :306-327)
//   Listener<T> / Event           src/Listener.hpp:24-32, src/Event.hpp:29-100 (accepted and kept; the device path raises no
//                                 per-block events -- SURVEY.md marks listeners out of scope)
//   ThreadPool                    src/concurrent.hpp (opaque here: accepted where the reference takes one, never used --
//                                 the device is the pool)
// Every forward/inverse/encode/decode call runs on the GPU through the C ABI in knz_hip.h
// (libknz_hip.so). There is no CPU implementation behind these classes: without a GPU the
// constructors throw.
#ifndef KANZI_AMD_HPP
#define KANZI_AMD_HPP

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <exception>
#include <istream>
#include <mutex>
#include <thread>
#include <map>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "knz_hip.h"

namespace kanzi_amd {

typedef uint8_t byte;
typedef unsigned int uint;
typedef uint64_t uint64;
typedef int64_t int64;

struct Error {
    enum ErrorCode {
        ERR_MISSING_PARAM = 1, ERR_BLOCK_SIZE = 2, ERR_INVALID_CODEC = 3, ERR_CREATE_COMPRESSOR = 4,
        ERR_CREATE_DECOMPRESSOR = 5, ERR_OUTPUT_IS_DIR = 6, ERR_OVERWRITE_FILE = 7, ERR_CREATE_FILE = 8,
        ERR_CREATE_BITSTREAM = 9, ERR_OPEN_FILE = 10, ERR_READ_FILE = 11, ERR_WRITE_FILE = 12,
        ERR_PROCESS_BLOCK = 13, ERR_CREATE_CODEC = 14, ERR_INVALID_FILE = 15, ERR_STREAM_VERSION = 16,
        ERR_CREATE_STREAM = 17, ERR_INVALID_PARAM = 18, ERR_CRC_CHECK = 19, ERR_RESERVED_NAME = 20, ERR_UNKNOWN = 127
    };
};

class IOException : public std::runtime_error {
public:
    IOException(const std::string& msg, int error = Error::ERR_UNKNOWN) : std::runtime_error(msg), _code(error) {}
    int error() const { return _code; }
private:
    int _code;
};

class BitStreamException : public std::runtime_error {
public:
    enum { UNDEFINED = 0, INPUT_OUTPUT = 1, END_OF_STREAM = 2, INVALID_STREAM = 3, STREAM_CLOSED = 4 };
    BitStreamException(const std::string& msg, int code = UNDEFINED) : std::runtime_error(msg), _code(code) {}
    int error() const { return _code; }
private:
    int _code;
};

template <class T>
class SliceArray {
public:
    T* _array;
    int _length;   // capacity
    int _index;
    SliceArray(T* arr, int len, int index = 0) : _array(arr), _length(len), _index(index) {}
    static bool isValid(const SliceArray& sa) { return (sa._array != nullptr) && (sa._index >= 0) && (sa._length >= 0) && (sa._index <= sa._length); }
};

// String-keyed parameter map (src/Context.hpp:49-86). Keys read here: bsVersion, entropy, jobs, size.
class Context {
public:
    bool has(const std::string& key) const { return _ints.count(key) || _strs.count(key); }
    int getInt(const std::string& key, int def = 0) const { auto it = _ints.find(key); return it == _ints.end() ? def : int(it->second); }
    int64 getLong(const std::string& key, int64 def = 0) const { auto it = _ints.find(key); return it == _ints.end() ? def : it->second; }
    std::string getString(const std::string& key, const std::string& def = "") const { auto it = _strs.find(key); return it == _strs.end() ? def : it->second; }
    void putInt(const std::string& key, int v) { _ints[key] = v; }
    void putLong(const std::string& key, int64 v) { _ints[key] = v; }
    void putString(const std::string& key, const std::string& v) { _strs[key] = v; }
private:
    std::map<std::string, int64> _ints;
    std::map<std::string, std::string> _strs;
};

// src/concurrent.hpp: the reference's stream constructors take a thread pool for their per-block tasks. Here the blocks of a batch
// run on the device, so the type only has to exist for the call forms of src/api/Compressor.cpp:230-237 and
// Decompressor.cpp:159-166 (positional nullptr) to compile; a non-null pool is accepted and ignored.
class ThreadPool;

// src/Listener.hpp:24-32
template <class T>
class Listener {
public:
    Listener() {}
    virtual void processEvent(const T& evt) = 0;
    virtual ~Listener() {}
};

// src/Event.hpp:29-100, the part callers of addListener() need to compile: the event types, ids, sizes and hashes.
class Event {
public:
    enum Type { COMPRESSION_START, COMPRESSION_END, BEFORE_TRANSFORM, AFTER_TRANSFORM, BEFORE_ENTROPY, AFTER_ENTROPY,
                DECOMPRESSION_START, DECOMPRESSION_END, AFTER_HEADER_DECODING, BLOCK_INFO };
    enum HashType { NO_HASH, SIZE_32, SIZE_64 };
    Event(Type type, int id, const std::string& msg) : _type(type), _id(id), _size(0), _hash(0), _hashType(NO_HASH), _msg(msg) {}
    Event(Type type, int id, int64 size, uint64 hash = 0, HashType hashType = NO_HASH)
        : _type(type), _id(id), _size(size), _hash(hash), _hashType(hashType) {}
    virtual ~Event() {}
    int getId() const { return _id; }
    int64 getSize() const { return _size; }
    Type getType() const { return _type; }
    uint64 getHash() const { return _hashType != NO_HASH ? _hash : 0; }
    HashType getHashType() const { return _hashType; }
    std::string toString() const { return _msg; }
private:
    Type _type; int _id; int64 _size; uint64 _hash; HashType _hashType; std::string _msg;
};

class OutputBitStream {
public:
    virtual void writeBit(int bit) = 0;
    virtual uint writeBits(uint64 bits, uint length) = 0;
    virtual uint writeBits(const byte bits[], uint length) = 0;
    virtual void close() = 0;
    virtual uint64 written() const = 0;
    virtual ~OutputBitStream() {}
};

class InputBitStream {
public:
    virtual int readBit() = 0;
    virtual uint64 readBits(uint length) = 0;
    virtual uint readBits(byte bits[], uint length) = 0;
    virtual void close() = 0;
    virtual uint64 read() const = 0;
    virtual bool hasMoreToRead() = 0;
    virtual ~InputBitStream() {}
};

// MSB-first bit streams over std streams (bitstream/DefaultOutputBitStream.hpp:83-131, DefaultInputBitStream.hpp:88-150)
class DefaultOutputBitStream : public OutputBitStream {
public:
    explicit DefaultOutputBitStream(std::ostream& os, uint bufferSize = 65536);
    ~DefaultOutputBitStream();
    void writeBit(int bit);
    uint writeBits(uint64 bits, uint length);
    uint writeBits(const byte bits[], uint length);
    void close();
    uint64 written() const { return _written; }
    bool isClosed() const { return _closed; }
private:
    std::ostream& _os;
    std::vector<byte> _buf;
    uint64 _current;
    uint _avail;       // free bits in _current
    uint64 _written;
    bool _closed;
    void push();
    void flush();
};

class DefaultInputBitStream : public InputBitStream {
public:
    explicit DefaultInputBitStream(std::istream& is, uint bufferSize = 65536);
    ~DefaultInputBitStream();
    int readBit();
    uint64 readBits(uint length);
    uint readBits(byte bits[], uint length);
    void close();
    uint64 read() const { return _read; }
    bool hasMoreToRead();
    // Device decoders need the bits they will consume in one buffer: returns everything not yet
    // consumed (bit 0 of the result = next unread bit) without consuming it, and lets the caller
    // advance afterwards.
    void peekRemaining(const byte** data, uint64* startBit, uint64* endBit);
    // the same, but only as far as `wantBits` behind the read position (less at the end of the stream)
    void peekAhead(uint64 wantBits, const byte** data, uint64* startBit, uint64* endBit);
    void skip(uint64 nbits);
private:
    std::istream& _is;
    std::vector<byte> _data;   // bytes fetched from the stream so far and not yet fully consumed
    uint64 _pos;               // bit position in _data
    uint64 _read;
    bool _closed;
    bool _eof;
    uint _chunk;
    bool fill(uint64 needBits);
};

template <class T>
class Transform {
public:
    virtual bool forward(SliceArray<T>& src, SliceArray<T>& dst, int length) = 0;
    virtual bool inverse(SliceArray<T>& src, SliceArray<T>& dst, int length) = 0;
    virtual int getMaxEncodedLength(int srcLen) const = 0;
    virtual ~Transform() {}
};

class EntropyEncoder {
public:
    virtual int encode(const byte block[], uint blkptr, uint len) = 0;
    virtual OutputBitStream& getBitStream() const = 0;
    virtual void dispose() = 0;
    virtual ~EntropyEncoder() {}
};

class EntropyDecoder {
public:
    virtual int decode(byte block[], uint blkptr, uint len) = 0;
    virtual InputBitStream& getBitStream() const = 0;
    virtual void dispose() = 0;
    virtual ~EntropyDecoder() {}
};

// One device context per process and device id (lazy). Throws IOException when no GPU is usable.
knz_ctx* deviceContext(int device = -1);
// The lanes the stream classes spread their batches over: (device, context) pairs from KNZ_DEVICES ("0,1,2,3": one lane per
// entry, a device may be named more than once) or, when that is not set, KNZ_LANES (default 4) lanes on the default device.
// setLaneDevices() overrides the environment for streams created afterwards (empty vector: back to the environment).
void setLaneDevices(const std::vector<int>& devices);
std::vector<int> laneDevices();
// The stream classes keep finished streams' staging and device buffers in process-wide pools (re-pinning 16 MiB costs milliseconds):
// at most 2 GiB of page-locked host memory and 1 GiB of device memory per lane context. An application that compresses once and then
// needs the memory calls this; buffers of live streams are not touched. Returns the bytes given back. (C callers: kanzi_api.h has no
// such call -- the reference has no pools; `KNZ_LANES=1` bounds what one GPU's lanes can hold.)
size_t releaseIdleBuffers();
knz_ctx* laneContext(int device, int index);
void setDefaultDevice(int device);

// ---- transforms on the device ------------------------------------------------------------------
class DeviceTransform : public Transform<byte> {
public:
    DeviceTransform(int type, Context* ctx);
    bool forward(SliceArray<byte>& src, SliceArray<byte>& dst, int length);
    bool inverse(SliceArray<byte>& src, SliceArray<byte>& dst, int length);
    int getMaxEncodedLength(int srcLen) const;
    int type() const { return _type; }
protected:
    int _type;
    int _entropy;     // stream entropy id from the context (RLT escape choice), -1 if absent
    int _bsVersion;   // "bsVersion" of the context (inverse of BWT / LZ / LZX blocks of streams older than version 6), default 6
};

class BWTBlockCodec : public DeviceTransform { public: explicit BWTBlockCodec(Context& ctx) : DeviceTransform(KNZ_T_BWT, &ctx) {} BWTBlockCodec() : DeviceTransform(KNZ_T_BWT, nullptr) {} };
class SBRT : public DeviceTransform {
public:
    static const int MODE_MTF = 1, MODE_RANK = 2, MODE_TIMESTAMP = 3;
    explicit SBRT(int mode);
    SBRT(int mode, Context& ctx);
};
class SRT : public DeviceTransform { public: explicit SRT(Context& ctx) : DeviceTransform(KNZ_T_SRT, &ctx) {} SRT() : DeviceTransform(KNZ_T_SRT, nullptr) {} };
class ZRLT : public DeviceTransform { public: explicit ZRLT(Context& ctx) : DeviceTransform(KNZ_T_ZRLT, &ctx) {} ZRLT() : DeviceTransform(KNZ_T_ZRLT, nullptr) {} };
class RLT : public DeviceTransform { public: explicit RLT(Context& ctx) : DeviceTransform(KNZ_T_RLT, &ctx) {} RLT() : DeviceTransform(KNZ_T_RLT, nullptr) {} };
// transform/LZCodec.hpp:27-52: "LZ" (16-bit hash) or "LZX" (19-bit hash, deeper look-ahead) chosen by the context's
// "lz" entry, which TransformFactory sets (TransformFactory.hpp:257-267). LZP has no device kernel.
class LZCodec : public DeviceTransform {
public:
    LZCodec() : DeviceTransform(KNZ_T_LZ, nullptr) {}
    explicit LZCodec(Context& ctx);
};
// ---- transforms on the host (kanzi-cpp_amd/host/text_codec.cpp): the stages of the level presets 5 and 6 that sit in front of the
// device chain. transform/TextCodec.hpp:140-215 (the encoding -- word indexes behind an escape byte or with the top bit set -- follows the
// context's "textcodec" entry, which TransformFactory derives from the entropy codec, TransformFactory.hpp:225-242), transform/UTFCodec.hpp:41-66.
// Both read and write the context's "dataType" like the reference's.
class TextCodec : public Transform<byte> {
public:
    TextCodec() : _ctx(nullptr), _variant(1), _blockSize(0), _bsVersion(6) {}
    explicit TextCodec(Context& ctx);
    bool forward(SliceArray<byte>& src, SliceArray<byte>& dst, int length);
    bool inverse(SliceArray<byte>& src, SliceArray<byte>& dst, int length);
    int getMaxEncodedLength(int n) const { return n; }
private:
    Context* _ctx;
    int _variant, _blockSize, _bsVersion;
};
class UTFCodec : public Transform<byte> {
public:
    UTFCodec() : _ctx(nullptr) {}
    explicit UTFCodec(Context& ctx) : _ctx(&ctx) {}
    bool forward(SliceArray<byte>& src, SliceArray<byte>& dst, int length);
    bool inverse(SliceArray<byte>& src, SliceArray<byte>& dst, int length);
    int getMaxEncodedLength(int n) const { return n + 8192; }
private:
    Context* _ctx;
};
class NullTransform : public Transform<byte> {
public:
    NullTransform() {}
    explicit NullTransform(Context&) {}
    bool forward(SliceArray<byte>& src, SliceArray<byte>& dst, int length) { return doCopy(src, dst, length); }
    bool inverse(SliceArray<byte>& src, SliceArray<byte>& dst, int length) { return doCopy(src, dst, length); }
    int getMaxEncodedLength(int n) const { return n; }
private:
    bool doCopy(SliceArray<byte>& src, SliceArray<byte>& dst, int length) const;
};

template <class T>
class TransformSequence : public Transform<T> {
public:
    TransformSequence(Transform<T>* transforms[8], bool deallocate);
    ~TransformSequence();
    bool forward(SliceArray<T>& src, SliceArray<T>& dst, int length);
    bool inverse(SliceArray<T>& src, SliceArray<T>& dst, int length);
    int getMaxEncodedLength(int srcLen) const;
    int getNbTransforms() const { return _length; }
    byte getSkipFlags() const { return _skipFlags; }
    void setSkipFlags(byte flags) { _skipFlags = flags; }
private:
    Transform<T>* _transforms[8];
    bool _deallocate;
    int _length;
    byte _skipFlags;
};

template <class T>
class TransformFactory {
public:
    enum TransformType { NONE_TYPE = 0, BWT_TYPE = 1, BWTS_TYPE = 2, LZ_TYPE = 3, SNAPPY_TYPE = 4, RLT_TYPE = 5, ZRLT_TYPE = 6,
                         MTFT_TYPE = 7, RANK_TYPE = 8, EXE_TYPE = 9, DICT_TYPE = 10, ROLZ_TYPE = 11, ROLZX_TYPE = 12, SRT_TYPE = 13,
                         LZP_TYPE = 14, MM_TYPE = 15, LZX_TYPE = 16, UTF_TYPE = 17, PACK_TYPE = 18, DNA_TYPE = 19 };
    static uint64 getType(const char* name);           // throws std::invalid_argument for unknown names / > 8 stages
    static uint64 getTypeToken(const char* name);
    static std::string getName(uint64 functionType);
    static TransformSequence<T>* newTransform(Context& ctx, uint64 functionType);   // throws for stages without a device kernel
    static const int ONE_SHIFT = 6;
    static const int MAX_SHIFT = (8 - 1) * ONE_SHIFT;
    static const int MASK = (1 << ONE_SHIFT) - 1;
};

// ---- entropy codecs on the device --------------------------------------------------------------
class DeviceEntropyEncoder : public EntropyEncoder {
public:
    DeviceEntropyEncoder(OutputBitStream& obs, int type) : _obs(obs), _type(type) {}
    int encode(const byte block[], uint blkptr, uint len);
    OutputBitStream& getBitStream() const { return _obs; }
    void dispose() {}
private:
    OutputBitStream& _obs;
    int _type;
};

class DeviceEntropyDecoder : public EntropyDecoder {
public:
    DeviceEntropyDecoder(InputBitStream& ibs, int type, int bsVersion = 6) : _ibs(ibs), _type(type), _bsVersion(bsVersion) {}
    int decode(byte block[], uint blkptr, uint len);
    InputBitStream& getBitStream() const { return _ibs; }
    void dispose() {}
private:
    InputBitStream& _ibs;
    int _type;
    int _bsVersion;
};

// Constructor parameters as in the reference (entropy/ANSRangeEncoder.hpp:48-51, ANSRangeDecoder.hpp:45-47,
// HuffmanEncoder.hpp:32, HuffmanDecoder.hpp:32): the same range checks with the same messages. The device kernels are
// built for the values the stream classes and the factories use (16 KiB chunks, scaled by 256 for order 1; logRange 12) --
// a valid value other than the default is refused with std::invalid_argument instead of being silently ignored.
// HuffmanDecoder takes "bsVersion" from its Context (chunk layout of versions below 6, HuffmanDecoder.cpp:349-352).
class ANSRangeEncoder : public DeviceEntropyEncoder {
public:
    static const int DEFAULT_ANS0_CHUNK_SIZE = 16384, DEFAULT_LOG_RANGE = 12, MIN_CHUNK_SIZE = 1024, MAX_CHUNK_SIZE = 1 << 27;
    ANSRangeEncoder(OutputBitStream& obs, int order = 0, int chunkSize = DEFAULT_ANS0_CHUNK_SIZE, int logRange = DEFAULT_LOG_RANGE);
};
class ANSRangeDecoder : public DeviceEntropyDecoder {
public:
    static const int DEFAULT_ANS0_CHUNK_SIZE = 16384, MIN_CHUNK_SIZE = 1024, MAX_CHUNK_SIZE = 1 << 27;
    ANSRangeDecoder(InputBitStream& ibs, int order = 0, int chunkSize = DEFAULT_ANS0_CHUNK_SIZE);
};
struct HuffmanCommon { static const int LOG_MAX_CHUNK_SIZE = 14, MAX_CHUNK_SIZE = 1 << 14; };
class HuffmanEncoder : public DeviceEntropyEncoder { public: HuffmanEncoder(OutputBitStream& obs, int chunkSize = HuffmanCommon::MAX_CHUNK_SIZE); };
class HuffmanDecoder : public DeviceEntropyDecoder { public: HuffmanDecoder(InputBitStream& ibs, Context* pCtx = nullptr, int chunkSize = HuffmanCommon::MAX_CHUNK_SIZE); };
class FPAQEncoder : public DeviceEntropyEncoder { public: explicit FPAQEncoder(OutputBitStream& obs) : DeviceEntropyEncoder(obs, KNZ_E_FPAQ) {} };
class FPAQDecoder : public DeviceEntropyDecoder { public: explicit FPAQDecoder(InputBitStream& ibs) : DeviceEntropyDecoder(ibs, KNZ_E_FPAQ) {} };
class NullEntropyEncoder : public DeviceEntropyEncoder { public: explicit NullEntropyEncoder(OutputBitStream& obs) : DeviceEntropyEncoder(obs, KNZ_E_NONE) {} };
class NullEntropyDecoder : public DeviceEntropyDecoder { public: explicit NullEntropyDecoder(InputBitStream& ibs) : DeviceEntropyDecoder(ibs, KNZ_E_NONE) {} };

class EntropyEncoderFactory {
public:
    static const short NONE_TYPE = 0, HUFFMAN_TYPE = 1, FPAQ_TYPE = 2, PAQ_TYPE = 3, RANGE_TYPE = 4, ANS0_TYPE = 5, CM_TYPE = 6,
                       TPAQ_TYPE = 7, ANS1_TYPE = 8, TPAQX_TYPE = 9;
    static EntropyEncoder* newEncoder(OutputBitStream& obs, Context& ctx, short entropyType);
    static const char* getName(short entropyType);
    static short getType(const char* name);
};

class EntropyDecoderFactory {
public:
    static EntropyDecoder* newDecoder(InputBitStream& ibs, Context& ctx, short entropyType);
    static const char* getName(short entropyType) { return EntropyEncoderFactory::getName(entropyType); }
    static short getType(const char* name) { return EntropyEncoderFactory::getType(name); }
};

// ---- block framing ------------------------------------------------------------------------------
// Compressed bytes fetched from the source and not yet handed to the device: a byte buffer that keeps its memory (page-locked,
// from the process-wide pool: a std::vector would zero-fill and page-fault every byte it grows by, which cost as much as reading
// the file) and always has 16 readable bytes behind its end.
class FetchBuf {
public:
    FetchBuf() : _p(nullptr), _n(0), _cap(0) {}
    ~FetchBuf();
    size_t size() const { return _n; }
    byte& operator[](size_t i) { return _p[i]; }
    const byte& operator[](size_t i) const { return _p[i]; }
    void resize(size_t n);                 // new bytes are NOT initialised
    void resize(size_t n, byte fill);      // new bytes are set to `fill`
    void dropFront(size_t n);
    void clear() { _n = 0; }
private:
    FetchBuf(const FetchBuf&);
    FetchBuf& operator=(const FetchBuf&);
    byte* _p; size_t _n, _cap;
    void reserve(size_t n);
};

typedef std::ostream OutputStream;     // src/types.hpp
typedef std::istream InputStream;

class CompressedOutputStream : public std::ostream {
public:
    // io/CompressedOutputStream.hpp:140-152 as the concurrent build declares it: the thread pool sits in front of `headerless`
    // (src/api/Compressor.cpp:230-237 passes nullptr there). Accepted and ignored: the blocks of a batch run on the device.
    CompressedOutputStream(std::ostream& os, int jobs = 1, const std::string& entropy = "NONE", const std::string& transform = "NONE",
                           int blockSize = 4 * 1024 * 1024, int checksum = 0, uint64 originalSize = 0, ThreadPool* pool = nullptr,
                           bool headerless = false);
    // the same without the pool: the reference's declaration when CONCURRENCY_ENABLED is not defined
    CompressedOutputStream(std::ostream& os, int jobs, const std::string& entropy, const std::string& transform,
                           int blockSize, int checksum, uint64 originalSize, bool headerless);
    // io/CompressedOutputStream.hpp:154, .cpp:148-240 (what app/BlockCompressor.cpp:757 calls): "jobs" (default 1), "blockSize",
    // "entropy", "transform", "checksum" (default 0), "fileSize" (default 0) from the context
    CompressedOutputStream(std::ostream& os, Context& ctx, bool headerless = false);
    ~CompressedOutputStream();
    // io/CompressedOutputStream.cpp:243-260: kept (a listener is registered once, removal reports whether it was there); the
    // device path raises no per-block events
    bool addListener(Listener<Event>& bl);
    bool removeListener(Listener<Event>& bl);
    std::ostream& write(const char* s, std::streamsize n);
    std::ostream& put(char c);
    // io/CompressedOutputStream.hpp:228-243
    std::ostream& flush() { return *this; }                                   // NOOP: the underlying stream flushes itself
    std::streampos tellp() { throw std::ios_base::failure("Not supported"); }
    std::ostream& seekp(std::streampos) { throw std::ios_base::failure("Not supported"); }
    void close();
    uint64 getWritten() const { return _written.load(); }
    // number of blocks handed to the device per call (default: jobs, like the reference keeps `jobs` blocks in flight)
    // (before the first write(): the page-locked staging slots are sized for the batch when the first byte arrives; one device
    // call takes at most 2 GiB of input)
    void setBatchBlocks(int n)
    {
        if (n <= 0 || _batchBytes != 0 || _hosted) return;
        const long long lim = (1ll << 31) / (long long)_blockSize - 1;
        _batchBlocks = (long long)n > lim ? int(lim < 1 ? 1 : lim) : n;
    }
private:
    size_t _batchBytes;           // batch size in bytes, latched by the first write()
    std::ostream& _os;
    int _jobs, _blockSize, _checksum;
    short _entropyType;
    int _hosted;                  // leading stages of the chain that run on the host (TEXT, UTF): blocks then go to the device one by one
    int _hostIds[8];
    uint64 _transformType;
    uint64 _inputSize;
    bool _headless, _closed, _headerDone;
    int _batchBlocks;
    int64 _blockId;               // blocks submitted so far
    byte _pendingByte;            // partial last byte of the stream written so far
    uint _pendingBits;
    std::atomic<uint64_t> _written;   // bytes that reached the sink
    // Lanes: one per entry of KNZ_DEVICES (default: four lanes on the default device). A lane owns a device context of its own
    // (stream, workspaces), a worker thread, a page-locked input slot with its device copy and a page-locked output buffer.
    // write() fills the lanes round robin, one batch each; every batch is compressed as an independent bit run (only the
    // first carries the stream header, block ids continue), so the lanes -- and the devices behind them -- work side by side.
    // Where a run starts in the stream is known once the runs in front of it have their lengths (published in batch order);
    // a run that does not start on a byte boundary is moved by 1..7 bits on its device before it comes back. The caller's
    // thread appends the runs in order (from write() / close()), OR-ing the byte two runs share: the reference's ordered
    // bit-granular append (io/CompressedOutputStream.cpp:835-868) with whole batches as the unit.
    struct Lane {
        std::vector<uint8_t> hostA, hostB;      // output of the host stages of a block (chains that start with TEXT / UTF)
        int device; knz_ctx* ctx; std::thread worker;
        byte* in; size_t inCap; size_t n; bool last; void* dIn; size_t dInCap; uint64 ticket;
        void* dOut; size_t dOutCap; void* dShift; size_t dShiftCap;
        byte* out; size_t outCap; size_t outBytes; uint shiftR; uint64 bits;
        int64 seq; int64 firstBlock;
        int state;                // 0 free (the caller may fill it), 1 queued / in the kernels, 2 compressed bytes wait for the sink
    };
    std::vector<Lane> _lanes;
    // The finished runs are appended to the sink by a thread of their own (KNZ_SINK_THREAD=0: by the caller's thread, between two
    // batches, as in round 4): the caller only fills staging slots, so the lanes are refilled while the sink is being written.
    // THREADING CONTRACT (differs from the reference, which writes to the sink from the caller's thread only): between the first
    // write() and the return of close() the std::ostream handed to the constructor is written from that thread -- the caller must not
    // touch it meanwhile (tellp() of this class is safe: it reads a counter). An exception the sink throws there (exceptions() set, a
    // streambuf that throws) is kept and rethrown, as it is, by the next write() / close() on the caller's thread.
    std::thread _sink;
    bool _sinkThread;
    bool _spreadCopies;           // staging copies over the helper threads (chains the device runs faster than one thread copies)
    void sinkLoop();
    int _fillLane;
    int64 _nextSeq, _sinkSeq, _pubSeq;    // batches handed out / appended to the sink / whose end position is known
    uint64 _cumBits;                      // end position (bits) of the batches published so far
    std::mutex _mu;
    std::condition_variable _cv;
    bool _stop;
    std::exception_ptr _err;
    std::vector<Listener<Event>*> _listeners;
    // where the wall time of a stream goes (nanoseconds; printed by close() when KNZ_HOST_TIMING is set): caller's thread
    // [0] copy into the staging slot [1] waiting for a free lane [2] writing to the sink; lane workers [3] upload wait [4] kernels
    // [5] waiting for the run's start position [6] bit shift + download
    std::atomic<uint64_t> _tns[8];
    std::chrono::steady_clock::time_point _t0;      // construction (KNZ_HOST_TIMING=2 prints a line per batch, times relative to it)
    void init(int jobs, const std::string& entropy, const std::string& transform, int blockSize, int checksum, uint64 originalSize, bool headerless);
    bool drainOne(std::unique_lock<std::mutex>& l);
    void enqueue(bool last);
    void workerLoop(int lane);
    void rethrow();
    void submit(Lane& ln);
};

class CompressedInputStream : public std::istream {
public:
    // io/CompressedInputStream.hpp:183-196 as the concurrent build declares it (src/api/Decompressor.cpp:159-166 passes nullptr for
    // the pool). Accepted and ignored.
    CompressedInputStream(std::istream& is, int jobs = 1, const std::string& entropy = "NONE", const std::string& transform = "NONE",
                          int blockSize = 4 * 1024 * 1024, int checksum = 0, uint64 originalSize = 0, ThreadPool* pool = nullptr,
                          bool headerless = false, int bsVersion = 6);
    // the same without the pool: the reference's declaration when CONCURRENCY_ENABLED is not defined
    CompressedInputStream(std::istream& is, int jobs, const std::string& entropy, const std::string& transform,
                          int blockSize, int checksum, uint64 originalSize, bool headerless, int bsVersion = 6);
    // io/CompressedInputStream.hpp:189 / .cpp:121-212: parameters from a Context -- "jobs", for a headerless stream "entropy",
    // "transform", "blockSize", "checksum", "outputSize", "bsVersion", and the block range "from" / "to" (1-based block ids, blocks
    // from <= id < to are decoded, the ones before are skipped on the host, io/CompressedInputStream.cpp:836-868)
    CompressedInputStream(std::istream& is, Context& ctx, bool headerless = false);
    ~CompressedInputStream();
    // io/CompressedInputStream.cpp:482-499 (see CompressedOutputStream::addListener)
    bool addListener(Listener<Event>& bl);
    bool removeListener(Listener<Event>& bl);
    // io/CompressedInputStream.hpp:306-327
    std::streampos tellg() { throw std::ios_base::failure("Not supported"); }
    std::istream& seekg(std::streampos) { throw std::ios_base::failure("Not supported"); }
    std::istream& putback(char) { setstate(std::ios::badbit); throw std::ios_base::failure("Not supported"); }
    std::istream& unget() { setstate(std::ios::badbit); throw std::ios_base::failure("Not supported"); }
    void setBlockRange(int from, int to) { _from = from < 1 ? 1 : from; _to = to; }
    std::istream& read(char* s, std::streamsize n);
    int get();
    int peek();
    std::streamsize gcount() const { return _gcount; }
    void close();
    uint64 getRead() const { return (_readBits + 7) >> 3; }
    // io/CompressedInputStream.hpp:227-229,329-384. The only valid positions are block boundaries. tell() is the bit
    // position (in the compressed stream) behind the batch of blocks read() is currently delivering (before the first read:
    // behind the stream header), hence always a valid argument for seek(); call setBatchBlocks(1) to step block by block.
    // seek() drops everything decoded and not yet read. A reader thread decodes the next batch while this one is consumed.
    bool seek(int64 bitPos);
    int64 tell();
    void setBatchBlocks(int n) { if (n > 0) { _batchBlocks = n; _batchFromEnv = true; } }
private:
    std::istream& _is;
    int _jobs, _blockSize, _checksum;
    short _entropyType;
    int _hosted;                  // see CompressedOutputStream
    int _hostIds[8];
    uint64 _transformType;
    uint64 _outputSize;
    bool _headless, _closed, _headerDone, _ended;
    int _bsVersion;                               // bitstream version of the stream being read (6 = current; 3..5: see readHeader)
    int _from, _to;               // block range (1-based ids), default everything
    int64 _nextBlockId;           // id of the next block the host walk will meet
    std::atomic<int> _batchBlocks;
    std::atomic<bool> _batchFromEnv;
    // owned by the reader thread once it runs
    FetchBuf _comp;               // compressed bytes fetched and not yet decoded
    bool _spreadCopies;           // see CompressedOutputStream (decided when the stream parameters are known)
    uint64 _compBit;              // next unread bit in _comp
    uint64 _consumedBits;
    int64 _originBit;             // bit position in the underlying stream that _compBit == 0 corresponds to
    bool _srcEof;
    // Three stages, two buffers between each pair. The reader thread fetches compressed bytes, walks the block length prefixes and
    // copies the batch to the device (page-locked staging, copy stream); the decoder thread runs the kernels and queues the
    // device-to-host copy of the result into a page-locked slot; read() waits for that copy and drains the slot. So the file reads
    // and both PCIe directions of neighbouring batches run beside the kernels.
    // One (Prep, PSlot, decoder thread) per lane -- an entry of KNZ_DEVICES, by default four lanes on the default device; the reader
    // hands the batches to the lanes round robin and read() takes them back in the same order, so consecutive batches are decoded
    // side by side on different contexts / devices.
    struct Prep { knz_ctx* ctx; void* dIn; size_t dInCap; byte* stage; size_t stageCap; size_t inBytes; uint64 startBit; int nb; bool last;
                  int64 endBit; uint64 consumedBits; std::exception_ptr err; uint64 ticket; int state; };                    // state: 0 free, 1 prepared
    std::vector<Prep> _prep;
    int _pprod;
    std::vector<int> _activeIdx;  // the lanes that get batches, in turn (per device bounded by the block size once it is known)
    struct PSlot { knz_ctx* ctx; byte* buf; size_t cap; size_t len; int64 endBit; uint64 consumedBits; bool last; std::exception_ptr err;
                   void* dOut; size_t dOutCap; uint64 ticket; int state; int device; };                                        // state: 0 free, 2 ready
    std::vector<PSlot> _ps;
    int _cons;
    std::thread _reader;
    std::vector<std::thread> _decoders;
    std::mutex _rmu;
    std::condition_variable _rcv;
    bool _rstop, _started;
    // owned by the caller's thread
    PSlot* _cur;                  // slot being delivered
    size_t _plainPos;
    bool _lastTaken;              // the final batch has been handed over: nothing more will come
    int64 _tellBit;
    uint64 _readBits;
    std::streamsize _gcount;
    std::vector<Listener<Event>*> _listeners;
    std::chrono::steady_clock::time_point _t0;      // see CompressedOutputStream
    // see CompressedOutputStream: reader thread [0] reading the source [1] prefix walk + staging copy; decoder threads [2] upload
    // wait [3] kernels; caller's thread [4] waiting for a decoded batch [5] download wait [6] copy into the caller's buffer
    std::atomic<uint64_t> _tns[8];
    void init(int jobs, const std::string& entropy, const std::string& transform, int blockSize, int checksum, uint64 originalSize, bool headerless,
              int bsVersion);
    void ensureStarted();
    void stopReader();
    void readerLoop();
    bool advance();
    void readHeader();
    bool fetch(size_t minBytes);
    void prepareBatch(Prep& pr);
    void decodeBatch(Prep& pr, PSlot& sl);
    void decoderLoop(int lane);
};

}  // namespace kanzi_amd
#endif
