// Host side of the drop-in: the reference's block-pipeline interfaces (include/kanzi_amd.hpp) and
// C API (include/kanzi_api.h) implemented over the device C ABI (include/knz_hip.h).
// Host code only splits, stages and appends; every transform / entropy / framing bit comes from
// the GPU. Reference semantics cited per function.
#include "kanzi_amd.hpp"
#include "kanzi_api.h"
#include "host_stages.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <sys/stat.h>
#include <chrono>
#include <deque>
#include <fcntl.h>
#include <functional>
#include <unistd.h>

namespace kanzi_amd {

// ------------------------------------------------------------------------------------------------
// device context
// ------------------------------------------------------------------------------------------------
static std::mutex g_devMutex;
static std::map<int, knz_ctx*> g_ctx;
static int g_defaultDevice = -1;

void setDefaultDevice(int device) { std::lock_guard<std::mutex> l(g_devMutex); g_defaultDevice = device; }

knz_ctx* deviceContext(int device)
{
    std::lock_guard<std::mutex> l(g_devMutex);
    if (device < 0) {
        if (g_defaultDevice < 0) {
            const char* e = getenv("KNZ_DEVICE");
            if (!e) e = getenv("LOCAL_RANK");
            g_defaultDevice = e ? atoi(e) : 0;
        }
        device = g_defaultDevice;
    }
    auto it = g_ctx.find(device);
    if (it != g_ctx.end()) return it->second;
    knz_ctx* c = nullptr;
    if (knz_hip_create(device, nullptr, &c) != 0 || c == nullptr)
        throw IOException("No usable GPU: the kanzi_amd block pipeline has no CPU fallback", Error::ERR_CREATE_CODEC);
    g_ctx[device] = c;
    return c;
}

// ---- lanes: (device, context) pairs the stream classes spread their batches over
static std::vector<int> g_laneOverride;
static bool g_laneOverrideSet = false;
static std::map<std::pair<int, int>, knz_ctx*> g_laneCtx;

void setLaneDevices(const std::vector<int>& devices)
{
    std::lock_guard<std::mutex> l(g_devMutex);
    g_laneOverride = devices;
    g_laneOverrideSet = !devices.empty();
}

std::vector<int> laneDevices()
{
    {
        std::lock_guard<std::mutex> l(g_devMutex);
        if (g_laneOverrideSet) return g_laneOverride;
    }
    std::vector<int> v;
    if (const char* e = getenv("KNZ_DEVICES")) {
        const char* p = e;
        while (*p) {
            while (*p == ',' || *p == ' ') p++;
            if (*p < '0' || *p > '9') break;
            v.push_back(int(strtol(p, const_cast<char**>(&p), 10)));
            if (v.size() >= 64) break;
        }
    }
    if (v.empty()) {
        int dev;
        {
            std::lock_guard<std::mutex> l(g_devMutex);
            if (g_defaultDevice < 0) {
                const char* e = getenv("KNZ_DEVICE");
                if (!e) e = getenv("LOCAL_RANK");
                g_defaultDevice = e ? atoi(e) : 0;
            }
            dev = g_defaultDevice;
        }
        int lanes = 6;                               // three in the kernels (DeviceGate), the others moving bytes
        if (const char* e = getenv("KNZ_LANES")) lanes = std::max(1, std::min(12, atoi(e)));
        v.assign(size_t(lanes), dev);
    }
    return v;
}

// context number `index` of a device (0 = the one the transform / entropy mirror classes use); kept for the life of the process
knz_ctx* laneContext(int device, int index)
{
    if (index == 0) return deviceContext(device);
    std::lock_guard<std::mutex> l(g_devMutex);
    auto key = std::make_pair(device, index);
    auto it = g_laneCtx.find(key);
    if (it != g_laneCtx.end()) return it->second;
    knz_ctx* c = nullptr;
    if (knz_hip_create(device, nullptr, &c) != 0 || c == nullptr)
        throw IOException("No usable GPU: the kanzi_amd block pipeline has no CPU fallback", Error::ERR_CREATE_CODEC);
    g_laneCtx[key] = c;
    return c;
}

// lane k of a stream -> its context (the k-th lane that names a device gets that device's context number k')
static std::vector<knz_ctx*> openLanes(const std::vector<int>& devs)
{
    std::vector<knz_ctx*> out;
    std::map<int, int> used;
    for (int d : devs) { const int idx = used[d]++; out.push_back(laneContext(d, idx)); }
    return out;
}

// KNZ_HOST_TIMING=1: the stream classes print where their wall time went when they are closed (developer aid)
namespace {
struct ScopedNs {
    std::atomic<uint64_t>& acc; std::chrono::steady_clock::time_point t0;
    explicit ScopedNs(std::atomic<uint64_t>& a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~ScopedNs() { acc += uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count()); }
};
bool hostTiming() { static const bool on = getenv("KNZ_HOST_TIMING") != nullptr && atoi(getenv("KNZ_HOST_TIMING")) != 0; return on; }
bool hostTimeline() { static const bool on = getenv("KNZ_HOST_TIMING") != nullptr && atoi(getenv("KNZ_HOST_TIMING")) >= 2; return on; }
double msSince(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

// A few helper threads for the byte moving the host layer does on somebody's critical path: the copies between the caller's memory
// and the page-locked staging slots, and the C API's file reads / writes (pread / pwrite of slices). KNZ_COPY_THREADS = threads a
// large copy is spread over, the calling thread included (default 4, 1 = everything on the calling thread). The pool lives as long
// as the process; a thread that waits for its slices runs queued slices of others meanwhile, so concurrent users never wait idle.
class HelperPool {
public:
    static HelperPool& get() { static HelperPool* p = new HelperPool(); return *p; }
    int width() const { return _width; }
    // fn(part) for part = 0 .. parts-1, part 0 on the calling thread; returns when all are done
    void run(int parts, const std::function<void(int)>& fn)
    {
        if (parts <= 1 || _width <= 1) { for (int i = 0; i < parts; i++) fn(i); return; }
        struct Job { std::atomic<int> left; };
        std::shared_ptr<Job> job = std::make_shared<Job>();
        job->left = parts - 1;
        {
            std::lock_guard<std::mutex> l(_mu);
            for (int i = 1; i < parts; i++) _q.push_back([job, i, &fn] { fn(i); job->left.fetch_sub(1, std::memory_order_acq_rel); });
        }
        _cv.notify_all();
        fn(0);
        while (job->left.load(std::memory_order_acquire) != 0) {
            std::function<void()> t;
            {
                std::lock_guard<std::mutex> l(_mu);
                if (!_q.empty()) { t = std::move(_q.front()); _q.pop_front(); }
            }
            if (t) t(); else std::this_thread::yield();
        }
    }
private:
    HelperPool()
    {
        const char* e = getenv("KNZ_COPY_THREADS");
        _width = e ? std::max(1, std::min(16, atoi(e))) : 4;
        for (int i = 1; i < _width; i++) std::thread([this] { loop(); }).detach();
    }
    void loop()
    {
        for (;;) {
            std::function<void()> t;
            {
                std::unique_lock<std::mutex> l(_mu);
                _cv.wait(l, [&] { return !_q.empty(); });
                t = std::move(_q.front());
                _q.pop_front();
            }
            t();
        }
    }
    int _width;
    std::mutex _mu;
    std::condition_variable _cv;
    std::deque<std::function<void()>> _q;
};

const size_t PAR_MIN = size_t(2) << 20;          // below this a copy is not worth waking anybody for

// how many slices n bytes are cut into (at least 1 MiB each)
int parParts(size_t n)
{
    if (n < PAR_MIN) return 1;
    return int(std::min<size_t>(size_t(HelperPool::get().width()), n >> 20));
}

void parCopy(void* dst, const void* src, size_t n, bool spread = true)
{
    const int parts = spread ? parParts(n) : 1;
    if (parts <= 1) { memcpy(dst, src, n); return; }
    const size_t slice = ((n / size_t(parts)) + 4095) & ~size_t(4095);
    HelperPool::get().run(parts, [&](int i) {
        const size_t a = std::min(n, size_t(i) * slice), b = (i == parts - 1) ? n : std::min(n, a + slice);
        if (b > a) memcpy(static_cast<char*>(dst) + a, static_cast<const char*>(src) + a, b - a);
    });
}
}

static void devCheck(knz_ctx* c, int rc, const char* what)
{
    if (rc == 0) return;
    std::stringstream ss;
    ss << what << ": " << knz_hip_last_error(c);
    if (rc > 0) throw IOException(ss.str(), rc);
    throw std::runtime_error(ss.str());
}

// ------------------------------------------------------------------------------------------------
// bit streams
// ------------------------------------------------------------------------------------------------
DefaultOutputBitStream::DefaultOutputBitStream(std::ostream& os, uint bufferSize)
    : _os(os), _current(0), _avail(64), _written(0), _closed(false)
{
    if (bufferSize < 1024) throw std::invalid_argument("Invalid buffer size (must be at least 1024)");
    if (bufferSize > (1u << 29)) throw std::invalid_argument("Invalid buffer size (must be at most 536870912)");
    if ((bufferSize & 7) != 0) throw std::invalid_argument("Invalid buffer size (must be a multiple of 8)");
    _buf.reserve(bufferSize);
}

DefaultOutputBitStream::~DefaultOutputBitStream() { try { close(); } catch (...) {} }

void DefaultOutputBitStream::flush()
{
    if (!_buf.empty()) {
        _os.write(reinterpret_cast<const char*>(_buf.data()), std::streamsize(_buf.size()));
        if (_os.fail()) throw BitStreamException("Write to bitstream failed", BitStreamException::INPUT_OUTPUT);
        _buf.clear();
    }
}

void DefaultOutputBitStream::push()
{
    for (int s = 56; s >= 0; s -= 8) _buf.push_back(byte(_current >> s));
    _current = 0;
    _avail = 64;
    if (_buf.size() + 8 >= _buf.capacity()) flush();
}

void DefaultOutputBitStream::writeBit(int bit)
{
    if (_closed) throw BitStreamException("Stream closed", BitStreamException::STREAM_CLOSED);
    _avail--;
    _current |= (uint64(bit & 1) << _avail);
    _written++;
    if (_avail == 0) push();
}

uint DefaultOutputBitStream::writeBits(uint64 value, uint count)
{
    if ((count == 0) || (count > 64)) return 0;
    if (_closed) throw BitStreamException("Stream closed", BitStreamException::STREAM_CLOSED);
    if (count < 64) value &= ((uint64(1) << count) - 1);
    _written += count;
    if (count < _avail) {
        _avail -= count;
        _current |= (value << _avail);
    } else {
        const uint remaining = count - _avail;
        _current |= (remaining == 64 ? 0 : (value >> remaining));
        push();
        if (remaining != 0) {
            _avail -= remaining;
            _current = value << _avail;
        }
    }
    return count;
}

uint DefaultOutputBitStream::writeBits(const byte bits[], uint count)
{
    if (_closed) throw BitStreamException("Stream closed", BitStreamException::STREAM_CLOSED);
    uint remaining = count, start = 0;
    if ((_avail & 7) == 0) {
        // byte aligned: drain the accumulator and append whole bytes directly
        while ((_avail != 64) && (remaining >= 8)) { writeBits(uint64(bits[start]), 8); start++; remaining -= 8; }
        const uint r = remaining >> 3;
        if (_avail == 64 && r > 0) {
            flush();
            _os.write(reinterpret_cast<const char*>(&bits[start]), std::streamsize(r));
            if (_os.fail()) throw BitStreamException("Write to bitstream failed", BitStreamException::INPUT_OUTPUT);
            _written += uint64(r) << 3;
            start += r;
            remaining -= (r << 3);
        }
    }
    while (remaining >= 8) { writeBits(uint64(bits[start]), 8); start++; remaining -= 8; }
    if (remaining > 0) writeBits(uint64(bits[start]) >> (8 - remaining), remaining);
    return count;
}

void DefaultOutputBitStream::close()
{
    if (_closed) return;
    // push the last bytes; the very last one may be incomplete (zero padded), written() stays exact
    for (int s = 56; _avail < 64; s -= 8, _avail += 8) _buf.push_back(byte(_current >> s));
    _avail = 64;
    _current = 0;
    flush();
    _os.flush();
    _closed = true;
}

DefaultInputBitStream::DefaultInputBitStream(std::istream& is, uint bufferSize)
    : _is(is), _pos(0), _read(0), _closed(false), _eof(false), _chunk(bufferSize)
{
    if (bufferSize < 1024) throw std::invalid_argument("Invalid buffer size (must be at least 1024)");
    if ((bufferSize & 7) != 0) throw std::invalid_argument("Invalid buffer size (must be a multiple of 8)");
}

DefaultInputBitStream::~DefaultInputBitStream() {}

bool DefaultInputBitStream::fill(uint64 needBits)
{
    while (!_eof && uint64(_data.size()) * 8 < _pos + needBits) {
        // drop fully consumed bytes from the front now and then
        if (_pos >= (uint64(1) << 26)) { const size_t drop = size_t(_pos >> 3); _data.erase(_data.begin(), _data.begin() + drop); _pos &= 7; }
        const size_t old = _data.size();
        _data.resize(old + _chunk);
        _is.read(reinterpret_cast<char*>(&_data[old]), std::streamsize(_chunk));
        const size_t got = size_t(_is.gcount());
        _data.resize(old + got);
        if (got < _chunk) _eof = true;
    }
    return uint64(_data.size()) * 8 >= _pos + needBits;
}

int DefaultInputBitStream::readBit() { return int(readBits(1)); }

uint64 DefaultInputBitStream::readBits(uint count)
{
    if (_closed) throw BitStreamException("Stream closed", BitStreamException::STREAM_CLOSED);
    if ((count == 0) || (count > 64)) throw BitStreamException("Invalid bit count", BitStreamException::INVALID_STREAM);
    if (!fill(count)) throw BitStreamException("No more data to read in the bitstream", BitStreamException::END_OF_STREAM);
    uint64 v = 0;
    uint left = count;
    while (left > 0) {
        const size_t b = size_t(_pos >> 3);
        const uint used = uint(_pos & 7), room = 8 - used;
        const uint take = left < room ? left : room;
        v = (v << take) | ((_data[b] >> (room - take)) & ((1u << take) - 1u));
        _pos += take;
        left -= take;
    }
    _read += count;
    return v;
}

uint DefaultInputBitStream::readBits(byte bits[], uint count)
{
    uint remaining = count, start = 0;
    while (remaining >= 8) { bits[start++] = byte(readBits(8)); remaining -= 8; }
    if (remaining > 0) bits[start] = byte(readBits(remaining) << (8 - remaining));
    return count;
}

void DefaultInputBitStream::close() { _closed = true; }

bool DefaultInputBitStream::hasMoreToRead() { return !_closed && fill(1); }

void DefaultInputBitStream::peekRemaining(const byte** data, uint64* startBit, uint64* endBit)
{
    while (!_eof) fill((uint64(_data.size()) * 8 - _pos) + uint64(_chunk) * 8);
    *data = _data.data();
    *startBit = _pos;
    *endBit = uint64(_data.size()) * 8;
}

void DefaultInputBitStream::peekAhead(uint64 wantBits, const byte** data, uint64* startBit, uint64* endBit)
{
    fill(wantBits);
    *data = _data.data();
    *startBit = _pos;
    const uint64 have = uint64(_data.size()) * 8;
    *endBit = (have - _pos > wantBits) ? _pos + wantBits : have;
}

void DefaultInputBitStream::skip(uint64 nbits)
{
    if (!fill(nbits)) throw BitStreamException("No more data to read in the bitstream", BitStreamException::END_OF_STREAM);
    _pos += nbits;
    _read += nbits;
}

// ------------------------------------------------------------------------------------------------
// transforms
// ------------------------------------------------------------------------------------------------
static int entropyIdFromName(const std::string& nm)
{
    std::string s = nm;
    std::transform(s.begin(), s.end(), s.begin(), ::toupper);
    if (s == "NONE") return 0; if (s == "HUFFMAN") return 1; if (s == "FPAQ") return 2; if (s == "RANGE") return 4;
    if (s == "ANS0") return 5; if (s == "CM") return 6; if (s == "TPAQ") return 7; if (s == "ANS1") return 8; if (s == "TPAQX") return 9;
    return -1;
}

DeviceTransform::DeviceTransform(int type, Context* ctx) : _type(type), _entropy(-1), _bsVersion(6)
{
    if (ctx != nullptr && ctx->has("entropy")) _entropy = entropyIdFromName(ctx->getString("entropy"));
    if (ctx != nullptr) _bsVersion = ctx->getInt("bsVersion", 6);       // BWTBlockCodec.cpp:34-36, LZCodec.cpp:101-104
    deviceContext();        // fail early (and loudly) when there is no GPU
}

int DeviceTransform::getMaxEncodedLength(int n) const
{
    switch (_type) {
    case KNZ_T_BWT: return n + 33;                      // BWTBlockCodec.hpp:47-50
    case KNZ_T_SRT: return n + 1024;                    // SRT.hpp:38
    case KNZ_T_RLT: return (n <= 512) ? n + 32 : n;     // RLT.hpp:43
    case KNZ_T_LZ: case KNZ_T_LZX: return ((n <= 1024) ? n + 16 : n + (n / 64)) + 2;    // LZCodec.hpp:91-95
    default: return n;
    }
}

bool DeviceTransform::forward(SliceArray<byte>& src, SliceArray<byte>& dst, int length)
{
    if (length == 0) return true;
    if (!SliceArray<byte>::isValid(src)) throw std::invalid_argument("Invalid input block");
    if (!SliceArray<byte>::isValid(dst)) throw std::invalid_argument("Invalid output block");
    if ((length < 0) || (length > src._length - src._index)) return false;
    if (src._array == dst._array) return false;
    knz_ctx* c = deviceContext();
    int32_t outLen = 0, ok = 0;
    devCheck(c, knz_hip_transform_forward(c, _type, src._array + src._index, length, dst._array + dst._index,
                                          dst._length - dst._index, _entropy, &outLen, &ok), "transform forward");
    if (!ok) return false;
    src._index += length;
    dst._index += outLen;
    return true;
}

bool DeviceTransform::inverse(SliceArray<byte>& src, SliceArray<byte>& dst, int length)
{
    if (length == 0) return true;
    if (!SliceArray<byte>::isValid(src)) throw std::invalid_argument("Invalid input block");
    if (!SliceArray<byte>::isValid(dst)) throw std::invalid_argument("Invalid output block");
    if ((length < 0) || (length > src._length - src._index)) return false;
    if (src._array == dst._array) return false;
    // LZCodec.cpp:486-490: readLength() may look two bytes past the block, which therefore must exist
    if ((_type == KNZ_T_LZ || _type == KNZ_T_LZX) && (length > src._length - src._index - 2)) return false;
    knz_ctx* c = deviceContext();
    int32_t outLen = 0, ok = 0;
    devCheck(c, knz_hip_transform_inverse_v(c, _type, _bsVersion == 0 ? 1 : _bsVersion, src._array + src._index, length, dst._array + dst._index,
                                            dst._length - dst._index, &outLen, &ok), "transform inverse");
    if (!ok) return false;
    src._index += length;
    dst._index += outLen;
    return true;
}

static int lzTypeFromContext(Context& ctx)
{
    const int t = ctx.getInt("lz", 3);
    if (t == 14) throw std::invalid_argument("LZP has no device kernel (out of scope of the accelerated block pipeline)");
    return (t == 16) ? KNZ_T_LZX : KNZ_T_LZ;
}

LZCodec::LZCodec(Context& ctx) : DeviceTransform(lzTypeFromContext(ctx), &ctx) {}

static int sbrtType(int mode)
{
    if ((mode != SBRT::MODE_MTF) && (mode != SBRT::MODE_RANK) && (mode != SBRT::MODE_TIMESTAMP)) throw std::invalid_argument("Invalid mode parameter");
    return (mode == SBRT::MODE_MTF) ? KNZ_T_MTFT : ((mode == SBRT::MODE_RANK) ? KNZ_T_RANK : KNZ_T_TIMESTAMP);
}

SBRT::SBRT(int mode) : DeviceTransform(sbrtType(mode), nullptr) {}

SBRT::SBRT(int mode, Context& ctx) : DeviceTransform(sbrtType(mode), &ctx) {}

bool NullTransform::doCopy(SliceArray<byte>& input, SliceArray<byte>& output, int length) const
{
    if (length == 0) return true;
    if (!SliceArray<byte>::isValid(input)) throw std::invalid_argument("Invalid input block");
    if (!SliceArray<byte>::isValid(output)) throw std::invalid_argument("Invalid output block");
    if (input._index + length > input._length) return false;
    if (output._index + length > output._length) return false;
    memmove(&output._array[output._index], &input._array[input._index], size_t(length));
    input._index += length;
    output._index += length;
    return true;
}

// ---- TransformSequence (transform/TransformSequence.hpp:58-265) ---------------------------------
template <class T>
TransformSequence<T>::TransformSequence(Transform<T>* transforms[8], bool deallocate)
{
    _deallocate = deallocate;
    _length = 8;
    _skipFlags = 0;
    for (int i = 7; i >= 0; i--) {
        _transforms[i] = transforms[i];
        if (_transforms[i] == nullptr) _length = i;
    }
    if (_length == 0) throw std::invalid_argument("At least one transform required");
}

template <class T>
TransformSequence<T>::~TransformSequence()
{
    if (_deallocate) for (int i = 0; i < 8; i++) delete _transforms[i];
}

template <class T>
int TransformSequence<T>::getMaxEncodedLength(int srcLength) const
{
    int req = srcLength;
    for (int i = 0; i < _length; i++) {
        if (_transforms[i] == nullptr) continue;
        const int nx = _transforms[i]->getMaxEncodedLength(req);
        if (nx > req) req = nx;
    }
    return req;
}

// The data of a block travels through the chain between two views: a stage reads the one that holds the data and writes the other, and
// the two change roles behind every stage that applied. The views start out as the caller's input and output; one of them that is too
// small for what a stage may write is replaced by memory of this object (the reference reallocates the caller's SliceArray with
// delete[] / new[] there, transform/TransformSequence.hpp:112-121; this mirror never frees an array it did not allocate, so the
// result is copied into the caller's output at the end wherever it ended up).
namespace {
template <class T>
class Relay {
public:
    Relay(SliceArray<T>& input, SliceArray<T>& output) : _data(&input), _spare(&output), _callerIn(&input), _callerOut(&output), _own(nullptr, 0, 0) {}

    // the view the next stage writes can take `need` elements
    void spareHolds(int need)
    {
        if (_spare->_length >= need) return;
        if (_spare == _callerIn || _spare == _callerOut) _spare = &_own;
        if (_own._length < need) { _mem.resize(size_t(need)); _own._array = _mem.data(); _own._length = need; }
    }

    // one stage over `count` elements: true when it applied (count becomes what it wrote, the views change roles); the indexes of both
    // views are where they were either way
    template <class Stage>
    bool run(Stage&& stage, int& count)
    {
        const int at = _data->_index, to = _spare->_index;
        const bool applied = stage(*_data, *_spare, count);
        if (applied) count = _spare->_index - to;
        _data->_index = at;
        _spare->_index = to;
        if (applied) std::swap(_data, _spare);
        return applied;
    }

    // the block, `count` elements, into the caller's output (false: it does not fit)
    bool deliver(int count) const
    {
        if (_data == _callerOut) return true;
        if ((count > _callerOut->_length - _callerOut->_index) || (count > _data->_length - _data->_index)) return false;
        memmove(&_callerOut->_array[_callerOut->_index], &_data->_array[_data->_index], size_t(count) * sizeof(T));
        return true;
    }

private:
    SliceArray<T>* _data;
    SliceArray<T>* _spare;
    SliceArray<T>* const _callerIn;
    SliceArray<T>* const _callerOut;
    SliceArray<T> _own;
    std::vector<T> _mem;
};
}

// transform/TransformSequence.hpp:88-162: every stage on the output of the one before it; a stage that declines is skipped and its bit
// stays set in the skip flags; "all skipped" (0xFF) means the block is stored as it is
template <class T>
bool TransformSequence<T>::forward(SliceArray<T>& input, SliceArray<T>& output, int count)
{
    if (!SliceArray<T>::isValid(input)) throw std::invalid_argument("Invalid input block");
    if (!SliceArray<T>::isValid(output)) throw std::invalid_argument("Invalid output block");
    if ((count < 0) || (count > input._length - input._index)) return false;
    _skipFlags = 0xFF;
    if (count == 0) return true;
    const int taken = count;
    const int worst = getMaxEncodedLength(taken);
    Relay<T> relay(input, output);
    for (int i = 0; i < _length; i++) {
        Transform<T>* stage = _transforms[i];
        if (stage == nullptr) continue;
        relay.spareHolds(worst);
        if (relay.run([stage](SliceArray<T>& src, SliceArray<T>& dst, int n) { return stage->forward(src, dst, n); }, count))
            _skipFlags &= byte(~(1 << (7 - i)));
    }
    if (!relay.deliver(count)) _skipFlags = 0xFF;
    input._index += taken;
    output._index += count;
    return _skipFlags != 0xFF;
}

// transform/TransformSequence.hpp:165-235: the stages whose bit is clear, last one first; the first failure ends the block
template <class T>
bool TransformSequence<T>::inverse(SliceArray<T>& input, SliceArray<T>& output, int count)
{
    if (!SliceArray<T>::isValid(input)) throw std::invalid_argument("Invalid input block");
    if (!SliceArray<T>::isValid(output)) throw std::invalid_argument("Invalid output block");
    if ((count < 0) || (count > input._length - input._index)) return false;
    if (count == 0) return true;
    if (count > output._length - output._index) return false;
    const int taken = count;
    bool good = true;
    Relay<T> relay(input, output);
    if (_skipFlags != 0xFF) {
        for (int i = _length - 1; i >= 0 && good; i--) {
            Transform<T>* stage = _transforms[i];
            if (stage == nullptr || (_skipFlags & byte(1 << (7 - i))) != 0) continue;
            relay.spareHolds(output._length);
            good = relay.run([stage](SliceArray<T>& src, SliceArray<T>& dst, int n) { return stage->inverse(src, dst, n); }, count);
        }
    }
    if (good) good = relay.deliver(count);
    input._index += taken;
    output._index += count;
    return good;
}

template class TransformSequence<byte>;

// ---- host transforms (text_codec.cpp) ----------------------------------------------------------------------
TextCodec::TextCodec(Context& ctx) : _ctx(&ctx)
{
    _variant = ctx.getInt("textcodec", 1);
    _blockSize = ctx.getInt("blockSize", 0);
    _bsVersion = ctx.getInt("bsVersion", 6);
}

bool TextCodec::forward(SliceArray<byte>& src, SliceArray<byte>& dst, int length)
{
    if (length == 0) return true;
    if (!SliceArray<byte>::isValid(src) || !SliceArray<byte>::isValid(dst)) throw std::invalid_argument("TextCodec: Invalid block");
    if (src._array == dst._array) return false;
    int dt = _ctx ? _ctx->getInt("dataType", hoststage::DT_UNDEFINED) : hoststage::DT_UNDEFINED;
    int outLen = 0;
    const bool ok = hoststage::textForward(_variant, reinterpret_cast<const uint8_t*>(&src._array[src._index]), length, reinterpret_cast<uint8_t*>(&dst._array[dst._index]),
                                           dst._length - dst._index, _blockSize, _bsVersion, &dt, &outLen);
    if (_ctx && length >= 1024) _ctx->putInt("dataType", dt);
    if (!ok) return false;
    src._index += length; dst._index += outLen;
    return true;
}

bool TextCodec::inverse(SliceArray<byte>& src, SliceArray<byte>& dst, int length)
{
    if (length == 0) return true;
    if (!SliceArray<byte>::isValid(src) || !SliceArray<byte>::isValid(dst)) throw std::invalid_argument("TextCodec: Invalid block");
    if (src._array == dst._array || src._index + length > src._length) return false;
    int outLen = 0;
    const bool ok = hoststage::textInverse(_variant, reinterpret_cast<const uint8_t*>(&src._array[src._index]), length, reinterpret_cast<uint8_t*>(&dst._array[dst._index]),
                                           dst._length - dst._index, _blockSize, _bsVersion, &outLen);
    src._index += length; dst._index += outLen;
    return ok;
}

bool UTFCodec::forward(SliceArray<byte>& src, SliceArray<byte>& dst, int length)
{
    if (length == 0) return true;
    if (!SliceArray<byte>::isValid(src) || !SliceArray<byte>::isValid(dst)) throw std::invalid_argument("UTFCodec: Invalid block");
    if (dst._length - dst._index < getMaxEncodedLength(length)) return false;
    int dt = _ctx ? _ctx->getInt("dataType", hoststage::DT_UNDEFINED) : hoststage::DT_UNDEFINED;
    int outLen = 0;
    const bool ok = hoststage::utfForward(reinterpret_cast<const uint8_t*>(&src._array[src._index]), length, reinterpret_cast<uint8_t*>(&dst._array[dst._index]),
                                          dst._length - dst._index, &dt, &outLen);
    if (_ctx) _ctx->putInt("dataType", dt);
    if (!ok) return false;
    src._index += length; dst._index += outLen;
    return true;
}

bool UTFCodec::inverse(SliceArray<byte>& src, SliceArray<byte>& dst, int length)
{
    if (length == 0) return true;
    if (!SliceArray<byte>::isValid(src) || !SliceArray<byte>::isValid(dst)) throw std::invalid_argument("UTFCodec: Invalid block");
    if (src._index + length > src._length) return false;
    int outLen = 0;
    const bool ok = hoststage::utfInverse(reinterpret_cast<const uint8_t*>(&src._array[src._index]), length, reinterpret_cast<uint8_t*>(&dst._array[dst._index]),
                                          dst._length - dst._index, &outLen);
    src._index += length; dst._index += outLen;
    return ok;
}

// Leading stages of a chain that run on the host. TEXT / UTF anywhere else in a chain has no place to run: refused.
// Chains the device runs far faster than one host thread copies memory (entropy coders alone, the byte transforms: tens of GB/s):
// the staging copies of such a stream are spread over the helper threads. With a sorting or matching stage in the chain (BWT, LZ,
// LZX; SRT / RANK are chains per block) the device is what a batch waits for and the extra threads only got in the way (measured,
// config 3 end to end: 2.94-3.14 GB/s with the copies on the caller's thread, 2.72-2.94 with four threads; config 2: 3.1 -> 3.9).
static bool chainIsHostBound(uint64 ttype)
{
    for (int i = 0; i < 8; i++) {
        const int t = int((ttype >> (42 - 6 * i)) & 63);
        if (t == KNZ_T_BWT || t == KNZ_T_LZ || t == KNZ_T_LZX || t == KNZ_T_SRT || t == KNZ_T_RANK || t == KNZ_T_TEXT || t == KNZ_T_UTF) return false;
    }
    return true;
}

static int hostedStagesOf(uint64 ttype, int ids[8])
{
    int n = 0, k = 0;
    bool device = false;
    for (int i = 0; i < 8; i++) {
        const int t = int((ttype >> (42 - 6 * i)) & 63);
        if (t == 0) continue;
        ids[k++] = t;
        const bool host = (t == KNZ_T_TEXT || t == KNZ_T_UTF);
        if (host && device) throw std::invalid_argument("TEXT / UTF behind a device transform is not supported");
        if (host) n++; else device = true;
    }
    return n;
}

// XXHash32 / XXHash64 of a block as the reference computes them (util/XXHash.hpp:61-115, :153-230; seed 0x4B414E5A; the 64-bit variant
// merges its accumulators with 32-bit style shifts). The device has the same in csrc/xxhash.hip; blocks that pass through host stages
// are hashed here, before the stages run.
static uint64 hostChecksum(const uint8_t* d, int length, int bits)
{
    auto ld32 = [](const uint8_t* p) { return uint32_t(p[0]) | (uint32_t(p[1]) << 8) | (uint32_t(p[2]) << 16) | (uint32_t(p[3]) << 24); };
    auto ld64 = [&](const uint8_t* p) { return uint64(ld32(p)) | (uint64(ld32(p + 4)) << 32); };
    const uint32_t SEED = 0x4B414E5Au;
    if (bits == 32) {
        const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
        uint32_t h;
        int i = 0;
        if (length >= 16) {
            uint32_t v[4] = { SEED + P1 + P2, SEED + P2, SEED, SEED - P1 };
            const int end16 = length - 16;
            do {
                for (int j = 0; j < 4; j++) { v[j] += ld32(d + i + 4 * j) * P2; v[j] = ((v[j] << 13) | (v[j] >> 19)) * P1; }
                i += 16;
            } while (i <= end16);
            h = ((v[0] << 1) | (v[0] >> 31)) + ((v[1] << 7) | (v[1] >> 25)) + ((v[2] << 12) | (v[2] >> 20)) + ((v[3] << 18) | (v[3] >> 14));
        } else h = SEED + P5;
        h += uint32_t(length);
        while (i <= length - 4) { h += ld32(d + i) * P3; h = ((h << 17) | (h >> 15)) * P4; i += 4; }
        while (i < length) { h += uint32_t(d[i]) * P5; h = ((h << 11) | (h >> 21)) * P1; i++; }
        h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3;
        return uint64(h ^ (h >> 16));
    }
    const uint64 P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull, P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
    auto round = [&](uint64 acc, uint64 val) { acc += val * P2; return ((acc << 31) | (acc >> 33)) * P1; };
    const uint64 seed = uint64(int64(int32_t(SEED)));
    uint64 h;
    int i = 0;
    if (length >= 32) {
        uint64 v[4] = { seed + P1 + P2, seed + P2, seed, seed - P1 };
        const int end32 = length - 32;
        do {
            for (int j = 0; j < 4; j++) v[j] = round(v[j], ld64(d + i + 8 * j));
            i += 32;
        } while (i <= end32);
        h = ((v[0] << 1) | (v[0] >> 31)) + ((v[1] << 7) | (v[1] >> 25)) + ((v[2] << 12) | (v[2] >> 20)) + ((v[3] << 18) | (v[3] >> 14));
        for (int j = 0; j < 4; j++) h = (h ^ round(0, v[j])) * P1 + P4;
    } else h = seed + P5;
    h += uint64(int64(length));
    while (i + 8 <= length) { h ^= round(0, ld64(d + i)); h = ((h << 27) | (h >> 37)) * P1 + P4; i += 8; }
    while (i + 4 <= length) { h ^= uint64(ld32(d + i)) * P1; h = ((h << 23) | (h >> 41)) * P2 + P3; i += 4; }
    while (i < length) { h ^= uint64(d[i]) * P5; h = ((h << 11) | (h >> 53)) * P1; i++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3;
    return h ^ (h >> 32);
}

static int textVariantOfEntropy(int etype) { return (etype == 0 || etype == 1 || etype == 4 || etype == 5) ? 2 : 1; }      // NONE, HUFFMAN, RANGE, ANS0

// the host stages of one block, in chain order: data ends up in `a` or `b`; returns the buffer that holds it
struct HostedResult { const uint8_t* data; int len; uint32_t applied; };
static HostedResult runHostStages(const int* ids, int count, const uint8_t* block, int n, int blockSize, int etype, std::vector<uint8_t>& a, std::vector<uint8_t>& b)
{
    HostedResult r{ block, n, 0u };
    if (n <= 15) return r;                                  // copy block: no stage runs (io/CompressedOutputStream.cpp:691-695)
    int dt = hoststage::presetDataType(block, n);
    std::vector<uint8_t>* bufs[2] = { &a, &b };
    int cur = 0;
    for (int i = 0; i < count; i++) {
        std::vector<uint8_t>& out = *bufs[cur];
        const size_t need = size_t(r.len) + 8192 + 64;
        if (out.size() < need) out.resize(need);
        int outLen = 0;
        bool ok;
        if (ids[i] == KNZ_T_TEXT) ok = hoststage::textForward(textVariantOfEntropy(etype), r.data, r.len, out.data(), r.len, blockSize, 6, &dt, &outLen);
        else ok = hoststage::utfForward(r.data, r.len, out.data(), r.len + 8192, &dt, &outLen);
        if (!ok) continue;
        r.data = out.data(); r.len = outLen; r.applied |= 1u << i;
        cur ^= 1;
    }
    return r;
}

// ---- TransformFactory (transform/TransformFactory.hpp:100-308) ----------------------------------
static const struct { const char* name; int type; } TNAMES[] = {
    {"NONE", 0}, {"BWT", 1}, {"BWTS", 2}, {"LZ", 3}, {"RLT", 5}, {"ZRLT", 6}, {"MTFT", 7}, {"RANK", 8}, {"EXE", 9}, {"TEXT", 10},
    {"ROLZ", 11}, {"ROLZX", 12}, {"SRT", 13}, {"LZP", 14}, {"MM", 15}, {"LZX", 16}, {"UTF", 17}, {"PACK", 18}, {"DNA", 19} };

template <class T>
uint64 TransformFactory<T>::getTypeToken(const char* tName)
{
    std::string name(tName);
    std::transform(name.begin(), name.end(), name.begin(), ::toupper);
    for (auto& e : TNAMES) if (name == e.name) return uint64(e.type);
    throw std::invalid_argument("Unknown transform type: '" + name + "'");
}

template <class T>
uint64 TransformFactory<T>::getType(const char* tName)
{
    std::string name(tName);
    size_t pos = name.find('+');
    if (pos == std::string::npos) return getTypeToken(name.c_str()) << MAX_SHIFT;
    size_t prv = 0;
    int n = 0;
    uint64 res = 0;
    int shift = MAX_SHIFT;
    name += '+';
    while (pos != std::string::npos) {
        if (++n > 8) throw std::invalid_argument("Only 8 transforms allowed: " + name);
        const std::string token = name.substr(prv, pos - prv);
        const uint64 typeTk = getTypeToken(token.c_str());
        if (typeTk != NONE_TYPE) { res |= (typeTk << shift); shift -= ONE_SHIFT; }
        prv = pos + 1;
        pos = name.find('+', prv);
    }
    return res;
}

template <class T>
std::string TransformFactory<T>::getName(uint64 functionType)
{
    std::string res;
    for (int i = 0; i < 8; i++) {
        const uint64 t = (functionType >> (MAX_SHIFT - ONE_SHIFT * i)) & MASK;
        if (t == NONE_TYPE) continue;
        const char* nm = nullptr;
        for (auto& e : TNAMES) if (uint64(e.type) == t) nm = e.name;
        if (nm == nullptr) throw std::invalid_argument("Unknown transform type");
        if (!res.empty()) res += '+';
        res += nm;
    }
    return res.empty() ? std::string("NONE") : res;
}

template <class T>
TransformSequence<T>* TransformFactory<T>::newTransform(Context& ctx, uint64 functionType)
{
    Transform<T>* transforms[8];
    int nbtr = 0;
    for (int i = 0; i < 8; i++) transforms[i] = nullptr;
    try {
        for (int i = 0; i < 8; i++) {
            const uint64 t = (functionType >> (MAX_SHIFT - ONE_SHIFT * i)) & MASK;
            if ((t == NONE_TYPE) && (i != 0)) continue;
            switch (int(t)) {
            case NONE_TYPE: transforms[nbtr++] = new NullTransform(ctx); break;
            case BWT_TYPE: transforms[nbtr++] = new BWTBlockCodec(ctx); break;
            case MTFT_TYPE: transforms[nbtr++] = new SBRT(SBRT::MODE_MTF, ctx); break;
            case RANK_TYPE: transforms[nbtr++] = new SBRT(SBRT::MODE_RANK, ctx); break;
            case SRT_TYPE: transforms[nbtr++] = new SRT(ctx); break;
            case ZRLT_TYPE: transforms[nbtr++] = new ZRLT(ctx); break;
            case RLT_TYPE: transforms[nbtr++] = new RLT(ctx); break;
            case LZ_TYPE: ctx.putInt("lz", LZ_TYPE); transforms[nbtr++] = new LZCodec(ctx); break;
            case LZX_TYPE: ctx.putInt("lz", LZX_TYPE); transforms[nbtr++] = new LZCodec(ctx); break;
            case DICT_TYPE:
                ctx.putInt("textcodec", hoststage::textVariantFor(ctx.has("entropy") ? ctx.getString("entropy").c_str() : ""));
                transforms[nbtr++] = new TextCodec(ctx);
                break;
            case UTF_TYPE: transforms[nbtr++] = new UTFCodec(ctx); break;
            default: {
                std::stringstream ss;
                ss << "Transform type " << t << " has no device kernel (out of scope of the accelerated block pipeline)";
                throw std::invalid_argument(ss.str());
            }
            }
        }
    } catch (...) {
        for (int i = 0; i < 8; i++) delete transforms[i];
        throw;
    }
    return new TransformSequence<T>(transforms, true);
}

template class TransformFactory<byte>;

// ------------------------------------------------------------------------------------------------
// entropy codecs
// ------------------------------------------------------------------------------------------------
int DeviceEntropyEncoder::encode(const byte block[], uint blkptr, uint len)
{
    if (len == 0) return 0;
    knz_ctx* c = deviceContext();
    knz_params p;
    memset(&p, 0, sizeof(p));
    p.entropy_type = _type;
    p.block_size = int32_t((len + 15) & ~15u);
    const size_t cap = knz_hip_encode_bound(&p, len) + 64;
    std::vector<byte> out(cap);
    uint64_t bits = 0;
    devCheck(c, knz_hip_entropy_encode(c, _type, &block[blkptr], len, out.data(), cap, &bits), "entropy encode");
    uint64 done = 0;
    while (done < bits) {
        const uint64 chunk = std::min<uint64>(bits - done, uint64(1) << 30);   // multiple of 8 except for the last piece
        _obs.writeBits(&out[size_t(done >> 3)], uint(chunk));
        done += chunk;
    }
    return int(len);
}

int DeviceEntropyDecoder::decode(byte block[], uint blkptr, uint len)
{
    if (len == 0) return 0;
    knz_ctx* c = deviceContext();
    int32_t decoded = 0;
    uint64_t used = 0;
    const int ver = (_bsVersion == 0) ? 1 : _bsVersion;     // a declared version 0 is an old layout (knz_params.bs_version: 0 = unset)
    DefaultInputBitStream* dibs = dynamic_cast<DefaultInputBitStream*>(&_ibs);
    if (dibs != nullptr) {
        // The device needs the block's bits in one buffer. Only as much of the stream as `len` symbols can occupy is handed
        // over (16 bits per symbol, a header per 16 KiB chunk, 256 tables per 4 MiB for ANS1) -- not the whole rest of the
        // stream on every call; should a valid block ever need more, the second try takes everything.
        const uint64 bound = 8 * (2ull * len + (uint64(len) / 16384 + 2) * 640 + (_type == KNZ_E_ANS1 ? (uint64(len) / (4u << 20) + 1) * 256 * 576 : 0) + 4096);
        const byte* data; uint64 startBit, endBit;
        dibs->peekAhead(bound, &data, &startBit, &endBit);
        devCheck(c, knz_hip_entropy_decode_v(c, _type, ver, data, endBit, startBit, &block[blkptr], len, &decoded, &used), "entropy decode");
        if (decoded != int32_t(len) && endBit - startBit >= bound) {
            dibs->peekRemaining(&data, &startBit, &endBit);
            devCheck(c, knz_hip_entropy_decode_v(c, _type, ver, data, endBit, startBit, &block[blkptr], len, &decoded, &used), "entropy decode");
        }
        if (decoded == int32_t(len)) dibs->skip(used);
        return int(decoded);
    }
    // generic InputBitStream (no way to look ahead without consuming): the rest of the stream is drained into one buffer, so
    // only ONE decode() per stream is possible through such an object; DefaultInputBitStream has no such limit
    std::vector<byte> rest;
    try { while (_ibs.hasMoreToRead()) rest.push_back(byte(_ibs.readBits(8))); } catch (const BitStreamException&) {}
    devCheck(c, knz_hip_entropy_decode_v(c, _type, ver, rest.data(), uint64(rest.size()) * 8, 0, &block[blkptr], len, &decoded, &used), "entropy decode");
    return int(decoded);
}

// entropy/ANSRangeEncoder.cpp:36-68, ANSRangeDecoder.cpp:36-64, HuffmanEncoder.cpp:32-44, HuffmanDecoder.cpp:32-48: the reference's checks,
// then the one this implementation adds (the kernels exist for the default chunk size and range only)
static void checkAnsArgs(int order, int chunkSize, int logRange)
{
    if ((order != 0) && (order != 1)) throw std::invalid_argument("ANS Codec: The order must be 0 or 1");
    if (chunkSize < ANSRangeEncoder::MIN_CHUNK_SIZE) throw std::invalid_argument("ANS Codec: The chunk size must be at least " + std::to_string(ANSRangeEncoder::MIN_CHUNK_SIZE));
    if (chunkSize > ANSRangeEncoder::MAX_CHUNK_SIZE) throw std::invalid_argument("ANS Codec: The chunk size must be at most " + std::to_string(ANSRangeEncoder::MAX_CHUNK_SIZE));
    if ((logRange < 8) || (logRange > 15)) throw std::invalid_argument("ANS Codec: Invalid range: " + std::to_string(logRange) + " (must be in [8..15])");
    if (chunkSize != ANSRangeEncoder::DEFAULT_ANS0_CHUNK_SIZE || logRange != ANSRangeEncoder::DEFAULT_LOG_RANGE)
        throw std::invalid_argument("ANS Codec: the device kernels are built for the default chunk size (16384) and range (12)");
}

static void checkHuffmanArgs(int chunkSize)
{
    if (chunkSize < 1024) throw std::invalid_argument("Huffman codec: The chunk size must be at least 1024");
    if (chunkSize > HuffmanCommon::MAX_CHUNK_SIZE) throw std::invalid_argument("Huffman codec: The chunk size must be at most " + std::to_string(HuffmanCommon::MAX_CHUNK_SIZE));
    if (chunkSize != HuffmanCommon::MAX_CHUNK_SIZE) throw std::invalid_argument("Huffman codec: the device kernels are built for the default chunk size (16384)");
}

ANSRangeEncoder::ANSRangeEncoder(OutputBitStream& obs, int order, int chunkSize, int logRange)
    : DeviceEntropyEncoder(obs, order == 1 ? KNZ_E_ANS1 : KNZ_E_ANS0)
{
    checkAnsArgs(order, chunkSize, logRange);
}

ANSRangeDecoder::ANSRangeDecoder(InputBitStream& ibs, int order, int chunkSize) : DeviceEntropyDecoder(ibs, order == 1 ? KNZ_E_ANS1 : KNZ_E_ANS0)
{
    checkAnsArgs(order, chunkSize, ANSRangeEncoder::DEFAULT_LOG_RANGE);
}

HuffmanEncoder::HuffmanEncoder(OutputBitStream& obs, int chunkSize) : DeviceEntropyEncoder(obs, KNZ_E_HUFFMAN) { checkHuffmanArgs(chunkSize); }

HuffmanDecoder::HuffmanDecoder(InputBitStream& ibs, Context* pCtx, int chunkSize)
    : DeviceEntropyDecoder(ibs, KNZ_E_HUFFMAN, pCtx != nullptr ? pCtx->getInt("bsVersion", 6) : 6)
{
    checkHuffmanArgs(chunkSize);
}

static const struct { const char* name; short type; } ENAMES[] = {
    {"NONE", 0}, {"HUFFMAN", 1}, {"FPAQ", 2}, {"RANGE", 4}, {"ANS0", 5}, {"CM", 6}, {"TPAQ", 7}, {"ANS1", 8}, {"TPAQX", 9} };

const char* EntropyEncoderFactory::getName(short entropyType)
{
    for (auto& e : ENAMES) if (e.type == entropyType) return e.name;
    throw std::invalid_argument("Unknown entropy codec type");
}

short EntropyEncoderFactory::getType(const char* str)
{
    std::string name(str);
    std::transform(name.begin(), name.end(), name.begin(), ::toupper);
    for (auto& e : ENAMES) if (name == e.name) return e.type;
    throw std::invalid_argument("Unsupported entropy codec type: '" + name + "'");
}

EntropyEncoder* EntropyEncoderFactory::newEncoder(OutputBitStream& obs, Context&, short entropyType)
{
    switch (entropyType) {
    case HUFFMAN_TYPE: return new HuffmanEncoder(obs);
    case ANS0_TYPE: return new ANSRangeEncoder(obs, 0);
    case ANS1_TYPE: return new ANSRangeEncoder(obs, 1);
    case FPAQ_TYPE: return new FPAQEncoder(obs);
    case NONE_TYPE: return new NullEntropyEncoder(obs);
    default: throw std::invalid_argument(std::string("Entropy codec '") + getName(entropyType) + "' has no device kernel");
    }
}

EntropyDecoder* EntropyDecoderFactory::newDecoder(InputBitStream& ibs, Context& ctx, short entropyType)
{
    switch (entropyType) {
    case EntropyEncoderFactory::HUFFMAN_TYPE: return new HuffmanDecoder(ibs, &ctx);      // EntropyDecoderFactory.hpp:66-67
    case EntropyEncoderFactory::ANS0_TYPE: return new ANSRangeDecoder(ibs, 0);
    case EntropyEncoderFactory::ANS1_TYPE: return new ANSRangeDecoder(ibs, 1);
    case EntropyEncoderFactory::FPAQ_TYPE: return new FPAQDecoder(ibs);
    case EntropyEncoderFactory::NONE_TYPE: return new NullEntropyDecoder(ibs);
    default: throw std::invalid_argument(std::string("Entropy codec '") + getName(entropyType) + "' has no device kernel");
    }
}

// ------------------------------------------------------------------------------------------------
// stream header (io/CompressedOutputStream.cpp:277-342, io/CompressedInputStream.cpp:511-663)
// ------------------------------------------------------------------------------------------------
static uint32_t headerChecksum(uint32_t ckSize, uint32_t etype, uint64 ttype, uint32_t blockSize, int szMask, uint64 size)
{
    const uint32_t HASH = 0x1E35A7BDu;
    uint32_t c = HASH * (0x01030507u * 6u);
    c ^= HASH * uint32_t(~ckSize);
    c ^= HASH * uint32_t(~etype);
    c ^= HASH * uint32_t((~ttype) >> 32);
    c ^= HASH * uint32_t(~ttype);
    c ^= HASH * uint32_t(~blockSize);
    if (szMask != 0) { c ^= HASH * uint32_t((~size) >> 32); c ^= HASH * uint32_t(~size); }
    return ((c >> 23) ^ (c >> 3)) & 0xFFFFFFu;
}

struct BitPacker {
    std::vector<byte> bytes; uint64 nbits = 0;
    void put(uint64 v, uint n) { for (int i = int(n) - 1; i >= 0; i--) { if ((nbits & 7) == 0) bytes.push_back(0); bytes.back() |= byte(((v >> i) & 1) << (7 - (nbits & 7))); nbits++; } }
};

template <class BUF>
static uint64 getBitsAt(const BUF& d, uint64 pos, uint n)
{
    uint64 v = 0;
    for (uint i = 0; i < n; i++, pos++) v = (v << 1) | ((d[size_t(pos >> 3)] >> (7 - (pos & 7))) & 1);
    return v;
}

// ------------------------------------------------------------------------------------------------
// CompressedOutputStream
// ------------------------------------------------------------------------------------------------
// Page-locked staging memory is expensive to create (the pages are pinned one by one) and cheap to keep: buffers go
// back to a process-wide pool when a stream is done with them.
namespace {
struct PinnedPool {
    std::mutex mu;
    std::vector<std::pair<void*, size_t>> idle;
    byte* get(size_t bytes, size_t* cap)
    {
        {
            std::lock_guard<std::mutex> l(mu);
            size_t best = idle.size();
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i].second >= bytes && (best == idle.size() || idle[i].second < idle[best].second)) best = i;
            if (best != idle.size()) { void* p = idle[best].first; *cap = idle[best].second; idleBytes -= *cap; idle.erase(idle.begin() + long(best)); return static_cast<byte*>(p); }
        }
        void* p = nullptr;
        const size_t want = bytes + (bytes >> 3) + 4096;
        if (knz_hip_host_alloc(want, &p) != 0 || p == nullptr) throw IOException("cannot allocate page-locked staging memory", Error::ERR_CREATE_STREAM);
        *cap = want;
        return static_cast<byte*>(p);
    }
    // what idles here is bounded by bytes (2 GiB) and entries (64), not by a handful of buffers: a stream with six lanes has a
    // dozen staging buffers, and pinning 16 MiB again costs milliseconds (with a pool of eight, every lane beyond the fourth paid
    // that on every stream: the lane sweeps of round 5 got slower with every lane for this reason alone)
    size_t idleBytes = 0;
    void put(byte* p, size_t cap)
    {
        if (!p) return;
        std::lock_guard<std::mutex> l(mu);
        if (idle.size() >= 64 || idleBytes + cap > (size_t(2) << 30)) { knz_hip_host_free(p); return; }
        idle.push_back(std::make_pair(static_cast<void*>(p), cap));
        idleBytes += cap;
    }
    // everything that idles goes back to the system (releaseIdleBuffers)
    size_t trim()
    {
        std::vector<std::pair<void*, size_t>> drop;
        { std::lock_guard<std::mutex> l(mu); drop.swap(idle); idleBytes = 0; }
        size_t n = 0;
        for (auto& d : drop) { knz_hip_host_free(d.first); n += d.second; }
        return n;
    }
};
PinnedPool g_pinned;

// Device buffers of the lanes go back to a pool per context as well: hipMalloc / hipFree cost tens to hundreds of microseconds
// each (hipFree waits for the device), and a stream has a dozen of them.
struct DevPool {
    std::mutex mu;
    std::map<knz_ctx*, std::vector<std::pair<void*, size_t>>> idle;
    void* get(knz_ctx* c, size_t bytes, size_t* cap)
    {
        {
            std::lock_guard<std::mutex> l(mu);
            auto& v = idle[c];
            size_t best = v.size();
            for (size_t i = 0; i < v.size(); i++)
                if (v[i].second >= bytes && v[i].second <= 2 * bytes + (size_t(1) << 20) && (best == v.size() || v[i].second < v[best].second)) best = i;
            if (best != v.size()) { void* p = v[best].first; *cap = v[best].second; v.erase(v.begin() + long(best)); return p; }
        }
        void* p = nullptr;
        if (knz_hip_malloc(c, bytes, &p) != 0 || p == nullptr) {
            // make room: whatever idles goes first -- in every context's pool (the lanes of one GPU are several contexts on the same memory)
            trim();
            devCheck(c, knz_hip_malloc(c, bytes, &p), "malloc");
        }
        *cap = bytes;
        return p;
    }
    size_t trim()
    {
        std::map<knz_ctx*, std::vector<std::pair<void*, size_t>>> drop;
        { std::lock_guard<std::mutex> l(mu); drop.swap(idle); }
        size_t n = 0;
        for (auto& kv : drop) for (auto& d : kv.second) { knz_hip_free(kv.first, d.first); n += d.second; }
        return n;
    }
    // kept per context: at most 12 buffers and 1 GiB, nothing above 256 MiB (a stream with 1 GiB blocks gives its memory back)
    void put(knz_ctx* c, void* p, size_t cap)
    {
        if (!p) return;
        if (cap <= (size_t(256) << 20)) {
            std::lock_guard<std::mutex> l(mu);
            auto& v = idle[c];
            size_t held = 0;
            for (auto& e : v) held += e.second;
            if (v.size() < 12 && held + cap <= (size_t(1) << 30)) { v.push_back(std::make_pair(p, cap)); return; }
        }
        knz_hip_free(c, p);
    }
};
DevPool g_devPool;

// How many device calls of the stream classes run on one GPU at a time (KNZ_DEVICE_CONCURRENCY, default 3). A stream has more lanes
// than that: three concurrent batches are what saturates the device (measured: 3, 4 and 6 concurrent two-block encodes all deliver
// about 7.7 GB/s), and the lanes beyond them hold the batches that are being filled, uploaded, downloaded or written meanwhile --
// with as many lanes as concurrent calls, every lane's kernels waited for its own staging traffic. First come, first served.
struct DeviceGate {
    std::mutex mu; std::condition_variable cv; int inUse = 0; uint64_t next = 0, serving = 0; int cap;
    DeviceGate()
    {
        const char* e = getenv("KNZ_DEVICE_CONCURRENCY");
        cap = e ? std::max(1, std::min(16, atoi(e))) : 3;
    }
    void acquire()
    {
        std::unique_lock<std::mutex> l(mu);
        const uint64_t my = next++;
        cv.wait(l, [&] { return my == serving && inUse < cap; });
        serving++; inUse++;
        cv.notify_all();
    }
    void release() { { std::lock_guard<std::mutex> l(mu); inUse--; } cv.notify_all(); }
};
struct GateHold {
    DeviceGate& g;
    explicit GateHold(DeviceGate& gate) : g(gate) { g.acquire(); }
    ~GateHold() { g.release(); }
};
DeviceGate& gateOf(int device)
{
    static std::mutex mu;
    static std::map<int, DeviceGate*> gates;
    std::lock_guard<std::mutex> l(mu);
    DeviceGate*& g = gates[device];
    if (!g) g = new DeviceGate();
    return *g;
}
}

// Page-locked staging buffers and device buffers that finished streams have left in the process-wide pools (up to 2 GiB of pinned host
// memory, up to 1 GiB of device memory per lane context) go back to the system. Buffers of live streams are not touched. Returns the bytes freed.
size_t releaseIdleBuffers() { return g_pinned.trim() + g_devPool.trim(); }

FetchBuf::~FetchBuf() { g_pinned.put(_p, _cap); }

void FetchBuf::reserve(size_t n)
{
    if (n + 16 <= _cap) return;
    size_t cap = 0;
    byte* q = g_pinned.get(std::max(n + 16, _cap + (_cap >> 1)), &cap);
    if (_n) memcpy(q, _p, _n);
    g_pinned.put(_p, _cap);
    _p = q; _cap = cap;
}

void FetchBuf::resize(size_t n) { reserve(n); _n = n; }

void FetchBuf::resize(size_t n, byte fill)
{
    reserve(n);
    if (n > _n) memset(_p + _n, fill, n - _n);
    _n = n;
}

void FetchBuf::dropFront(size_t n)
{
    if (n >= _n) { _n = 0; return; }
    memmove(_p, _p + n, _n - n);
    _n -= n;
}

CompressedOutputStream::CompressedOutputStream(std::ostream& os, int tasks, const std::string& entropy, const std::string& transform,
                                               int blockSize, int checksum, uint64 fileSize, ThreadPool*, bool headerless)
    : std::ostream(os.rdbuf()), _os(os)
{
    init(tasks, entropy, transform, blockSize, checksum, fileSize, headerless);
}

CompressedOutputStream::CompressedOutputStream(std::ostream& os, int tasks, const std::string& entropy, const std::string& transform,
                                               int blockSize, int checksum, uint64 fileSize, bool headerless)
    : std::ostream(os.rdbuf()), _os(os)
{
    init(tasks, entropy, transform, blockSize, checksum, fileSize, headerless);
}

CompressedOutputStream::CompressedOutputStream(std::ostream& os, Context& ctx, bool headerless)
    : std::ostream(os.rdbuf()), _os(os)
{
    const int64 fileSize = ctx.getLong("fileSize", 0);
    init(ctx.getInt("jobs", 1), ctx.getString("entropy"), ctx.getString("transform"), ctx.getInt("blockSize"), ctx.getInt("checksum", 0),
         fileSize < 0 ? 0 : uint64(fileSize), headerless);
}

bool CompressedOutputStream::addListener(Listener<Event>& bl) { _listeners.push_back(&bl); return true; }

bool CompressedOutputStream::removeListener(Listener<Event>& bl)
{
    auto it = std::find(_listeners.begin(), _listeners.end(), &bl);
    if (it == _listeners.end()) return false;
    _listeners.erase(it);
    return true;
}

void CompressedOutputStream::init(int tasks, const std::string& entropy, const std::string& transform, int blockSize, int checksum, uint64 fileSize,
                                  bool headerless)
{
    if ((tasks <= 0) || (tasks > 64)) throw std::invalid_argument("The number of jobs must be in [1..64], got " + std::to_string(tasks));
    if (blockSize > 1024 * 1024 * 1024) throw std::invalid_argument("The block size must be at most 1024 MB");
    if (blockSize < 1024) throw std::invalid_argument("The block size must be at least 1024");
    if ((blockSize & -16) != blockSize) throw std::invalid_argument("The block size must be a multiple of 16");
    if ((checksum != 0) && (checksum != 32) && (checksum != 64)) throw std::invalid_argument("The block checksum size must be 0, 32 or 64");
    _jobs = tasks; _blockSize = blockSize; _checksum = checksum;
    _entropyType = EntropyEncoderFactory::getType(entropy.c_str());
    _transformType = TransformFactory<byte>::getType(transform.c_str());
    _inputSize = fileSize;
    _headless = headerless; _closed = false; _headerDone = false;
    // Blocks per device call.  `jobs` only selects the reference's buffer-slot capacities in the bitstream; the
    // GPU wants many blocks per launch and the host wants several batches in flight (one per lane: the caller fills a staging
    // slot while the lanes work), so by default a batch is 24 MiB (at least one block, at most 64; measured best with six lanes of which three are in the kernels at a time: short tail, decode batches overlap; round 4: 16 MiB with four lanes).
    { const int64_t want = (int64_t(24) << 20) / int64_t(blockSize); _batchBlocks = int(std::min<int64_t>(64, std::max<int64_t>(1, want))); }
    const char* e = getenv("KNZ_BATCH_BLOCKS");
    if (e && atoi(e) > 0) _batchBlocks = atoi(e);
    // one device call takes at most 2 GiB of input (32-bit positions on the device side)
    { const int64_t lim = (int64_t(1) << 31) / int64_t(blockSize) - 1; if (_batchBlocks > lim) _batchBlocks = int(lim < 1 ? 1 : lim); }
    // chains that start with TEXT / UTF (the level presets 5 and 6): those stages run on the host, block by block, and every block
    // goes to the device on its own (its length after them differs from block to block)
    _hosted = hostedStagesOf(_transformType, _hostIds);
    if (_hosted) _batchBlocks = 1;
    _blockId = 0;
    _pendingByte = 0; _pendingBits = 0; _written = 0;
    _fillLane = 0; _nextSeq = 0; _sinkSeq = 0; _pubSeq = 0; _cumBits = 0; _stop = false;
    _batchBytes = 0;
    for (auto& t : _tns) t = 0;
    _t0 = std::chrono::steady_clock::now();
    { const char* st = getenv("KNZ_SINK_THREAD"); _sinkThread = !(st && atoi(st) == 0); }
    _spreadCopies = chainIsHostBound(_transformType);
    // lanes per device bounded by the block size: a lane holds a batch's input, output and the suffix sort's scratch (about 64
    // bytes per input byte), so that with 1 GiB blocks one lane per GPU is what fits comfortably, with blocks up to 256 MiB four
    std::vector<int> devs = laneDevices();
    {
        const size_t perDev = std::max<size_t>(1, (size_t(1) << 30) / size_t(std::max(blockSize, 1)));
        std::map<int, size_t> seen;
        std::vector<int> kept;
        for (int d : devs) if (seen[d]++ < perDev) kept.push_back(d);
        devs.swap(kept);
    }
    const std::vector<knz_ctx*> ctxs = openLanes(devs);
    _lanes.resize(devs.size());
    for (size_t i = 0; i < _lanes.size(); i++) {
        Lane& ln = _lanes[i];
        ln.device = devs[i]; ln.ctx = ctxs[i];
        ln.in = nullptr; ln.inCap = 0; ln.n = 0; ln.last = false; ln.dIn = nullptr; ln.dInCap = 0; ln.ticket = 0;
        ln.dOut = nullptr; ln.dOutCap = 0; ln.dShift = nullptr; ln.dShiftCap = 0;
        ln.out = nullptr; ln.outCap = 0; ln.outBytes = 0; ln.shiftR = 0; ln.bits = 0; ln.seq = -1; ln.firstBlock = 0; ln.state = 0;
    }
}

CompressedOutputStream::~CompressedOutputStream()
{
    try { close(); } catch (...) {}
    { std::lock_guard<std::mutex> l(_mu); _stop = true; }
    _cv.notify_all();
    if (_sink.joinable()) _sink.join();
    for (Lane& ln : _lanes) if (ln.worker.joinable()) ln.worker.join();
    for (Lane& ln : _lanes) {
        knz_hip_copy_wait(ln.ctx, ln.ticket);
        g_devPool.put(ln.ctx, ln.dIn, ln.dInCap);
        g_devPool.put(ln.ctx, ln.dOut, ln.dOutCap);
        g_devPool.put(ln.ctx, ln.dShift, ln.dShiftCap);
        g_pinned.put(ln.in, ln.inCap);
        g_pinned.put(ln.out, ln.outCap);
    }
}

void CompressedOutputStream::rethrow()
{
    std::exception_ptr e;
    { std::lock_guard<std::mutex> l(_mu); e = _err; }          // a failure is sticky: no batch is submitted after a failed one
    if (e) std::rethrow_exception(e);
}

std::ostream& CompressedOutputStream::write(const char* data, std::streamsize length)
{
    if (length < 0) throw IOException("Invalid buffer size");
    if (_closed) throw IOException("Stream closed", Error::ERR_WRITE_FILE);
    rethrow();
    if (_batchBytes == 0) _batchBytes = size_t(_batchBlocks) * size_t(_blockSize);      // latched: the staging slots are sized for it
    const size_t batchBytes = _batchBytes;
    size_t off = 0;
    while (off < size_t(length)) {
        Lane& ln = _lanes[size_t(_fillLane)];                  // free: enqueue() waited for it
        if (ln.in == nullptr) ln.in = g_pinned.get(batchBytes + 64, &ln.inCap);
        const size_t take = std::min(size_t(length) - off, batchBytes - ln.n);
        { ScopedNs t_(_tns[0]); parCopy(ln.in + ln.n, data + off, take, _spreadCopies); }
        ln.n += take;
        off += take;
        if (ln.n == batchBytes) enqueue(false);
    }
    return *this;
}

std::ostream& CompressedOutputStream::put(char c) { return write(&c, 1); }

// caller's thread: the next batch in stream order goes to the sink if its bytes are back (the lock is dropped around the
// write); false if it is not ready
bool CompressedOutputStream::drainOne(std::unique_lock<std::mutex>& l)
{
    Lane& ln = _lanes[size_t(_sinkSeq % int64(_lanes.size()))];
    if (ln.state != 2 || ln.seq != _sinkSeq) return false;
    l.unlock();
    bool bad = false;
    std::exception_ptr thrown;               // a sink with exceptions() set, or a streambuf that throws (disk full, a custom sink): on the
                                             // sink thread nothing is above this frame, so the exception is kept for the caller's next call
    try {
        // the run starts with shiftR zero bits: the place of the previous run's last bits
        if (ln.shiftR != 0 && ln.outBytes != 0) ln.out[0] |= _pendingByte;
        const uint64 totalBits = uint64(ln.shiftR) + ln.bits;
        const size_t full = size_t(totalBits >> 3);
        const uint rem = uint(totalBits & 7);
        const size_t toWrite = (ln.last && rem) ? full + 1 : full;          // close(): the last byte is zero padded
        if (toWrite) {
            ScopedNs t_(_tns[2]);
            _os.write(reinterpret_cast<const char*>(ln.out), std::streamsize(toWrite));
            bad = _os.fail();
            if (!bad) _written += toWrite;
        }
        _pendingBits = ln.last ? 0 : rem;
        _pendingByte = (rem && !ln.last) ? ln.out[full] : 0;
        if (hostTimeline()) fprintf(stderr, "[knz out batch %lld] in the sink at %.2f ms\n", (long long)ln.seq, msSince(_t0));
    } catch (...) {
        thrown = std::current_exception();
    }
    l.lock();
    ln.state = 0;
    ln.n = 0;
    _sinkSeq++;
    if (thrown && !_err) _err = thrown;      // (rethrown as it is by write() / close() on the caller's thread: rethrow())
    if (bad && !_err) _err = std::make_exception_ptr(IOException("Write to bitstream failed", Error::ERR_WRITE_FILE));
    _cv.notify_all();
    return true;
}

// hand the lane being filled to its worker (the host-to-device copy starts right away), append finished batches to the sink,
// and wait until the next lane is free
void CompressedOutputStream::enqueue(bool last)
{
    rethrow();
    Lane& ln = _lanes[size_t(_fillLane)];
    if (!ln.worker.joinable()) ln.worker = std::thread(&CompressedOutputStream::workerLoop, this, _fillLane);
    {
        // the device buffer of this lane was last read by the batch that freed the lane: safe to overwrite
        knz_ctx* c = ln.ctx;
        if (ln.dInCap < ln.n + 64) {
            g_devPool.put(c, ln.dIn, ln.dInCap);
            ln.dIn = nullptr; ln.dInCap = 0;
            const size_t want = std::max(ln.n, _batchBytes) + 64;       // a full batch (a stream shorter than one still gets a slot others can reuse)
            ln.dIn = g_devPool.get(c, want, &ln.dInCap);
        }
        ln.ticket = 0;
        if (ln.n) devCheck(c, knz_hip_memcpy_h2d_async(c, ln.dIn, ln.in, ln.n, &ln.ticket), "h2d");
    }
    {
        std::unique_lock<std::mutex> l(_mu);
        ln.last = last;
        ln.seq = _nextSeq++;
        if (hostTimeline()) fprintf(stderr, "[knz out batch %lld] enqueued at %.2f ms\n", (long long)ln.seq, msSince(_t0));
        ln.firstBlock = _blockId;
        _blockId += int64((ln.n + size_t(_blockSize) - 1) / size_t(_blockSize));
        ln.state = 1;
        _fillLane = (_fillLane + 1) % int(_lanes.size());
        _cv.notify_all();
        if (_sinkThread) {
            if (!_sink.joinable()) _sink = std::thread(&CompressedOutputStream::sinkLoop, this);
            ScopedNs t_(_tns[1]);
            _cv.wait(l, [&] { return _lanes[size_t(_fillLane)].state == 0 || _err; });
        } else
        for (;;) {
            {
                ScopedNs t_(_tns[1]);
                _cv.wait(l, [&] { const Lane& nx = _lanes[size_t(_fillLane)];
                                  const Lane& sk = _lanes[size_t(_sinkSeq % int64(_lanes.size()))];
                                  return nx.state == 0 || _err || (sk.state == 2 && sk.seq == _sinkSeq); });
            }
            if (!drainOne(l)) break;
        }
    }
    rethrow();
}

// sink thread: the runs go to the sink in batch order as their bytes arrive; ends with the stream (close() sets _stop once
// everything is out) or behind a failed batch
void CompressedOutputStream::sinkLoop()
{
    std::unique_lock<std::mutex> l(_mu);
    for (;;) {
        _cv.wait(l, [&] { const Lane& sk = _lanes[size_t(_sinkSeq % int64(_lanes.size()))];
                          return _stop || _err || (sk.state == 2 && sk.seq == _sinkSeq); });
        if (!drainOne(l)) { if (_stop || _err) return; }
    }
}

void CompressedOutputStream::workerLoop(int lane)
{
    Lane& ln = _lanes[size_t(lane)];
    for (;;) {
        {
            std::unique_lock<std::mutex> l(_mu);
            _cv.wait(l, [&] { return _stop || _err || ln.state == 1; });
            if (ln.state != 1 || _err) return;
        }
        const bool last = ln.last;
        std::exception_ptr ex;
        try {
            submit(ln);
        } catch (...) {
            ex = std::current_exception();
        }
        {
            // a failure is recorded and the lane released in ONE critical section: a caller that leaves its wait because of
            // the error never sees a lane its worker still owns
            std::lock_guard<std::mutex> l(_mu);
            if (ex) { if (!_err) _err = ex; ln.n = 0; ln.state = 0; }
        }
        _cv.notify_all();
        if (last || ex) return;
    }
}

// worker thread of a lane: one batch through the device; the compressed run ends up in the lane's page-locked output buffer
void CompressedOutputStream::submit(Lane& ln)
{
    knz_ctx* c = ln.ctx;
    const size_t n = ln.n;
    knz_params p;
    memset(&p, 0, sizeof(p));
    p.transform_type = _transformType; p.entropy_type = _entropyType; p.block_size = _blockSize; p.checksum_bits = _checksum; p.jobs = _jobs;
    // prologue: the stream header in front of the first batch
    BitPacker pro;
    if (ln.seq == 0 && !_headless) {
        const uint32_t ckSize = _checksum == 32 ? 1 : (_checksum == 64 ? 2 : 0);
        pro.put(0x4B414E5Au, 32); pro.put(6, 4); pro.put(ckSize, 2); pro.put(uint64(_entropyType), 5); pro.put(_transformType, 48);
        pro.put(uint64(_blockSize >> 4), 28);
        int szMask = 0;
        if (_inputSize != 0 && _inputSize < (uint64(1) << 48)) { int lg = 63; while (!((_inputSize >> lg) & 1)) lg--; szMask = (lg >> 4) + 1; }
        pro.put(uint64(szMask), 2);
        if (szMask) pro.put(_inputSize, uint(16 * szMask));
        pro.put(0, 15);
        pro.put(headerChecksum(ckSize, uint32_t(_entropyType), _transformType, uint32_t(_blockSize), szMask, _inputSize), 24);
    }
    const size_t cap = knz_hip_encode_bound(&p, n) + pro.bytes.size() + 256;
    if (ln.dOutCap < cap) { g_devPool.put(c, ln.dOut, ln.dOutCap); ln.dOut = nullptr; ln.dOutCap = 0; ln.dOut = g_devPool.get(c, cap + (cap >> 2), &ln.dOutCap); }
    { ScopedNs t_(_tns[3]); devCheck(c, knz_hip_copy_wait(c, ln.ticket), "h2d"); }          // queued by enqueue(), normally long complete
    ln.ticket = 0;
    uint64_t bits = 0;
    const double tlUp = hostTimeline() ? msSince(_t0) : 0;
    GateHold* gate = new GateHold(gateOf(ln.device));       // released as soon as the kernels are done (below), whatever happens
    const double tlGate = hostTimeline() ? msSince(_t0) : 0;
    std::unique_ptr<GateHold> gateOwner(gate);
    std::chrono::steady_clock::time_point tk0 = std::chrono::steady_clock::now();
    if (_hosted && n == 0) p.transform_type = 0;             // (the empty last batch: end marker only; the device call checks the chain before it looks at the size)
    if (_hosted && n > 0) {
        // host stages first (the original bytes are still in the lane's staging buffer), then the block in its new length to the device
        if (n > size_t(_blockSize)) throw IOException("a chain with host stages takes one block per call", Error::ERR_PROCESS_BLOCK);
        const uint8_t* orig = reinterpret_cast<const uint8_t*>(ln.in);
        knz_host_stages hs;
        memset(&hs, 0, sizeof(hs));
        hs.stages = _hosted; hs.orig_len = uint32_t(n);
        if (_checksum) hs.checksum = hostChecksum(orig, int(n), _checksum);
        const HostedResult r = runHostStages(_hostIds, _hosted, orig, int(n), _blockSize, _entropyType, ln.hostA, ln.hostB);
        hs.applied_mask = r.applied;
        if (r.applied) devCheck(c, knz_hip_memcpy_h2d(c, ln.dIn, r.data, size_t(r.len)), "h2d");
        devCheck(c, knz_hip_encode_block_hosted(c, &p, &hs, static_cast<const uint8_t*>(ln.dIn), size_t(r.len), pro.bytes.empty() ? nullptr : pro.bytes.data(), uint32_t(pro.nbits),
                                                ln.firstBlock, ln.last ? 1 : 0, static_cast<uint8_t*>(ln.dOut), ln.dOutCap, &bits), "encode block");
    } else
    devCheck(c, knz_hip_encode_blocks(c, &p, static_cast<const uint8_t*>(ln.dIn), n, pro.bytes.empty() ? nullptr : pro.bytes.data(), uint32_t(pro.nbits),
                                      ln.firstBlock, ln.last ? 1 : 0, static_cast<uint8_t*>(ln.dOut), ln.dOutCap, &bits), "encode blocks");
    gateOwner.reset();
    const double tlKern = hostTimeline() ? msSince(_t0) : 0;
    _tns[4] += uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tk0).count());
    // where the run starts: behind the runs of the batches before it, whose lengths are published in batch order
    uint64 start;
    {
        ScopedNs t_(_tns[5]);
        std::unique_lock<std::mutex> l(_mu);
        _cv.wait(l, [&] { return _pubSeq == ln.seq || _stop || _err; });
        if (_pubSeq != ln.seq) return;
        start = _cumBits;
        _cumBits += bits;
        _pubSeq++;
    }
    _cv.notify_all();
    ScopedNs t6_(_tns[6]);
    const uint r = uint(start & 7);
    const size_t bytes = size_t((uint64(r) + bits + 7) >> 3);
    const void* src = ln.dOut;
    if (r != 0 && bits != 0) {
        if (ln.dShiftCap < bytes + 8) {
            // sized by what a batch actually produced (plus a quarter, so that batches of similar size do not reallocate), never
            // beyond the encode bound: with large blocks the bound is several GiB per lane, the compressed run a fraction of it
            const size_t want = std::min(ln.dOutCap + 8, std::max(bytes + 8 + (bytes >> 2), size_t(1) << 20));
            g_devPool.put(c, ln.dShift, ln.dShiftCap);
            ln.dShift = nullptr; ln.dShiftCap = 0;
            ln.dShift = g_devPool.get(c, want, &ln.dShiftCap);
        }
        devCheck(c, knz_hip_shift_bits(c, static_cast<const uint8_t*>(ln.dOut), bits, r, static_cast<uint8_t*>(ln.dShift)), "shift");
        src = ln.dShift;
    }
    if (ln.outCap < bytes + 8) { g_pinned.put(ln.out, ln.outCap); ln.out = nullptr; ln.outCap = 0; ln.out = g_pinned.get(std::max(bytes + 8, cap / 2), &ln.outCap); }
    if (bits == 0) { if (ln.outCap) ln.out[0] = 0; }
    else devCheck(c, knz_hip_memcpy_d2h(c, ln.out, src, bytes), "d2h");
    {
        std::lock_guard<std::mutex> l(_mu);
        ln.shiftR = (bits != 0) ? r : 0;
        ln.bits = bits;
        ln.outBytes = bytes;
        if (bits == 0 && r != 0) { ln.shiftR = r; ln.outBytes = 1; }       // (an empty run in the middle of a byte: only the shared byte)
        ln.state = 2;
    }
    if (hostTimeline()) fprintf(stderr, "[knz out batch %lld] %zu B lane-uploaded %.2f gate %.2f kernels-done %.2f downloaded %.2f ms\n", (long long)ln.seq, n, tlUp, tlGate, tlKern, msSince(_t0));
    _cv.notify_all();
}

void CompressedOutputStream::close()
{
    if (_closed) return;
    _closed = true;
    try {
        bool failed;
        { std::lock_guard<std::mutex> l(_mu); failed = bool(_err); }
        if (!failed) enqueue(true);         // the last batch (possibly empty) carries the end marker; nothing follows a failed batch
        {
            std::unique_lock<std::mutex> l(_mu);
            if (_sinkThread) _cv.wait(l, [&] { return _sinkSeq == _nextSeq || _err; });
            else
            for (;;) {
                _cv.wait(l, [&] { const Lane& sk = _lanes[size_t(_sinkSeq % int64(_lanes.size()))];
                                  return _sinkSeq == _nextSeq || _err || (sk.state == 2 && sk.seq == _sinkSeq); });
                if (!drainOne(l)) break;
            }
            _stop = true;                     // releases the workers (all of them idle, or stuck behind a failed batch) and the sink thread
        }
        _cv.notify_all();
        if (_sink.joinable()) _sink.join();
        for (Lane& ln : _lanes) if (ln.worker.joinable()) ln.worker.join();
        rethrow();
        _os.flush();
        if (hostTimeline()) fprintf(stderr, "[knz out] closed at %.2f ms\n", msSince(_t0));
        if (hostTiming())
            fprintf(stderr, "[knz out] lanes %zu batch %zu B: caller copy-in %.2f ms, wait-lane %.2f, sink-write %.2f | workers upload-wait %.2f, kernels %.2f, order-wait %.2f, shift+download %.2f\n",
                    _lanes.size(), _batchBytes, _tns[0] / 1e6, _tns[1] / 1e6, _tns[2] / 1e6, _tns[3] / 1e6, _tns[4] / 1e6, _tns[5] / 1e6, _tns[6] / 1e6);
    } catch (const IOException&) {
        setstate(std::ios::badbit);
        throw;
    } catch (const std::exception& e) {
        setstate(std::ios::badbit);
        throw IOException(e.what(), Error::ERR_WRITE_FILE);
    }
    setstate(std::ios::eofbit);
}

// ------------------------------------------------------------------------------------------------
// CompressedInputStream
// ------------------------------------------------------------------------------------------------
CompressedInputStream::CompressedInputStream(std::istream& is, int tasks, const std::string& entropy, const std::string& transform,
                                             int blockSize, int checksum, uint64 originalSize, ThreadPool*, bool headerless, int bsVersion)
    : std::istream(is.rdbuf()), _is(is)
{
    init(tasks, entropy, transform, blockSize, checksum, originalSize, headerless, bsVersion);
}

CompressedInputStream::CompressedInputStream(std::istream& is, int tasks, const std::string& entropy, const std::string& transform,
                                             int blockSize, int checksum, uint64 originalSize, bool headerless, int bsVersion)
    : std::istream(is.rdbuf()), _is(is)
{
    init(tasks, entropy, transform, blockSize, checksum, originalSize, headerless, bsVersion);
}

bool CompressedInputStream::addListener(Listener<Event>& bl) { _listeners.push_back(&bl); return true; }

bool CompressedInputStream::removeListener(Listener<Event>& bl)
{
    auto it = std::find(_listeners.begin(), _listeners.end(), &bl);
    if (it == _listeners.end()) return false;
    _listeners.erase(it);
    return true;
}

void CompressedInputStream::init(int tasks, const std::string& entropy, const std::string& transform, int blockSize, int checksum, uint64 originalSize,
                                 bool headerless, int bsVersion)
{
    if ((tasks <= 0) || (tasks > 64)) throw std::invalid_argument("The number of jobs must be in [1..64], got " + std::to_string(tasks));
    _jobs = tasks; _blockSize = blockSize; _checksum = checksum; _outputSize = originalSize;
    _headless = headerless; _closed = false; _headerDone = false; _ended = false;
    _entropyType = 0; _transformType = 0; _bsVersion = 6;
    _hosted = 0;
    if (headerless) {
        // (io/CompressedInputStream.cpp:97-98 takes the caller's word for the version of a headerless stream)
        if (bsVersion < 0 || bsVersion > 6) throw std::invalid_argument("Invalid or missing bitstream version, cannot read this version of the stream");
        _bsVersion = bsVersion;
        if ((blockSize < 1024) || (blockSize > 1024 * 1024 * 1024) || ((blockSize & -16) != blockSize)) throw std::invalid_argument("Invalid block size");
        _entropyType = EntropyEncoderFactory::getType(entropy.c_str());
        _transformType = TransformFactory<byte>::getType(transform.c_str());
        _hosted = hostedStagesOf(_transformType, _hostIds);
        _spreadCopies = chainIsHostBound(_transformType);
    }
    _batchBlocks = std::max(tasks, 64);               // clamped to 256 MiB / 2 GiB once the block size is known
    const char* e = getenv("KNZ_BATCH_BLOCKS");
    _batchFromEnv = false;
    if (e && atoi(e) > 0) { _batchBlocks = atoi(e); _batchFromEnv = true; }
    _compBit = 0; _consumedBits = 0; _plainPos = 0; _gcount = 0; _srcEof = false;
    {
        const std::vector<int> devs = laneDevices();
        const std::vector<knz_ctx*> ctxs = openLanes(devs);
        _ps.resize(ctxs.size()); _prep.resize(ctxs.size());
        for (size_t i = 0; i < ctxs.size(); i++) {
            _ps[i].device = devs[i];
            _ps[i].ctx = ctxs[i]; _ps[i].buf = nullptr; _ps[i].cap = 0; _ps[i].len = 0; _ps[i].endBit = 0; _ps[i].consumedBits = 0; _ps[i].last = false; _ps[i].state = 0;
            _ps[i].dOut = nullptr; _ps[i].dOutCap = 0; _ps[i].ticket = 0;
            _prep[i].ctx = ctxs[i]; _prep[i].dIn = nullptr; _prep[i].dInCap = 0; _prep[i].stage = nullptr; _prep[i].stageCap = 0; _prep[i].inBytes = 0; _prep[i].startBit = 0;
            _prep[i].nb = 0; _prep[i].last = false; _prep[i].endBit = 0; _prep[i].consumedBits = 0; _prep[i].ticket = 0; _prep[i].state = 0;
        }
    }
    _pprod = 0;
    _activeIdx.clear();
    for (size_t i = 0; i < _ps.size(); i++) _activeIdx.push_back(int(i));
    _cons = 0; _rstop = false; _started = false; _cur = nullptr; _lastTaken = false; _tellBit = 0; _readBits = 0;
    _from = 1; _to = 0x7FFFFFFF; _nextBlockId = 1;
    for (auto& t : _tns) t = 0;
    _t0 = std::chrono::steady_clock::now();
    if (!headerless) _spreadCopies = false;          // until the header says what the chain is
    { _is.clear(); const std::streamoff at = std::streamoff(_is.tellg()); _originBit = (at < 0) ? 0 : 8 * int64(at); _is.clear(); }
}

CompressedInputStream::CompressedInputStream(std::istream& is, Context& ctx, bool headerless)
    : CompressedInputStream(is, ctx.getInt("jobs", 1), ctx.getString("entropy", "NONE"), ctx.getString("transform", "NONE"),
                            ctx.getInt("blockSize", 4 * 1024 * 1024), ctx.getInt("checksum", 0), uint64(ctx.getLong("outputSize", 0)),
                            static_cast<ThreadPool*>(nullptr), headerless, ctx.getInt("bsVersion", 6))
{
    setBlockRange(ctx.getInt("from", 1), ctx.getInt("to", 0x7FFFFFFF));
}

CompressedInputStream::~CompressedInputStream()
{
    stopReader();
    for (size_t i = 0; i < _ps.size(); i++) {
        g_devPool.put(_prep[i].ctx, _prep[i].dIn, _prep[i].dInCap);
        g_devPool.put(_ps[i].ctx, _ps[i].dOut, _ps[i].dOutCap);
        g_pinned.put(_ps[i].buf, _ps[i].cap); g_pinned.put(_prep[i].stage, _prep[i].stageCap);
    }
}

bool CompressedInputStream::fetch(size_t minBytes)
{
    // make at least minBytes available after the current byte position
    const size_t have = _comp.size() - size_t(_compBit >> 3);
    if (have >= minBytes) return true;
    if (_srcEof) return false;
    // KNZ_READ_AHEAD bytes at least per read (default 1 MiB). Large pieces (16 MiB, which the C API's file buffer reads with several
    // threads, FileInBuf::xsgetn) were measured and lost: the first batch waits for the whole piece (config 3 end to end: decompress
    // 6.9-7.9 -> 5.3-5.7 GB/s), and the source is not what a lane waits for afterwards
    static const size_t readAhead = []() { const char* e = getenv("KNZ_READ_AHEAD"); const long long v = e ? atoll(e) : 0; return v > 0 ? size_t(v) : (size_t(1) << 20); }();
    // (chains the device runs faster than a thread reads the source: 4 MiB pieces, which the C API's file buffer reads with the helper threads)
    size_t want = std::max<size_t>(minBytes - have, (_spreadCopies && getenv("KNZ_READ_AHEAD") == nullptr) ? (size_t(4) << 20) : readAhead);
    const size_t old = _comp.size();
    ScopedNs t_(_tns[0]);
    _comp.resize(old + want);
    _is.read(reinterpret_cast<char*>(&_comp[old]), std::streamsize(want));
    const size_t got = size_t(_is.gcount());
    _comp.resize(old + got);
    if (got < want) _srcEof = true;
    return (_comp.size() - size_t(_compBit >> 3)) >= minBytes;
}

void CompressedInputStream::readHeader()
{
    if (_headerDone) return;
    _headerDone = true;
    if (_headless) return;
    if (!fetch(17)) throw IOException("Invalid stream type", Error::ERR_INVALID_FILE);       // (17 bytes: the shortest old header; 20: the shortest current one)
    fetch(24);
    const size_t headerBytesSeen = _comp.size();
    _comp.resize(_comp.size() + 8, 0);           // reading margin for getBitsAt
    uint64 pos = _compBit;
    auto get = [&](uint n) { const uint64 v = getBitsAt(_comp, pos, n); pos += n; return v; };
    const bool enough24 = (_comp.size() - 8) >= 24;
    if (uint32_t(get(32)) != 0x4B414E5Au) throw IOException("Invalid stream type", Error::ERR_INVALID_FILE);
    const int bsVersion = int(get(4));
    if (bsVersion > 6) throw IOException("Invalid bitstream, cannot read this version of the stream", Error::ERR_STREAM_VERSION);
    if (bsVersion >= 6 && headerBytesSeen < 20) throw IOException("Invalid stream type", Error::ERR_INVALID_FILE);
    _bsVersion = bsVersion;
    // io/CompressedInputStream.cpp:541-558: two bits of checksum size since version 6, one checksum flag before
    uint64 ckSize;
    if (bsVersion >= 6) {
        ckSize = get(2);
        if (ckSize == 3) throw IOException("Invalid bitstream, incorrect block checksum size", Error::ERR_INVALID_FILE);
    } else {
        ckSize = get(1);
    }
    _checksum = int(32 * ckSize);
    _entropyType = short(get(5));
    try { EntropyEncoderFactory::getName(_entropyType); } catch (const std::invalid_argument&) { throw IOException("Invalid bitstream, unknown entropy type", Error::ERR_INVALID_CODEC); }
    _transformType = get(48);
    _hosted = hostedStagesOf(_transformType, _hostIds);
    _spreadCopies = chainIsHostBound(_transformType);
    try { TransformFactory<byte>::getName(_transformType); } catch (const std::invalid_argument&) { throw IOException("Invalid bitstream, unknown transform type", Error::ERR_INVALID_CODEC); }
    _blockSize = int(get(28) << 4);
    if ((_blockSize < 1024) || (_blockSize > 1024 * 1024 * 1024)) throw IOException("Invalid bitstream, incorrect block size", Error::ERR_BLOCK_SIZE);
    const int szMask = int(get(2));
    // (the whole header has to be there: 160 + 16 szMask bits since version 6, 136 + 16 szMask before)
    if (szMask != 0) {
        const size_t need = bsVersion >= 6 ? size_t(20 + 2 * szMask) : size_t((136 + 16 * szMask + 7) / 8);
        if ((!enough24 && szMask > 1 && bsVersion >= 6) || headerBytesSeen < need) throw IOException("Invalid stream type", Error::ERR_INVALID_FILE);
        _outputSize = get(uint(16 * szMask));
    }
    // :606-645: padding and 24 checksum bits since version 6; before, no padding, 16 bits, seeded with the bare version and
    // without the checksum size
    uint32_t ck1, ck2;
    if (bsVersion >= 6) {
        get(15);
        ck1 = uint32_t(get(24));
        ck2 = headerChecksum(uint32_t(ckSize), uint32_t(_entropyType), _transformType, uint32_t(_blockSize), szMask, _outputSize);
    } else {
        ck1 = uint32_t(get(16));
        const uint32_t HASH = 0x1E35A7BDu;
        uint32_t c = HASH * uint32_t(bsVersion);
        c ^= HASH * uint32_t(~uint32_t(_entropyType));
        c ^= HASH * uint32_t((~_transformType) >> 32);
        c ^= HASH * uint32_t(~_transformType);
        c ^= HASH * uint32_t(~uint32_t(_blockSize));
        if (szMask != 0) { c ^= HASH * uint32_t((~_outputSize) >> 32); c ^= HASH * uint32_t(~_outputSize); }
        ck2 = ((c >> 23) ^ (c >> 3)) & 0xFFFFu;
    }
    if (ck1 != ck2) throw IOException("Invalid bitstream, header checksum mismatch", Error::ERR_CRC_CHECK);
    _comp.resize(_comp.size() - 8);
    _consumedBits += pos - _compBit;
    _compBit = pos;
}

// reader thread: the compressed bytes of the next batch of blocks, on their way to the device (pr.last: nothing behind it)
void CompressedInputStream::prepareBatch(Prep& pr)
{
    pr.nb = 0; pr.inBytes = 0; pr.ticket = 0;
    if (_ended) { pr.last = true; pr.endBit = _originBit + int64(_compBit); pr.consumedBits = _consumedBits; return; }
    readHeader();
    // Drop the consumed prefix of the fetched bytes here, once per batch, before any bit cursor of the walk below
    // is taken: the walk keeps positions relative to _comp, so nothing may rebase them while it runs.
    if ((_compBit >> 3) > (size_t(1) << 20)) {   // keep 16-byte alignment of the remainder
        const size_t drop = size_t(_compBit >> 3) & ~size_t(15);
        _comp.dropFront(drop);
        _compBit -= uint64(drop) * 8;
        _originBit += int64(drop) * 8;
    }
    // walk the block length prefixes on the host (framing only) to find complete blocks
    uint64 pos = _compBit;
    int nb = 0;
    bool sawEnd = false;
    // one device call produces at most 2 GiB of output (32-bit positions on the device side)
    const int64_t bsz = int64_t(_blockSize > 0 ? _blockSize : 1);
    const int64_t lim = (int64_t(1) << 31) / bsz - 1;
    const int wantBatch = _batchBlocks.load();
    int batch = (wantBatch > lim) ? int(lim < 1 ? 1 : lim) : wantBatch;
    if (!_batchFromEnv.load()) batch = int(std::min<int64_t>(batch, std::max<int64_t>(1, (int64_t(24) << 20) / bsz)));
    if (_hosted) batch = 1;                                  // (the host undoes its stages block by block)
    while (nb < batch) {
        if (!fetch(size_t(((pos + 40) >> 3) + 1 - (_compBit >> 3)))) {
            if (uint64(_comp.size()) * 8 < pos + 8) { if (nb == 0) throw IOException("Unexpected end of stream", Error::ERR_READ_FILE); break; }
        }
        _comp.resize(_comp.size() + 8, 0);
        const uint lr = 3 + uint(getBitsAt(_comp, pos, 5));
        const uint64 len = getBitsAt(_comp, pos + 5, lr);
        _comp.resize(_comp.size() - 8);
        if (uint64(_comp.size()) * 8 < pos + 5 + lr) throw IOException("Unexpected end of stream", Error::ERR_READ_FILE);
        if (len == 0) { sawEnd = true; pos += 5 + lr; break; }
        if (len > (uint64(1) << 34)) throw IOException("Invalid block size", Error::ERR_BLOCK_SIZE);
        if (_nextBlockId >= int64(_to)) { sawEnd = true; break; }          // io/CompressedInputStream.cpp:866-868: the range is over
        const uint64 next = pos + 5 + lr + len;
        if (!fetch(size_t(((next + 7) >> 3) - (_compBit >> 3)))) throw IOException("Unexpected end of stream", Error::ERR_READ_FILE);
        if (_nextBlockId < int64(_from)) {
            // a block in front of the range: its bits are consumed, nothing is decoded (:843-865); only happens before the
            // first block of a batch, so the batch simply starts behind it
            _consumedBits += next - _compBit;
            _compBit = next;
            pos = next;
            _nextBlockId++;
            continue;
        }
        pos = next;
        nb++;
        _nextBlockId++;
    }
    if (nb > 0) {
        knz_ctx* c = pr.ctx;
        const size_t firstByte = size_t(_compBit >> 3) & ~size_t(15);
        const size_t lastByte = size_t((pos + 7) >> 3);
        const size_t inBytes = lastByte - firstByte;
        if (pr.dInCap < inBytes + 64) {
            g_devPool.put(c, pr.dIn, pr.dInCap);
            pr.dIn = nullptr; pr.dInCap = 0;
            pr.dIn = g_devPool.get(c, inBytes + 64 + (inBytes >> 2), &pr.dInCap);
        }
        // through page-locked staging: the pageable vector would be bounced by the runtime at a fraction of the PCIe rate
        if (pr.stageCap < inBytes) { g_pinned.put(pr.stage, pr.stageCap); pr.stage = nullptr; pr.stageCap = 0; pr.stage = g_pinned.get(inBytes, &pr.stageCap); }
        { ScopedNs t_(_tns[1]); parCopy(pr.stage, &_comp[firstByte], inBytes, _spreadCopies); }
        devCheck(c, knz_hip_memcpy_h2d_async(c, pr.dIn, pr.stage, inBytes, &pr.ticket), "h2d");
        pr.inBytes = inBytes;
        pr.startBit = _compBit - uint64(firstByte) * 8;
        pr.nb = nb;
    }
    _consumedBits += pos - _compBit;
    _compBit = pos;
    if (sawEnd) _ended = true;
    pr.endBit = _originBit + int64(_compBit);
    pr.consumedBits = _consumedBits;
    pr.last = sawEnd || nb == 0;
    if (hostTimeline()) fprintf(stderr, "[knz in batch] %d blocks, %zu B prepared (upload queued) at %.2f ms\n", nb, pr.inBytes, msSince(_t0));
}

// decoder thread: the kernels over a prepared batch; the plain bytes start their way back into `sl`
void CompressedInputStream::decodeBatch(Prep& pr, PSlot& sl)
{
    sl.len = 0; sl.ticket = 0;
    sl.endBit = pr.endBit; sl.consumedBits = pr.consumedBits; sl.last = pr.last;
    if (pr.nb == 0) return;
    knz_ctx* c = sl.ctx;
    knz_params p;
    memset(&p, 0, sizeof(p));
    p.transform_type = _transformType; p.entropy_type = _entropyType; p.block_size = _blockSize; p.checksum_bits = _checksum; p.jobs = _jobs;
    p.bs_version = _bsVersion == 0 ? 1 : _bsVersion;      // (0 is the C ABI's "unset = current"; the reference reads a version-0 stream as an old one)
    const size_t outCap = size_t(pr.nb) * size_t(_blockSize) + 64;
    // the slot is free, so the copy out of its device buffer (two batches ago) has been waited for
    if (sl.dOutCap < outCap) {
        g_devPool.put(c, sl.dOut, sl.dOutCap);
        sl.dOut = nullptr; sl.dOutCap = 0;
        sl.dOut = g_devPool.get(c, outCap, &sl.dOutCap);
    }
    { ScopedNs t_(_tns[2]); devCheck(c, knz_hip_copy_wait(c, pr.ticket), "h2d"); }
    pr.ticket = 0;
    uint64_t outBytes = 0, endBit = 0;
    const double tlUp = hostTimeline() ? msSince(_t0) : 0;
    GateHold gate_(gateOf(sl.device));
    const double tlGate = hostTimeline() ? msSince(_t0) : 0;
    struct TlEnd { bool on; double a, b; std::chrono::steady_clock::time_point t0; int nb;
                   ~TlEnd() { if (on) fprintf(stderr, "[knz in batch] %d blocks uploaded %.2f gate %.2f kernels-done (download queued) %.2f ms\n", nb, a, b, msSince(t0)); } } tlEnd{ hostTimeline(), tlUp, tlGate, _t0, pr.nb };
    ScopedNs t3_(_tns[3]);
    int64_t done = 0;
    if (_hosted) {
        // the device undoes its stages; the block comes back with its skip flags and the host undoes TEXT / UTF, last stage first
        uint32_t skip = 0xFF;
        uint64_t stored = 0;
        int32_t got = 0;
        devCheck(c, knz_hip_decode_block_hosted(c, &p, _hosted, static_cast<const uint8_t*>(pr.dIn), uint64(pr.inBytes) * 8, pr.startBit, static_cast<uint8_t*>(sl.dOut), outCap,
                                                &outBytes, &endBit, &skip, &stored, &got), "decode block");
        const size_t room = size_t(_blockSize) + 64;
        if (sl.cap < room) { g_pinned.put(sl.buf, sl.cap); sl.buf = nullptr; sl.cap = 0; sl.buf = g_pinned.get(room, &sl.cap); }
        std::vector<uint8_t> a(size_t(outBytes) + 64), b;
        if (outBytes) devCheck(c, knz_hip_memcpy_d2h(c, a.data(), sl.dOut, size_t(outBytes)), "d2h");
        const uint8_t* cur = a.data();
        int len = int(outBytes);
        std::vector<uint8_t>* spare = &b;
        std::vector<uint8_t>* held = &a;
        for (int i = _hosted - 1; i >= 0 && got; i--) {
            if ((skip >> (7 - i)) & 1u) continue;
            if (spare->size() < room) spare->resize(room);
            int outLen = 0;
            const bool ok = (_hostIds[i] == KNZ_T_TEXT)
                ? hoststage::textInverse(textVariantOfEntropy(_entropyType), cur, len, spare->data(), int(room), _blockSize, _bsVersion == 0 ? 1 : _bsVersion, &outLen)
                : hoststage::utfInverse(cur, len, spare->data(), int(room), &outLen);
            if (ok && outLen > _blockSize) throw IOException("Invalid data: block larger than the block size", Error::ERR_PROCESS_BLOCK);
            if (!ok) throw IOException("Invalid data: inverse transform failed", Error::ERR_PROCESS_BLOCK);
            cur = spare->data(); len = outLen;
            std::swap(spare, held);
        }
        if (_checksum && got) {
            const uint64 sum = hostChecksum(cur, len, _checksum);
            if (sum != (_checksum == 32 ? (stored & 0xFFFFFFFFull) : stored)) throw IOException("Corrupted bitstream: invalid block checksum", Error::ERR_CRC_CHECK);
        }
        memcpy(sl.buf, cur, size_t(len));
        sl.len = size_t(len);
        return;
    }
    devCheck(c, knz_hip_decode_blocks(c, &p, static_cast<const uint8_t*>(pr.dIn), uint64(pr.inBytes) * 8, pr.startBit, pr.nb,
                                      static_cast<uint8_t*>(sl.dOut), outCap, &outBytes, &endBit, &done), "decode blocks");
    if (sl.cap < size_t(outBytes)) { g_pinned.put(sl.buf, sl.cap); sl.buf = nullptr; sl.cap = 0; sl.buf = g_pinned.get(std::max(size_t(outBytes), outCap), &sl.cap); }
    sl.len = size_t(outBytes);
    if (outBytes) devCheck(c, knz_hip_memcpy_d2h_async(c, sl.buf, sl.dOut, size_t(outBytes), &sl.ticket), "d2h");
}


void CompressedInputStream::readerLoop()
{
    for (;;) {
        {
            std::unique_lock<std::mutex> l(_rmu);
            _rcv.wait(l, [&] { return _rstop || _prep[size_t(_activeIdx[size_t(_pprod)])].state == 0; });
            if (_rstop) return;
        }
        Prep& pr = _prep[size_t(_activeIdx[size_t(_pprod)])];
        pr.err = nullptr; pr.last = false;
        try {
            prepareBatch(pr);
        } catch (...) {
            pr.err = std::current_exception();
            pr.last = true; pr.nb = 0;
        }
        const bool last = pr.last;
        {
            std::lock_guard<std::mutex> l(_rmu);
            pr.state = 1;
            _pprod = (_pprod + 1) % int(_activeIdx.size());
        }
        _rcv.notify_all();
        if (last) return;
    }
}

void CompressedInputStream::decoderLoop(int lane)
{
    for (;;) {
        {
            std::unique_lock<std::mutex> l(_rmu);
            _rcv.wait(l, [&] { return _rstop || (_prep[size_t(lane)].state == 1 && _ps[size_t(lane)].state == 0); });
            if (_rstop) return;
        }
        Prep& pr = _prep[size_t(lane)];
        PSlot& sl = _ps[size_t(lane)];
        sl.err = nullptr; sl.last = false; sl.len = 0; sl.ticket = 0;
        if (pr.err) { sl.err = pr.err; pr.err = nullptr; sl.last = true; }
        else {
            try {
                decodeBatch(pr, sl);
            } catch (...) {
                sl.err = std::current_exception();
                sl.last = true; sl.len = 0;
            }
        }
        const bool last = sl.last;
        {
            std::lock_guard<std::mutex> l(_rmu);
            pr.state = 0;
            sl.state = 2;
        }
        _rcv.notify_all();
        if (last) return;
    }
}

// caller's thread: header first (its errors belong to the caller), then the reader thread
void CompressedInputStream::ensureStarted()
{
    if (_started) return;
    readHeader();
    _tellBit = _originBit + int64(_compBit);
    _readBits = _consumedBits;
    _rstop = false;
    _started = true;
    // lanes per device bounded by the block size, as in CompressedOutputStream: a lane holds a batch (at least one block), its output
    // and the inverse stages' scratch, so that with 1 GiB blocks one lane per GPU is what fits comfortably, with blocks up to 256 MiB
    // four. The block size is known now (header read, or given); lanes beyond the bound get no batches.
    {
        const size_t perDev = std::max<size_t>(1, (size_t(1) << 30) / size_t(std::max(_blockSize, 1)));
        std::map<int, size_t> seen;
        _activeIdx.clear();
        for (size_t i = 0; i < _ps.size(); i++) if (seen[_ps[i].device]++ < perDev) _activeIdx.push_back(int(i));
        if (_activeIdx.empty()) _activeIdx.push_back(0);
    }
    _reader = std::thread(&CompressedInputStream::readerLoop, this);
    _decoders.clear();
    for (int i : _activeIdx) _decoders.emplace_back(&CompressedInputStream::decoderLoop, this, i);
}

void CompressedInputStream::stopReader()
{
    bool any = _reader.joinable();
    for (std::thread& t : _decoders) any = any || t.joinable();
    if (any) {
        { std::lock_guard<std::mutex> l(_rmu); _rstop = true; }
        _rcv.notify_all();
        if (_reader.joinable()) _reader.join();
        for (std::thread& t : _decoders) if (t.joinable()) t.join();
        _decoders.clear();
    }
    // copies still on their way belong to batches nobody will look at: let them land before the buffers are reused or freed
    for (size_t i = 0; i < _ps.size(); i++) {
        knz_hip_copy_wait(_prep[i].ctx, _prep[i].ticket); _prep[i].ticket = 0;
        knz_hip_copy_wait(_ps[i].ctx, _ps[i].ticket); _ps[i].ticket = 0;
    }
    _started = false;
}

// caller's thread: hand back the slot just drained and take the next batch; false at the end of the stream
bool CompressedInputStream::advance()
{
    if (_cur) {
        { std::lock_guard<std::mutex> l(_rmu); _cur->state = 0; }
        _rcv.notify_all();
        _cur = nullptr;
    }
    _plainPos = 0;
    for (;;) {
        if (_lastTaken) return false;
        ensureStarted();
        PSlot* sl;
        {
            ScopedNs t_(_tns[4]);
            std::unique_lock<std::mutex> l(_rmu);
            _rcv.wait(l, [&] { return _ps[size_t(_activeIdx[size_t(_cons)])].state == 2; });
            sl = &_ps[size_t(_activeIdx[size_t(_cons)])];
            _cons = (_cons + 1) % int(_activeIdx.size());
        }
        if (sl->last) _lastTaken = true;
        if (sl->err) {
            std::exception_ptr e = sl->err;
            sl->err = nullptr;
            { std::lock_guard<std::mutex> l(_rmu); sl->state = 0; }
            _rcv.notify_all();
            std::rethrow_exception(e);
        }
        _tellBit = sl->endBit;
        _readBits = sl->consumedBits;
        if (sl->len != 0 && sl->ticket != 0) {               // the device-to-host copy the decoder thread queued
            int rc;
            { ScopedNs t_(_tns[5]); rc = knz_hip_copy_wait(sl->ctx, sl->ticket); }
            sl->ticket = 0;
            if (rc != 0) {
                { std::lock_guard<std::mutex> l(_rmu); sl->state = 0; }
                _rcv.notify_all();
                throw IOException("device to host copy failed", Error::ERR_READ_FILE);
            }
        }
        if (sl->len == 0) {
            { std::lock_guard<std::mutex> l(_rmu); sl->state = 0; }
            _rcv.notify_all();
            continue;
        }
        _cur = sl;
        if (hostTimeline()) fprintf(stderr, "[knz in batch] %zu B handed to the caller at %.2f ms\n", sl->len, msSince(_t0));
        return true;
    }
}

std::istream& CompressedInputStream::read(char* data, std::streamsize length)
{
    _gcount = 0;
    if (_closed) throw IOException("Stream closed", Error::ERR_READ_FILE);
    std::streamsize remaining = length;
    while (remaining > 0) {
        if (_cur == nullptr || _plainPos >= _cur->len) {
            if (!advance()) { setstate(std::ios::eofbit); break; }
        }
        const size_t take = std::min<size_t>(size_t(remaining), _cur->len - _plainPos);
        { ScopedNs t_(_tns[6]); parCopy(data + _gcount, _cur->buf + _plainPos, take, _spreadCopies); }
        _plainPos += take;
        _gcount += std::streamsize(take);
        remaining -= std::streamsize(take);
    }
    return *this;
}

int CompressedInputStream::peek()
{
    if (_cur == nullptr || _plainPos >= _cur->len) {
        if (!advance()) { setstate(std::ios::eofbit); return EOF; }
    }
    return int(_cur->buf[_plainPos]);
}

int CompressedInputStream::get()
{
    const int c = peek();
    if (c != EOF) { _plainPos++; _gcount = 1; } else _gcount = 0;
    return c;
}

int64 CompressedInputStream::tell()
{
    if (_closed) return -1;
    if (!_started && !_lastTaken) {           // the first block starts behind the stream header
        try { ensureStarted(); } catch (const IOException&) { _headerDone = false; return _originBit + int64(_compBit); }
    }
    return _tellBit;
}

bool CompressedInputStream::seek(int64 bitPos)
{
    if (_closed || bitPos < 0) return false;
    stopReader();
    _is.clear();
    _is.seekg(std::streampos(bitPos >> 3));
    if (_is.fail()) return false;
    // forget everything fetched or decoded; the stream parameters (header) stay
    _comp.clear();
    for (size_t i = 0; i < _ps.size(); i++) {
        _ps[i].state = 0; _ps[i].len = 0; _ps[i].err = nullptr; _ps[i].last = false;
        _prep[i].state = 0; _prep[i].nb = 0; _prep[i].err = nullptr; _prep[i].last = false;
    }
    _cons = 0; _pprod = 0; _cur = nullptr; _lastTaken = false;
    _plainPos = 0;
    _gcount = 0;
    _srcEof = false; _ended = false;
    _nextBlockId = 1;
    _originBit = (bitPos >> 3) * 8;
    _compBit = uint64(bitPos & 7);
    _tellBit = bitPos;
    this->clear();
    return true;
}

void CompressedInputStream::close()
{
    if (_closed) return;
    _closed = true;
    stopReader();
    if (hostTimeline()) fprintf(stderr, "[knz in] closed at %.2f ms\n", msSince(_t0));
    if (hostTiming())
        fprintf(stderr, "[knz in] lanes %zu: reader source-read %.2f ms, stage-copy %.2f | decoders upload-wait %.2f, kernels %.2f | caller wait-batch %.2f, download-wait %.2f, copy-out %.2f\n",
                _ps.size(), _tns[0] / 1e6, _tns[1] / 1e6, _tns[2] / 1e6, _tns[3] / 1e6, _tns[4] / 1e6, _tns[5] / 1e6, _tns[6] / 1e6);
    setstate(std::ios::eofbit);
}

}  // namespace kanzi_amd

// ================================================================================================
// C API (src/api/Compressor.cpp:183-358, src/api/Decompressor.cpp:108-313)
// ================================================================================================
using namespace kanzi_amd;

namespace {

// The C API's sink and source: the caller's FILE. Large pieces go through the descriptor in slices, pwrite / pread from the helper
// threads (the reference's FileOutputStream / FileInputStream work on the descriptor too, src/api/Compressor.cpp:226-229): a tmpfs
// or page-cache copy runs at a few GB/s per thread, which is below what the device delivers. Anything that is not a regular file
// opened without O_APPEND takes the stdio calls.
class FileOutBuf : public std::streambuf {
public:
    explicit FileOutBuf(FILE* f) : _f(f), _bulk(false)
    {
        const int fd = fileno(f);
        struct stat sb;
        if (fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) {
            const int fl = fcntl(fd, F_GETFL);
            _bulk = fl >= 0 && !(fl & O_APPEND);
        }
    }
protected:
    std::streamsize xsputn(const char* s, std::streamsize n) override
    {
        // KNZ_PAR_WRITE=1: slices through pwrite. Off by default: writers of ONE file take its inode lock in turn, so on tmpfs / ext4
        // four writers were half as fast as one fwrite (measured: 59 MB in 18.8 ms against 8.8 ms). Reads do scale (FileInBuf).
        static const bool parWrite = getenv("KNZ_PAR_WRITE") != nullptr && atoi(getenv("KNZ_PAR_WRITE")) != 0;
        const int parts = (_bulk && parWrite) ? parParts(size_t(n)) : 1;
        if (parts > 1 && fflush(_f) == 0) {
            const off_t at = ftello(_f);
            const int fd = fileno(_f);
            if (at >= 0) {
                std::atomic<bool> ok(true);
                const size_t total = size_t(n), slice = ((total / size_t(parts)) + 4095) & ~size_t(4095);
                HelperPool::get().run(parts, [&](int i) {
                    size_t a = std::min(total, size_t(i) * slice);
                    const size_t b = (i == parts - 1) ? total : std::min(total, a + slice);
                    while (a < b) {
                        const ssize_t w = pwrite(fd, s + a, b - a, at + off_t(a));
                        if (w <= 0) { ok = false; return; }
                        a += size_t(w);
                    }
                });
                if (!ok || fseeko(_f, at + off_t(total), SEEK_SET) != 0) return 0;
                return n;
            }
        }
        return std::streamsize(fwrite(s, 1, size_t(n), _f));
    }
    int_type overflow(int_type ch) override { if (ch == traits_type::eof()) return traits_type::not_eof(ch); const char c = char(ch); return fwrite(&c, 1, 1, _f) == 1 ? ch : traits_type::eof(); }
    int sync() override { return fflush(_f) == 0 ? 0 : -1; }
private:
    FILE* _f;
    bool _bulk;
};

class FileInBuf : public std::streambuf {
public:
    explicit FileInBuf(FILE* f) : _f(f), _bulk(false)
    {
        const int fd = fileno(f);
        struct stat sb;
        _bulk = fd >= 0 && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
    }
protected:
    std::streamsize xsgetn(char* s, std::streamsize n) override
    {
        static const bool parRead = !(getenv("KNZ_PAR_READ") != nullptr && atoi(getenv("KNZ_PAR_READ")) == 0);
        if (_bulk && parRead && size_t(n) >= PAR_MIN) {
            const off_t at = ftello(_f);                    // (accounts for what stdio holds in its buffer)
            const int fd = fileno(_f);
            struct stat sb;
            if (at >= 0 && fstat(fd, &sb) == 0 && sb.st_size > at) {
                const size_t total = std::min<size_t>(size_t(n), size_t(sb.st_size - at));
                const int parts = std::max(1, parParts(total));
                std::atomic<bool> ok(true);
                const size_t slice = ((total / size_t(parts)) + 4095) & ~size_t(4095);
                HelperPool::get().run(parts, [&](int i) {
                    size_t a = std::min(total, size_t(i) * slice);
                    const size_t b = (i == parts - 1) ? total : std::min(total, a + slice);
                    while (a < b) {
                        const ssize_t r = pread(fd, s + a, b - a, at + off_t(a));
                        if (r <= 0) { ok = false; return; }
                        a += size_t(r);
                    }
                });
                if (ok && fseeko(_f, at + off_t(total), SEEK_SET) == 0) return std::streamsize(total);
                if (fseeko(_f, at, SEEK_SET) != 0) return 0;      // a slice failed (the file shrank?): the plain way from where we were
            }
        }
        return std::streamsize(fread(s, 1, size_t(n), _f));
    }
    int_type underflow() override { const size_t r = fread(&_c, 1, 1, _f); if (r != 1) return traits_type::eof(); setg(&_c, &_c, &_c + 1); return traits_type::to_int_type(_c); }
private:
    FILE* _f; char _c;
    bool _bulk;
};

}  // namespace

struct cContext {
    CompressedOutputStream* pCos; FileOutBuf* buf; std::ostream* os; size_t blockSize;
};

struct dContext {
    CompressedInputStream* pCis; FileInBuf* buf; std::istream* is; size_t bufferSize;
};

extern "C" {

unsigned int getCompressorVersion(void) { return (1u << 16) | (0u << 8) | 0u; }
unsigned int getDecompressorVersion(void) { return (1u << 16) | (0u << 8) | 0u; }

int initCompressor(struct cData* pData, FILE* dst, struct cContext** pCtx)
{
    if ((pData == nullptr) || (pCtx == nullptr) || (dst == nullptr)) return Error::ERR_INVALID_PARAM;
    cContext* cctx = nullptr;
    try {
        if ((memchr(pData->transform, 0, sizeof(pData->transform)) == nullptr) || (memchr(pData->entropy, 0, sizeof(pData->entropy)) == nullptr))
            return Error::ERR_INVALID_PARAM;
        const std::string transform = TransformFactory<byte>::getName(TransformFactory<byte>::getType(pData->transform));
        const std::string entropy = EntropyEncoderFactory::getName(EntropyEncoderFactory::getType(pData->entropy));
        if ((transform.length() >= sizeof(pData->transform)) || (entropy.length() >= sizeof(pData->entropy))) return Error::ERR_INVALID_PARAM;
        memset(pData->transform, 0, sizeof(pData->transform));
        strncpy(pData->transform, transform.c_str(), sizeof(pData->transform) - 1);
        memset(pData->entropy, 0, sizeof(pData->entropy));
        strncpy(pData->entropy, entropy.c_str(), sizeof(pData->entropy) - 1);
        pData->blockSize = (pData->blockSize + 15) & size_t(-16);
        *pCtx = nullptr;
        size_t fileSize = 0;
        const int fd = fileno(dst);
        struct stat sbuf;
        if (fd >= 0 && fstat(fd, &sbuf) == 0) fileSize = size_t(sbuf.st_size);
        cctx = new cContext();
        cctx->buf = new FileOutBuf(dst);
        cctx->os = new std::ostream(cctx->buf);
        cctx->pCos = nullptr;
        // the call form of src/api/Compressor.cpp:230-237 (concurrent build: a null thread pool in front of `headerless`)
        cctx->pCos = new CompressedOutputStream(*cctx->os, int(pData->jobs), pData->entropy, pData->transform, int(pData->blockSize),
                                                pData->checksum, uint64(fileSize), nullptr, pData->headerless != 0);
        cctx->blockSize = pData->blockSize;
        *pCtx = cctx;
    } catch (const std::exception&) {
        if (cctx) { delete cctx->pCos; delete cctx->os; delete cctx->buf; delete cctx; }
        return Error::ERR_CREATE_COMPRESSOR;
    }
    return 0;
}

int compress(struct cContext* pCtx, const unsigned char* src, size_t inSize, size_t* outSize)
{
    if ((pCtx == nullptr) || (outSize == nullptr)) return Error::ERR_INVALID_PARAM;
    if ((src == nullptr) && (inSize != 0)) return Error::ERR_INVALID_PARAM;
    if (inSize > pCtx->blockSize) return Error::ERR_INVALID_PARAM;
    *outSize = 0;
    CompressedOutputStream* pCos = pCtx->pCos;
    if (pCos == nullptr) return Error::ERR_INVALID_PARAM;
    try {
        const uint64 w = pCos->getWritten();
        pCos->write(reinterpret_cast<const char*>(src), std::streamsize(inSize));
        *outSize = size_t(pCos->getWritten() - w);
        return pCos->good() ? 0 : int(Error::ERR_WRITE_FILE);
    } catch (const IOException& ioe) {
        return ioe.error();
    } catch (const std::exception&) {
        return Error::ERR_UNKNOWN;
    }
}

int disposeCompressor(struct cContext** ppCtx, size_t* outSize)
{
    if ((ppCtx == nullptr) || (*ppCtx == nullptr) || (outSize == nullptr)) return Error::ERR_INVALID_PARAM;
    *outSize = 0;
    cContext* pCtx = *ppCtx;
    int res = 0;
    try {
        if (pCtx->pCos != nullptr) {
            const uint64 w = pCtx->pCos->getWritten();
            pCtx->pCos->close();
            *outSize = size_t(pCtx->pCos->getWritten() - w);
        }
    } catch (const IOException& ioe) {
        res = ioe.error();
    } catch (const std::exception&) {
        res = Error::ERR_UNKNOWN;
    }
    delete pCtx->pCos;
    if (pCtx->os) pCtx->os->flush();
    delete pCtx->os;
    delete pCtx->buf;
    delete pCtx;
    *ppCtx = nullptr;
    return res;
}

int initDecompressor(struct dData* pData, FILE* src, struct dContext** pCtx)
{
    if ((pData == nullptr) || (pCtx == nullptr) || (src == nullptr)) return Error::ERR_INVALID_PARAM;
    if (pData->bufferSize > size_t(2) * 1024 * 1024 * 1024) return Error::ERR_INVALID_PARAM;
    dContext* dctx = nullptr;
    try {
        if ((pData->headerless != 0) && ((memchr(pData->transform, 0, sizeof(pData->transform)) == nullptr) ||
                                         (memchr(pData->entropy, 0, sizeof(pData->entropy)) == nullptr)))
            return Error::ERR_INVALID_PARAM;
        *pCtx = nullptr;
        dctx = new dContext();
        dctx->buf = new FileInBuf(src);
        dctx->is = new std::istream(dctx->buf);
        dctx->pCis = nullptr;
        if (pData->headerless != 0) {
            const std::string transform = TransformFactory<byte>::getName(TransformFactory<byte>::getType(pData->transform));
            const std::string entropy = EntropyEncoderFactory::getName(EntropyEncoderFactory::getType(pData->entropy));
            if ((transform.length() >= sizeof(pData->transform)) || (entropy.length() >= sizeof(pData->entropy))) {
                delete dctx->is; delete dctx->buf; delete dctx;
                return Error::ERR_INVALID_PARAM;
            }
            memset(pData->transform, 0, sizeof(pData->transform));
            strncpy(pData->transform, transform.c_str(), sizeof(pData->transform) - 1);
            memset(pData->entropy, 0, sizeof(pData->entropy));
            strncpy(pData->entropy, entropy.c_str(), sizeof(pData->entropy) - 1);
            pData->blockSize = (pData->blockSize + 15) & unsigned(-16);
            // the call form of src/api/Decompressor.cpp:159-166
            dctx->pCis = new CompressedInputStream(*dctx->is, int(pData->jobs), pData->entropy, pData->transform, int(pData->blockSize),
                                                   pData->checksum, uint64(pData->originalSize), nullptr, true, pData->bsVersion);
        } else {
            dctx->pCis = new CompressedInputStream(*dctx->is, int(pData->jobs));
        }
        dctx->bufferSize = pData->bufferSize;
        *pCtx = dctx;
    } catch (const std::exception&) {
        if (dctx) { delete dctx->pCis; delete dctx->is; delete dctx->buf; delete dctx; }
        return Error::ERR_CREATE_DECOMPRESSOR;
    }
    return 0;
}

int decompress(struct dContext* pCtx, unsigned char* dst, size_t* inSize, size_t* outSize)
{
    if ((pCtx == nullptr) || (outSize == nullptr)) return Error::ERR_INVALID_PARAM;
    if (*outSize > pCtx->bufferSize) return Error::ERR_INVALID_PARAM;
    if (*outSize == 0) return 0;
    if (dst == nullptr) return Error::ERR_INVALID_PARAM;
    if (inSize) *inSize = 0;
    CompressedInputStream* pCis = pCtx->pCis;
    if (pCis == nullptr) { *outSize = 0; return Error::ERR_INVALID_PARAM; }
    try {
        const uint64 r = pCis->getRead();
        pCis->read(reinterpret_cast<char*>(dst), std::streamsize(*outSize));
        if (!pCis->good() && !pCis->eof()) return Error::ERR_READ_FILE;
        if (inSize) *inSize = size_t(pCis->getRead() - r);
        *outSize = size_t(pCis->gcount());
    } catch (const IOException& ioe) {
        *outSize = 0;
        return ioe.error();
    } catch (const std::exception&) {
        *outSize = 0;
        return Error::ERR_UNKNOWN;
    }
    return 0;
}

int disposeDecompressor(struct dContext** ppCtx)
{
    if ((ppCtx == nullptr) || (*ppCtx == nullptr)) return Error::ERR_INVALID_PARAM;
    dContext* pCtx = *ppCtx;
    int res = 0;
    try { if (pCtx->pCis) pCtx->pCis->close(); } catch (const IOException& ioe) { res = ioe.error(); } catch (const std::exception&) { res = Error::ERR_UNKNOWN; }
    delete pCtx->pCis;
    delete pCtx->is;
    delete pCtx->buf;
    delete pCtx;
    *ppCtx = nullptr;
    return res;
}

}  // extern "C"
