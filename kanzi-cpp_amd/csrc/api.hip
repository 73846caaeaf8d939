// C-ABI layer (include/knz_hip.h): context, workspaces, and the batch drivers that chain the
// transform / entropy / bit-assembly kernels on one HIP stream.
#include "common.hpp"
#include "stages.hpp"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <algorithm>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace knz;

namespace knz {

const char* hipErrStr(hipError_t e) { return hipGetErrorString(e); }

struct WsBuf { void* p = nullptr; size_t cap = 0; };

struct ProfEntry { std::string name; hipEvent_t a, b; };

// A helper thread of a context that lives as long as the context (the parts of a BWT stage each need a host thread of their own:
// the suffix sort reads counters back between its rounds). Creating threads per call cost ~0.1 ms each on a 4-block batch.
struct Helper {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool has = false, done = true, quit = false;
    void loop()
    {
        for (;;) {
            std::function<void()> j;
            { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return has || quit; }); if (!has) return; j = std::move(job); has = false; }
            j();
            { std::lock_guard<std::mutex> l(m); done = true; }
            cv.notify_all();
        }
    }
    void run(std::function<void()> f)
    {
        if (!th.joinable()) th = std::thread(&Helper::loop, this);
        { std::lock_guard<std::mutex> l(m); job = std::move(f); has = true; done = false; }
        cv.notify_all();
    }
    void wait() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return done; }); }
    void stop()
    {
        if (!th.joinable()) return;
        { std::lock_guard<std::mutex> l(m); quit = true; }
        cv.notify_all();
        th.join();
    }
};

struct Ctx {
    int device = 0;
    Helper helpers[3];
    hipStream_t stream = nullptr;
    bool ownStream = false;
    char err[512] = { 0 };
    std::map<std::string, WsBuf> ws;
    bool profiling = false;
    std::vector<ProfEntry> prof;
    void* pinned = nullptr;      // small pinned host scratch
    size_t pinnedCap = 0;
    std::recursive_mutex mu;     // a context is one stream + one set of workspaces: calls on it are serialised
    std::mutex errMu;            // the last-error string is also written by the copy entry points, which do not take `mu`
    // Copy engine beside the kernels: one stream per direction, outside the lock above, so that a host thread can move the next
    // batch in (or the last one out) while another thread sits in knz_hip_encode_blocks / knz_hip_decode_blocks.
    hipStream_t stream2[3] = { nullptr, nullptr, nullptr };       // further streams for the split of the BWT stages
    hipEvent_t evFork = nullptr, evJoin[3] = { nullptr, nullptr, nullptr };
    std::mutex copyMu;
    hipStream_t copyIn = nullptr, copyOut = nullptr;
    std::vector<hipEvent_t> copyIdle;
    std::map<uint64_t, hipEvent_t> copyPending;
    uint64_t copyNext = 1;
};
#define CTX_LOCK(c) std::lock_guard<std::recursive_mutex> ctx_lock_((c)->mu)

static int fail(Ctx* c, int code, const char* fmt, ...)
{
    std::lock_guard<std::mutex> el(c->errMu);
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof(c->err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(c, -1, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

static int ws_get(Ctx* c, const char* name, size_t bytes, void** out)
{
    WsBuf& w = c->ws[name];
    if (w.cap < bytes) {
        if (w.p) HIPCHK(c, hipFree(w.p));
        w.p = nullptr; w.cap = 0;
        const size_t want = bytes + (bytes >> 3) + 4096;
        HIPCHK(c, hipMalloc(&w.p, want));
        w.cap = want;
    }
    *out = w.p;
    // KNZ_POISON_WS=1 (debugging aid): fill every workspace with a pattern on every request, so that a kernel that
    // relies on what an earlier call left behind fails deterministically instead of once in a few thousand runs
    static const int poison = getenv("KNZ_POISON_WS") ? atoi(getenv("KNZ_POISON_WS")) : 0;
    if (poison && bytes) HIPCHK(c, hipMemsetAsync(w.p, poison == 2 ? 0xFF : 0xA5, bytes, c->stream));
    return 0;
}

thread_local ProfHook* g_prof = nullptr;

struct CtxProf : ProfHook {
    Ctx* c; int idx = -1;
    explicit CtxProf(Ctx* ctx) : c(ctx) {}
    void begin(const char* name) override {
        ProfEntry e; e.name = name;
        hipEventCreate(&e.a); hipEventCreate(&e.b);
        hipEventRecord(e.a, c->stream);
        c->prof.push_back(e);
        idx = (int)c->prof.size() - 1;
    }
    void end() override { if (idx >= 0) hipEventRecord(c->prof[idx].b, c->stream); idx = -1; }
};

// Installs the hook for the duration of one API call when profiling is enabled.
struct ProfInstall {
    CtxProf hook;
    explicit ProfInstall(Ctx* c) : hook(c) { g_prof = c->profiling ? &hook : nullptr; }
    ~ProfInstall() { g_prof = nullptr; }
};

// memset / memcpy segments are timed under their own names
struct ProfScope {
    bool on;
    ProfScope(Ctx*, const char* name) : on(g_prof != nullptr) { if (on) g_prof->begin(name); }
    ~ProfScope() { if (on) g_prof->end(); }
};

static int count_transforms(uint64_t t, int* tok)
{
    int nb = 0;
    for (int i = 0; i < 8; i++) {
        const int v = (int)((t >> (42 - 6 * i)) & 63);
        if (v != 0 || i == 0) tok[nb++] = v;
    }
    return nb;
}

static bool host_stage_id(int t) { return t == KNZ_T_TEXT || t == KNZ_T_UTF; }
static bool transform_supported(int t) { return t == KNZ_T_NONE || t == KNZ_T_ZRLT || t == KNZ_T_MTFT || t == KNZ_T_BWT || t == KNZ_T_SRT || t == KNZ_T_RLT || t == KNZ_T_LZ || t == KNZ_T_LZX || t == KNZ_T_RANK || t == KNZ_T_TIMESTAMP; }
static bool entropy_supported(int e) { return e == KNZ_E_NONE || e == KNZ_E_ANS0 || e == KNZ_E_ANS1 || e == KNZ_E_HUFFMAN || e == KNZ_E_FPAQ; }

static int max_encoded_len(int t, int n)
{
    switch (t) {
    case KNZ_T_BWT: return n + 33;
    case KNZ_T_SRT: return n + 1024;
    case KNZ_T_RLT: return (n <= 512) ? n + 32 : n;
    case KNZ_T_LZ: case KNZ_T_LZX: return ((n <= 1024) ? n + 16 : n + n / 64) + 2;     // LZCodec.hpp:91-95
    case KNZ_T_UTF: return n + 8192;                                                   // UTFCodec.hpp:54 (a host stage: only its share of the chain's buffer size matters here)
    default: return n;
    }
}

}  // namespace knz

extern "C" {

int knz_hip_device_count(int* count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return e == hipSuccess ? 0 : -1;
}

int knz_hip_create(int device, void* stream, knz_ctx** out)
{
    *out = nullptr;
    Ctx* c = new Ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete c; return -1; }
    if (stream) { c->stream = (hipStream_t)stream; c->ownStream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return -1; }
        c->ownStream = true;
    }
    c->pinnedCap = 1 << 20;
    if (hipHostMalloc(&c->pinned, c->pinnedCap, hipHostMallocDefault) != hipSuccess) { delete c; return -1; }
    *out = reinterpret_cast<knz_ctx*>(c);
    return 0;
}

void knz_hip_destroy(knz_ctx* ctx)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return;
    for (Helper& h : c->helpers) h.stop();
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (auto& kv : c->ws) if (kv.second.p) hipFree(kv.second.p);
    for (auto& e : c->prof) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    if (c->pinned) hipHostFree(c->pinned);
    for (auto& kv : c->copyPending) { hipEventSynchronize(kv.second); hipEventDestroy(kv.second); }
    for (auto& e : c->copyIdle) hipEventDestroy(e);
    for (int k = 0; k < 3; k++) { if (c->stream2[k]) hipStreamDestroy(c->stream2[k]); if (c->evJoin[k]) hipEventDestroy(c->evJoin[k]); }
    if (c->evFork) hipEventDestroy(c->evFork);
    if (c->copyIn) hipStreamDestroy(c->copyIn);
    if (c->copyOut) hipStreamDestroy(c->copyOut);
    if (c->ownStream) hipStreamDestroy(c->stream);
    delete c;
}

static std::atomic<int>& bwt_split_knob()
{
    static std::atomic<int> v([] { const char* e = getenv("KNZ_BWT_SPLIT"); const int x = e ? atoi(e) : 3; return x < 1 ? 1 : (x > 4 ? 4 : x); }());
    return v;
}

// fewest blocks a part of a split BWT stage may have (KNZ_BWT_PART_MIN / knob "bwt_part_min"; see bwt_parts_wanted)
static std::atomic<int>& bwt_part_min_knob()
{
    static std::atomic<int> v([] { const char* e = getenv("KNZ_BWT_PART_MIN"); const int x = e ? atoi(e) : 2; return x < 1 ? 1 : x; }());
    return v;
}

// ranges of a batch that knz_hip_decode_blocks runs side by side on streams of their own (KNZ_DEC_PARTS / knob "dec_parts"; decode_impl)
static std::atomic<int>& dec_parts_knob()
{
    static std::atomic<int> v([] { const char* e = getenv("KNZ_DEC_PARTS"); const int x = e ? atoi(e) : 3; return x < 1 ? 1 : (x > 3 ? 3 : x); }());
    return v;
}

int knz_hip_tune(const char* name, int value)
{
    if (name == nullptr) return -1;
    if (!strcmp(name, "dec_parts")) { dec_parts_knob().store(value < 1 ? 1 : (value > 3 ? 3 : value)); return 0; }
    if (!strcmp(name, "bwt_part_min")) { bwt_part_min_knob().store(value < 1 ? 1 : value); return 0; }
    if (!strcmp(name, "mtf_tile")) return mtft_tune(value);
    if (!strcmp(name, "mtf_chain")) return mtft_tune_chain(value);
    if (!strcmp(name, "lz_serial_decode")) { lz_serial_decode(value ? 1 : 0); return 0; }
    if (!strcmp(name, "bwt_split")) { bwt_split_knob().store(value < 1 ? 1 : (value > 4 ? 4 : value)); return 0; }
    return bwt_forward_tune(name, value);
}

// (a copy taken under the lock the writers hold, into a buffer of the calling thread: lane contexts are shared between the stream
// classes' worker threads and the copy entry points run outside `mu`, so the context's own buffer may be rewritten while it is read)
const char* knz_hip_last_error(knz_ctx* ctx)
{
    if (!ctx) return "null context";
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    thread_local char copy[sizeof(c->err)];
    std::lock_guard<std::mutex> el(c->errMu);
    memcpy(copy, c->err, sizeof(copy));
    copy[sizeof(copy) - 1] = 0;
    return copy;
}

size_t knz_hip_encode_bound(const knz_params* p, size_t n)
{
    const size_t bs = (size_t)p->block_size;
    const size_t nb = (n + bs - 1) / bs + 1;
    // worst case: every symbol emits 16 bits (ANS) or 12 bits (Huffman) plus per-chunk headers
    size_t bound = 2 * n + nb * 64 + ((n / ENT_CHUNK) + nb) * (HDR_BYTES + 32) + 4096;
    // order-1 rANS writes 256 frequency tables per 4 MiB chunk (<= 3498 bits each), however small the block
    if (p->entropy_type == KNZ_E_ANS1) bound += (n / ANS1_CHUNK + nb) * 256 * (size_t)HDR_BYTES;
    return bound;
}

int knz_hip_set_profiling(knz_ctx* ctx, int enabled)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    CTX_LOCK(c);
    c->profiling = enabled != 0;
    for (auto& e : c->prof) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    c->prof.clear();
    return 0;
}

int knz_hip_get_kernel_times(knz_ctx* ctx, knz_kernel_time* out, int cap)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    CTX_LOCK(c);
    hipStreamSynchronize(c->stream);
    std::vector<std::string> order;
    std::map<std::string, std::pair<float, uint64_t>> agg;
    for (auto& e : c->prof) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess) ms = 0;
        if (!agg.count(e.name)) order.push_back(e.name);
        agg[e.name].first += ms;
        agg[e.name].second += 1;
    }
    int n = 0;
    for (auto& nm : order) {
        if (n >= cap) break;
        memset(&out[n], 0, sizeof(out[n]));
        snprintf(out[n].name, sizeof(out[n].name), "%s", nm.c_str());
        out[n].ms = agg[nm].first;
        out[n].launches = agg[nm].second;
        n++;
    }
    for (auto& e : c->prof) { hipEventDestroy(e.a); hipEventDestroy(e.b); }
    c->prof.clear();
    return n;
}

int knz_hip_malloc(knz_ctx* ctx, size_t bytes, void** d_ptr)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    CTX_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(d_ptr, bytes ? bytes : 1));
    return 0;
}

int knz_hip_free(knz_ctx* ctx, void* d_ptr)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    CTX_LOCK(c);
    HIPCHK(c, hipFree(d_ptr));
    return 0;
}

int knz_hip_memcpy_h2d(knz_ctx* ctx, void* d_dst, const void* src, size_t bytes)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    CTX_LOCK(c);
    HIPCHK(c, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int knz_hip_memcpy_d2h(knz_ctx* ctx, void* dst, const void* d_src, size_t bytes)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    CTX_LOCK(c);
    HIPCHK(c, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

static int copy_async(Ctx* c, void* dst, const void* src, size_t bytes, bool in, uint64_t* ticket)
{
    *ticket = 0;
    std::lock_guard<std::mutex> l(c->copyMu);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t& st = in ? c->copyIn : c->copyOut;
    if (st == nullptr) HIPCHK(c, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t ev;
    if (!c->copyIdle.empty()) { ev = c->copyIdle.back(); c->copyIdle.pop_back(); }
    else HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = bytes ? hipMemcpyAsync(dst, src, bytes, in ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, st) : hipSuccess;
    if (e == hipSuccess) e = hipEventRecord(ev, st);
    if (e != hipSuccess) {                                       // the event goes back to the idle list, the caller gets no ticket
        c->copyIdle.push_back(ev);
        return fail(c, -1, "asynchronous copy failed: %s", hipGetErrorString(e));
    }
    *ticket = c->copyNext++;
    c->copyPending[*ticket] = ev;
    return 0;
}

int knz_hip_memcpy_h2d_async(knz_ctx* ctx, void* d_dst, const void* src, size_t bytes, uint64_t* ticket)
{
    return copy_async(reinterpret_cast<Ctx*>(ctx), d_dst, src, bytes, true, ticket);
}

int knz_hip_memcpy_d2h_async(knz_ctx* ctx, void* dst, const void* d_src, size_t bytes, uint64_t* ticket)
{
    return copy_async(reinterpret_cast<Ctx*>(ctx), dst, d_src, bytes, false, ticket);
}

int knz_hip_copy_wait(knz_ctx* ctx, uint64_t ticket)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> l(c->copyMu);
        auto it = c->copyPending.find(ticket);
        if (it == c->copyPending.end()) return 0;               // unknown or already waited for
        ev = it->second;
        c->copyPending.erase(it);
    }
    const hipError_t e = hipEventSynchronize(ev);
    {
        std::lock_guard<std::mutex> l(c->copyMu);
        c->copyIdle.push_back(ev);
    }
    if (e != hipSuccess) return fail(c, -1, "copy failed: %s", hipGetErrorString(e));
    return 0;
}

int knz_hip_host_alloc(size_t bytes, void** ptr)
{
    *ptr = nullptr;
    return hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? 0 : -1;
}

int knz_hip_host_free(void* ptr) { return (ptr == nullptr || hipHostFree(ptr) == hipSuccess) ? 0 : -1; }

int knz_hip_sync(knz_ctx* ctx)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    CTX_LOCK(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int knz_hip_shift_bits(knz_ctx* ctx, const uint8_t* d_in, uint64_t nbits, uint32_t r, uint8_t* d_out)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (c == nullptr || d_in == nullptr || d_out == nullptr || r == 0 || r > 7) return KNZ_ERR_INVALID_PARAM;
    CTX_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    launch_shift_bits(c->stream, d_in, nbits, r, d_out);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// shared plumbing
// ------------------------------------------------------------------------------------------------
static int seq_required(const int* tok, int nTok, int n)
{
    int req = n;
    for (int i = 0; i < nTok; i++) { const int nx = max_encoded_len(tok[i], req); if (nx > req) req = nx; }
    return req;
}

struct SeqWs {
    SeqArrays a;
    u32* d_capEven; u32* d_capOdd;
    const u8** d_viewPtr;
    u8** d_entDst;
    u8* A; u8* B; u64 S;
    u32* scratch;
};

// (sfx: "" or the suffix of a decode lane -- the parts of a pipelined decode have workspaces of their own)
static int seq_alloc(Ctx* c, int nBlocks, u64 S, bool needAB, size_t scratchU32, SeqWs* w, const char* sfx = "")
{
    u8* base;
    const size_t nb = (size_t)nBlocks;
    const std::string nSmall = std::string("seqSmall") + sfx, nA = std::string("xfA") + sfx, nB = std::string("xfB") + sfx, nScr = std::string("xfScratch") + sfx;
    // one slab for all small per-block arrays
    const size_t bytes = nb * (5 * 1 + 4 * 6 + 8 * 4) + 1024;
    if (int r = ws_get(c, nSmall.c_str(), bytes + 256, (void**)&base)) return r;
    size_t off = 0;
    auto take = [&](size_t sz, size_t align) { off = (off + align - 1) & ~(align - 1); u8* p = base + off; off += sz; return p; };
    w->a.src = (const u8**)take(nb * 8, 16);
    w->a.dst = (u8**)take(nb * 8, 16);
    w->d_viewPtr = (const u8**)take(nb * 8, 16);
    w->d_entDst = (u8**)take(nb * 8, 16);
    w->a.len = (u32*)take(nb * 4, 16);
    w->a.alen = (u32*)take(nb * 4, 16);
    w->a.cap = (u32*)take(nb * 4, 16);
    w->a.newLen = (u32*)take(nb * 4, 16);
    w->d_capEven = (u32*)take(nb * 4, 16);
    w->d_capOdd = (u32*)take(nb * 4, 16);
    w->a.where = take(nb, 16);
    w->a.swaps = take(nb, 16);
    w->a.active = take(nb, 16);
    w->a.skip = take(nb, 16);
    w->a.ok = take(nb, 16);
    w->a.bufCap = w->d_capEven;
    w->a.dataCap = w->d_capOdd;
    w->S = S;
    w->A = w->B = nullptr;
    if (needAB) {
        if (int r = ws_get(c, nA.c_str(), (size_t)S * nb + 256, (void**)&w->A)) return r;
        if (int r = ws_get(c, nB.c_str(), (size_t)S * nb + 256, (void**)&w->B)) return r;
    }
    w->scratch = nullptr;
    if (scratchU32) { if (int r = ws_get(c, nScr.c_str(), scratchU32 * 4 + 64, (void**)&w->scratch)) return r; }
    return 0;
}

static size_t stage_scratch_u32(int t, int nBlocks, u32 maxLen, bool forward = true)
{
    switch (t) {
    case KNZ_T_ZRLT: return zrlt_scratch_u32(nBlocks, maxLen);
    case KNZ_T_MTFT: return mtft_scratch_u32(nBlocks, maxLen);
    case KNZ_T_SRT: return forward ? srt_scratch_u32(nBlocks, maxLen) : srt_inverse_scratch_u32(nBlocks, maxLen);
    default: return 0;
    }
}

// The BWT stages of a batch in several parts at once. A suffix sort (and the inverse's list ranking) is a long sequence of launches
// with host read-backs in between and rounds that occupy a fraction of the CUs; the blocks are independent, so the batch is cut
// into KNZ_BWT_SPLIT (default 3, 1 = off, at most 4; 26 blocks of 8 MiB: 63.7 / 60.4 / 59.9 / 60.3 ms per step with 1 / 2 / 3 / 4) runs of blocks: the caller's thread drives the first on the context's stream,
// helper threads drive the others on streams of their own, each with its own scratch and read-back area. Not while per-kernel
// timing is on (the timing hooks belong to the caller's thread).
static int bwt_parts_wanted(const Ctx* c, int nBlocks)
{
    if (c->profiling) return 1;
    int parts = bwt_split_knob().load();
    const int least = bwt_part_min_knob().load();
    while (parts > 1 && nBlocks < least * parts) parts--;        // at least `least` blocks per part
    return parts;
}

static int bwt_in_parts(Ctx* c, hipStream_t s, const XfStage& st, int parts, const std::function<size_t(int)>& scratchBytes,
                        const std::function<int(hipStream_t, const XfStage&, void*, size_t, u32*)>& launch, const char* what, bool needThreads)
{
    static const char* const wsName[4] = { "bwtScratch", "bwtScratch2", "bwtScratch3", "bwtScratch4" };
    XfStage part[4];
    void* sc[4];
    size_t bytes[4];
    int first = 0;
    for (int k = 0; k < parts; k++) {
        const int nb = st.nBlocks / parts + (k < st.nBlocks % parts ? 1 : 0);
        part[k] = st;
        part[k].nBlocks = nb;
        part[k].src += first; part[k].dst += first; part[k].len += first; part[k].cap += first; part[k].ok += first; part[k].newLen += first;
        first += nb;
        bytes[k] = scratchBytes(nb);
        if (int r = ws_get(c, wsName[k], bytes[k], &sc[k])) return r;
    }
    HIPCHK(c, hipSetDevice(c->device));                          // the extra streams belong to the context's device
    for (int k = 1; k < parts; k++)
        if (c->stream2[k - 1] == nullptr) HIPCHK(c, hipStreamCreateWithFlags(&c->stream2[k - 1], hipStreamNonBlocking));
    if (c->evFork == nullptr) {
        HIPCHK(c, hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
        for (int k = 0; k < 3; k++) HIPCHK(c, hipEventCreateWithFlags(&c->evJoin[k], hipEventDisableTiming));
    }
    HIPCHK(c, hipEventRecord(c->evFork, s));                     // the other streams start behind what is queued on the first
    for (int k = 1; k < parts; k++) HIPCHK(c, hipStreamWaitEvent(c->stream2[k - 1], c->evFork, 0));
    int rc[4] = { 0, 0, 0, 0 };
    if (!needThreads) {
        // a stage that never waits for the device (the inverse): the caller's thread queues every part on its stream
        for (int k = 0; k < parts; k++) rc[k] = launch(k ? c->stream2[k - 1] : s, part[k], sc[k], bytes[k], reinterpret_cast<u32*>(c->pinned) + 32768 * k);
    } else {
        try {
            for (int k = 1; k < parts; k++)
                c->helpers[k - 1].run([&, k] {
                    if (hipSetDevice(c->device) != hipSuccess) { rc[k] = -1; return; }
                    rc[k] = launch(c->stream2[k - 1], part[k], sc[k], bytes[k], reinterpret_cast<u32*>(c->pinned) + 32768 * k);   // own read-back area (the context's is 1 MiB)
                });
        } catch (...) {
            for (int k = 1; k < parts; k++) c->helpers[k - 1].wait();
            return fail(c, -1, "%s: cannot start a helper thread", what);
        }
        rc[0] = launch(s, part[0], sc[0], bytes[0], reinterpret_cast<u32*>(c->pinned));
        for (int k = 1; k < parts; k++) c->helpers[k - 1].wait();
    }
    for (int k = 0; k < parts; k++) if (rc[k] != 0) return fail(c, -1, "%s failed: %s", what, hipGetErrorString(hipGetLastError()));
    for (int k = 1; k < parts; k++) {                            // and the first stream continues behind the others
        HIPCHK(c, hipEventRecord(c->evJoin[k - 1], c->stream2[k - 1]));
        HIPCHK(c, hipStreamWaitEvent(s, c->evJoin[k - 1], 0));
    }
    return 0;
}

static int run_forward_stage(Ctx* c, hipStream_t s, int t, const XfStage& st)
{
    switch (t) {
    case KNZ_T_ZRLT: launch_zrlt_forward(s, st); break;
    case KNZ_T_MTFT: launch_mtft_forward(s, st); break;
    case KNZ_T_SRT: launch_srt_forward(s, st); break;
    case KNZ_T_RLT: launch_rlt_forward(s, st); break;
    case KNZ_T_RANK: launch_sbrt_forward(s, st, 2); break;
    case KNZ_T_TIMESTAMP: launch_sbrt_forward(s, st, 3); break;
    case KNZ_T_LZ: case KNZ_T_LZX: {
        const size_t bytes = lz_forward_scratch_bytes(t, st.nBlocks, st.maxLen);
        void* sc;
        if (int r = ws_get(c, "lzScratch", bytes, &sc)) return r;
        if (launch_lz_forward(s, st, t, sc, bytes) != 0) return fail(c, -1, "LZ forward failed: %s", hipGetErrorString(hipGetLastError()));
        break;
    }
    case KNZ_T_BWT: {
        if (const int parts = bwt_parts_wanted(c, st.nBlocks); parts > 1)
            return bwt_in_parts(c, s, st, parts, [&](int nb) { return bwt_forward_scratch_bytes(nb, st.maxLen, (size_t)nb * st.maxLen); },
                                 [](hipStream_t q, const XfStage& h, void* sc, size_t bytes, u32* pin) { return launch_bwt_forward(q, h, sc, bytes, pin); }, "BWT forward", true);
        const size_t bytes = bwt_forward_scratch_bytes(st.nBlocks, st.maxLen, (size_t)st.nBlocks * st.maxLen);
        void* sc;
        if (int r = ws_get(c, "bwtScratch", bytes, &sc)) return r;
        if (launch_bwt_forward(s, st, sc, bytes, reinterpret_cast<u32*>(c->pinned)) != 0) return fail(c, -1, "BWT forward failed: %s", hipGetErrorString(hipGetLastError()));
        break;
    }
    default: break;
    }
    return 0;
}

// lane >= 0: one part of a pipelined decode (decode_impl): the BWT inverse of the part is not split further and uses the scratch and the
// read-back area of split part `lane`
static int run_inverse_stage(Ctx* c, hipStream_t s, int t, const XfStage& st, int lane = -1)
{
    switch (t) {
    case KNZ_T_ZRLT: launch_zrlt_inverse(s, st); break;
    case KNZ_T_MTFT: launch_mtft_inverse(s, st); break;
    case KNZ_T_SRT: launch_srt_inverse(s, st); break;
    case KNZ_T_RLT: launch_rlt_inverse(s, st); break;
    case KNZ_T_RANK: launch_sbrt_inverse(s, st, 2); break;
    case KNZ_T_TIMESTAMP: launch_sbrt_inverse(s, st, 3); break;
    case KNZ_T_LZ: case KNZ_T_LZX: {
        void* sc = nullptr;
        size_t bytes = 0;
        if (st.maxCap != 0 && st.maxCap <= (1u << 30) && !lz_serial_decode(-1)) {
            // about 20 bytes per output byte; when the device cannot spare that (1 GiB blocks, a 2 GiB batch), or the workspace would
            // exceed the budget below, the blocks are decoded by the one-wave-per-block decoder, which needs no scratch
            bytes = lz_inverse_scratch_bytes(st.nBlocks, st.maxCap);
            static const size_t budget = getenv("KNZ_LZ_INV_SCRATCH_MAX") ? (size_t)atoll(getenv("KNZ_LZ_INV_SCRATCH_MAX")) : ((size_t)48 << 30);
            WsBuf& wb = c->ws["lzInvScratch"];
            if (bytes > budget) { bytes = 0; }
            else if (wb.cap >= bytes) sc = wb.p;
            else {
                if (wb.p) { HIPCHK(c, hipFree(wb.p)); wb.p = nullptr; wb.cap = 0; }
                const size_t want = bytes + (bytes >> 3) + 4096;
                void* p = nullptr;
                if (hipMalloc(&p, want) == hipSuccess) { wb.p = p; wb.cap = want; sc = p; }
                else { (void)hipGetLastError(); bytes = 0; }           // out of memory: the serial decoder
            }
        }
        launch_lz_inverse(s, st, sc, bytes, st.maxCap);
        break;
    }
    case KNZ_T_BWT: {
        if (lane < 0) {
            if (const int parts = bwt_parts_wanted(c, st.nBlocks); parts > 1)
                return bwt_in_parts(c, s, st, parts, [&](int nb) { return bwt_inverse_scratch_bytes(nb, st.maxLen, (size_t)nb * st.maxLen); },
                                     [](hipStream_t q, const XfStage& h, void* sc, size_t bytes, u32* pin) { return launch_bwt_inverse(q, h, sc, bytes, pin); }, "BWT inverse", false);
        }
        static const char* const wsName[4] = { "bwtScratch", "bwtScratch2", "bwtScratch3", "bwtScratch4" };
        const int k = lane < 0 ? 0 : (lane & 3);
        const size_t bytes = bwt_inverse_scratch_bytes(st.nBlocks, st.maxLen, (size_t)st.nBlocks * st.maxLen);
        void* sc;
        if (int r = ws_get(c, wsName[k], bytes, &sc)) return r;
        if (launch_bwt_inverse(s, st, sc, bytes, reinterpret_cast<u32*>(c->pinned) + 32768 * k) != 0) return fail(c, -1, "BWT inverse failed: %s", hipGetErrorString(hipGetLastError()));
        break;
    }
    default: break;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------------------
// capsMode: 0 = reference stream buffers (jobs model), otherwise every destination capacity = capsMode (per-stage API)
static int encode_impl(Ctx* c, const knz_params* p, const uint8_t* d_in, size_t n, const uint8_t* prologue,
                       uint32_t prologueBits, int framing, int finish, int64_t firstBlock, uint8_t* d_out, size_t outCap,
                       uint64_t* outBits, const knz_host_stages* hs = nullptr)
{
    ProfInstall pi_(c);
    HIPCHK(c, hipSetDevice(c->device));
    if ((reinterpret_cast<uintptr_t>(d_in) & 15) || (reinterpret_cast<uintptr_t>(d_out) & 15))
        return fail(c, KNZ_ERR_INVALID_PARAM, "device buffers must be 16-byte aligned");
    const u32 bs = (u32)p->block_size;
    if (framing && (bs < 1024 || bs > (1u << 30) || (bs & 15))) return fail(c, KNZ_ERR_INVALID_PARAM, "invalid block size %u", bs);
    int tok[8];
    const int nTok = count_transforms(p->transform_type, tok);
    const int nHosted = hs ? hs->stages : 0;
    if (nHosted < 0 || nHosted > nTok) return fail(c, KNZ_ERR_INVALID_PARAM, "host stage count %d does not fit the chain", nHosted);
    for (int i = 0; i < nTok; i++) {
        if (i < nHosted) { if (!host_stage_id(tok[i])) return fail(c, KNZ_ERR_INVALID_CODEC, "transform id %d is no host stage", tok[i]); }
        else if (!transform_supported(tok[i])) return fail(c, KNZ_ERR_INVALID_CODEC, "transform id %d not implemented on device", tok[i]);
    }
    if (!entropy_supported(p->entropy_type)) return fail(c, KNZ_ERR_INVALID_CODEC, "entropy id %d not implemented on device", p->entropy_type);
    if (p->checksum_bits != 0 && p->checksum_bits != 32 && p->checksum_bits != 64) return fail(c, KNZ_ERR_INVALID_PARAM, "checksum must be 0, 32 or 64");
    hipStream_t s = c->stream;

    const int nBlocks = (n == 0) ? 0 : (int)((n + bs - 1) / bs);
    if (hs) {
        // one block, in the length the host stages left it (never longer than the block it came from)
        if (nBlocks != 1 || hs->orig_len > bs || n > hs->orig_len + 8192u) return fail(c, KNZ_ERR_INVALID_PARAM, "a hosted call takes exactly one block");
        if ((hs->orig_len <= 15) != (n <= 15) || (hs->orig_len <= 15 && hs->applied_mask)) return fail(c, KNZ_ERR_INVALID_PARAM, "copy blocks go through no stage");
    }
    // device-side positions of a batch are 32-bit (suffix array slots, bit offsets inside staging areas): a batch
    // is limited to 2 GiB of input; the host layers split larger inputs into several calls
    if (n > (size_t)0x7FFFFFFF - 8ull * (size_t)(nBlocks + 1) * 1056) return fail(c, KNZ_ERR_INVALID_PARAM, "batch of %zu bytes exceeds the 2 GiB per-call limit", n);
    if ((prologueBits + 7) / 8 > 200) return fail(c, KNZ_ERR_INVALID_PARAM, "prologue too long");
    const size_t needOut = ((size_t)prologueBits + 7) / 8 + 16;
    if (outCap < needOut) return fail(c, KNZ_ERR_WRITE_FILE, "output buffer too small");
    u64* d_total;
    if (int r = ws_get(c, "total", 64, (void**)&d_total)) return r;

    if (nBlocks == 0) {
        HIPCHK(c, hipMemsetAsync(d_out, 0, needOut, s));
        if (prologueBits) {
            u8* d_pro;
            if (int r = ws_get(c, "prologue", 256, (void**)&d_pro)) return r;
            HIPCHK(c, hipMemcpyAsync(d_pro, prologue, (prologueBits + 7) / 8, hipMemcpyHostToDevice, s));
            launch_put_prologue(s, reinterpret_cast<u32*>(d_out), d_pro, prologueBits);
        }
        HIPCHK(c, hipStreamSynchronize(s));
        if (outBits) *outBits = (u64)prologueBits + ((framing && finish) ? 8 : 0);
        return 0;
    }

    // ---- block bookkeeping
    u32 *d_origLen; BlockInfo* d_info;
    if (int r = ws_get(c, "origLen", sizeof(u32) * nBlocks, (void**)&d_origLen)) return r;
    if (int r = ws_get(c, "info", sizeof(BlockInfo) * nBlocks, (void**)&d_info)) return r;
    const int maxIn = (int)((n < bs) ? n : bs);
    const int required = seq_required(tok, nTok, maxIn);
    const u64 S = ((u64)required + 255) & ~255ull;
    bool realStages = false;
    size_t scratch = 0;
    for (int i = 0; i < nTok; i++) if (tok[i] != KNZ_T_NONE) { realStages = true; const size_t q = (i < nHosted) ? 0 : stage_scratch_u32(tok[i], nBlocks, (u32)S); if (q > scratch) scratch = q; }
    SeqWs w;
    if (int r = seq_alloc(c, nBlocks, S, realStages, scratch, &w)) return r;
    w.a.origLen = d_origLen;
    const bool direct = !realStages && !p->checksum_bits;      // NullTransforms only: one bookkeeping launch
    if (!direct) launch_init_blocks(s, n, bs, nBlocks, d_origLen, w.a.len);
    // block checksums of the ORIGINAL bytes (io/CompressedOutputStream.cpp:675-682)
    u64* d_sums = nullptr;
    if (p->checksum_bits) {
        if (int r = ws_get(c, "sums", sizeof(u64) * nBlocks, (void**)&d_sums)) return r;
        if (hs) {
            // (the checksum is the ORIGINAL block's: the host computed it before its stages ran)
            u64* hsum = reinterpret_cast<u64*>(c->pinned) + 512;
            *hsum = hs->checksum;
            HIPCHK(c, hipMemcpyAsync(d_sums, hsum, sizeof(u64), hipMemcpyHostToDevice, s));
        } else {
            launch_block_ptrs(s, d_in, bs, nBlocks, w.d_viewPtr);
            launch_xxhash(s, w.d_viewPtr, d_origLen, nBlocks, p->checksum_bits, d_sums);
        }
    }

    // destination capacities the reference would present (io/CompressedOutputStream.cpp:141,461-462,733-739):
    // block i runs on buffer slot i % jobs; "data" of slot 0 is max(bs + bs/8, 256 KiB), of the others
    // max(bs + bs/64, 64 KiB); "buffer" grows to the largest requiredSize seen on the slot.
    if (realStages) {
        const int jobs = (p->jobs <= 0) ? 1 : (p->jobs > 64 ? 64 : p->jobs);
        if ((size_t)nBlocks * 8 > c->pinnedCap) return fail(c, KNZ_ERR_INVALID_PARAM, "too many blocks in one batch (%d)", nBlocks);
        u32* h = reinterpret_cast<u32*>(c->pinned);
        u32* hEven = h; u32* hOdd = h + nBlocks;
        std::vector<u32> slotBuf((size_t)jobs, 0);
        for (int64_t b = 0; b < nBlocks; b++) {
            const int64_t gid = firstBlock + b;
            const int slot = (int)(gid % jobs);
            const u32 len = hs ? hs->orig_len : (u32)(((size_t)(b + 1) * bs <= n) ? bs : n - (size_t)b * bs);      // (the reference sizes its buffers by the block as read)
            const u32 req = (u32)seq_required(tok, nTok, (int)len);
            // a slot's buffer is at least what a full block needed earlier on that slot
            u32 bufc = slotBuf[slot];
            if (gid >= jobs) { const u32 full = (u32)seq_required(tok, nTok, (int)bs); if (bufc < full) bufc = full; }
            if (bufc < req) bufc = req;
            slotBuf[slot] = bufc;
            u32 datac = (slot == 0) ? std::max(bs + (bs >> 3), 256u * 1024u) : std::max(bs + (bs >> 6), 65536u);
            if (!framing) { bufc = (u32)p->jobs; datac = (u32)p->jobs; }      // per-stage API: explicit capacity
            hEven[b] = framing ? std::max(bufc, req) : bufc;
            hOdd[b] = framing ? std::max(datac, req) : datac;
        }
        HIPCHK(c, hipMemcpyAsync(w.d_capEven, hEven, sizeof(u32) * nBlocks, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(w.d_capOdd, hOdd, sizeof(u32) * nBlocks, hipMemcpyHostToDevice, s));
        HIPCHK(c, hipStreamSynchronize(s));     // pinned scratch is reused below
    }

    // ---- transform stages
    if (direct) launch_seq_fwd_direct(s, w.a, d_origLen, n, bs, nBlocks, nTok, d_in, w.d_viewPtr);
    for (int i = 0; i < nTok && !direct; i++) {
        launch_seq_fwd_prepare(s, w.a, nBlocks, i, d_in, bs, w.A, w.B, S);
        if (i < nHosted) {
            launch_seq_fwd_hosted(s, w.a, nBlocks, i, (hs->applied_mask >> i) & 1u);
            continue;
        }
        if (tok[i] == KNZ_T_NONE) {
            launch_seq_fwd_null(s, w.a, nBlocks, i);
            continue;
        }
        XfStage st;
        st.src = w.a.src; st.dst = w.a.dst; st.len = w.a.alen; st.cap = w.a.cap; st.ok = w.a.ok; st.newLen = w.a.newLen;
        st.nBlocks = nBlocks; st.maxLen = (u32)S; st.scratchU32 = w.scratch; st.entropyType = p->entropy_type;
        if (int r = run_forward_stage(c, s, tok[i], st)) return r;
        launch_seq_fwd_commit(s, w.a, nBlocks, i);
    }
    if (!direct) launch_seq_fwd_finish(s, w.a, nBlocks, d_in, bs, w.A, w.B, S, w.d_viewPtr);
    BlockView view;
    view.ptr = w.d_viewPtr; view.len = w.a.len;
    u32* d_blockLen = w.a.len;
    u8* d_skip = w.a.skip;

    // ---- entropy stage
    const bool ans1 = (p->entropy_type == KNZ_E_ANS1);
    const u32 entChunk = (p->entropy_type == KNZ_E_FPAQ || ans1) ? (4u << 20) : ENT_CHUNK;
    const u32 slotMul = ans1 ? ANS1_SLOTS : 1u;
    u32 hdrStride = TMP_STRIDE;
    const int chunksPerBlock = (int)((S + entChunk - 1) / entChunk);
    const int maxChunks = chunksPerBlock * (int)slotMul;
    const size_t nSlots = (size_t)nBlocks * maxChunks;
    ChunkDesc* d_desc; u8* d_tmp; uint2* d_encTab;
    if (int r = ws_get(c, "desc", sizeof(ChunkDesc) * nSlots, (void**)&d_desc)) return r;
    if (p->entropy_type == KNZ_E_ANS0) {
        if (int r = ws_get(c, "chunkTmp", (size_t)TMP_STRIDE * nSlots, (void**)&d_tmp)) return r;
        if (int r = ws_get(c, "encTab", sizeof(uint2) * 256 * nSlots, (void**)&d_encTab)) return r;
        launch_ans0_encode(s, view, nBlocks, maxChunks, d_desc, d_encTab, d_tmp);
    } else if (ans1) {
        const size_t nCh = (size_t)nBlocks * chunksPerBlock;
        Ans1EncWs aw;
        aw.payStride = (2ull * std::min<u64>(S, ANS1_CHUNK) + 511) & ~255ull;
        if (int r = ws_get(c, "ans1Hist", ans1_hist_bytes(nCh), (void**)&aw.hist)) return r;
        if (int r = ws_get(c, "ans1EncTab", ans1_enctab_bytes(nCh), (void**)&aw.encTab)) return r;
        if (int r = ws_get(c, "chunkTmp", (size_t)HDR_BYTES * nSlots, (void**)&d_tmp)) return r;
        if (int r = ws_get(c, "ans1Pay", (size_t)aw.payStride * nCh, (void**)&aw.pay)) return r;
        aw.hdr = d_tmp;
        hdrStride = HDR_BYTES;
        launch_ans1_encode(s, view, nBlocks, chunksPerBlock, d_desc, aw);
    } else if (p->entropy_type == KNZ_E_HUFFMAN) {
        if (int r = ws_get(c, "chunkTmp", (size_t)TMP_STRIDE * nSlots, (void**)&d_tmp)) return r;
        launch_huffman_encode(s, view, nBlocks, maxChunks, d_desc, d_tmp);
    } else if (p->entropy_type == KNZ_E_FPAQ) {
        const u64 fStride = (4u << 20) + (4u << 17) + 256;      // FPAQEncoder.cpp:65-68 buffer size (+ slack)
        if (int r = ws_get(c, "chunkTmp", (size_t)fStride * nSlots, (void**)&d_tmp)) return r;
        u16* d_fprobs;
        if (int r = ws_get(c, "fpaqProbs", fpaq_probs_bytes(nBlocks, S), (void**)&d_fprobs)) return r;
        launch_fpaq_encode(s, view, d_origLen, framing ? 15u : 0u, nBlocks, maxChunks, d_desc, d_tmp, fStride, d_fprobs, S);
    } else {
        if (int r = ws_get(c, "chunkTmp", 64, (void**)&d_tmp)) return r;
        launch_none_encode(s, view, nBlocks, maxChunks, d_desc);
    }

    // ---- framing + assembly
    FrameParams fp;
    fp.framing = framing; fp.nTransforms = nTok; fp.checksumBits = p->checksum_bits; fp.finish = finish; fp.prologueBits = prologueBits;
    launch_block_sum(s, d_desc, d_info, d_blockLen, nBlocks, maxChunks, entChunk, slotMul);
    launch_block_scan(s, d_info, d_blockLen, d_origLen, nBlocks, fp, d_total);
    // The output must be zero before the OR-assembly; its size is only known on the device, so the
    // total is read back first (8 bytes) and only the used part is cleared.
    u64* h_total = reinterpret_cast<u64*>(c->pinned);
    HIPCHK(c, hipMemcpyAsync(h_total, d_total, sizeof(u64), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const u64 totalBits = *h_total;
    const size_t outBytes = (size_t)((totalBits + 7) >> 3);
    if (((outBytes + 8 + 3) & ~(size_t)3) > outCap) return fail(c, KNZ_ERR_WRITE_FILE, "output buffer too small: need %zu have %zu", (outBytes + 8 + 3) & ~(size_t)3, outCap);
    {
        ProfScope ps(c, "memset_out");
        HIPCHK(c, hipMemsetAsync(d_out, 0, (outBytes + 8 + 3) & ~(size_t)3, s));
    }
    if (prologueBits) {
        u8* d_pro;
        if (int r = ws_get(c, "prologue", 256, (void**)&d_pro)) return r;
        HIPCHK(c, hipMemcpyAsync(d_pro, prologue, (prologueBits + 7) / 8, hipMemcpyHostToDevice, s));
        launch_put_prologue(s, reinterpret_cast<u32*>(d_out), d_pro, prologueBits);
    }
    launch_assemble(s, d_desc, d_info, d_blockLen, d_origLen, d_skip, d_sums, d_tmp, nBlocks, maxChunks, entChunk, slotMul, hdrStride, fp,
                    reinterpret_cast<u32*>(d_out));
    HIPCHK(c, hipGetLastError());
    if (outBits) {
        HIPCHK(c, hipStreamSynchronize(s));
        *outBits = totalBits;
    }
    return 0;
}

int knz_hip_encode_blocks(knz_ctx* ctx, const knz_params* p, const uint8_t* d_in, size_t n, const uint8_t* prologue,
                          uint32_t prologue_bits, int64_t first_block_id, int finish, uint8_t* d_out, size_t out_cap,
                          uint64_t* out_bits)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    CTX_LOCK(c);
    return encode_impl(c, p, d_in, n, prologue, prologue_bits, 1, finish, first_block_id, d_out, out_cap, out_bits);
}

int knz_hip_encode_block_hosted(knz_ctx* ctx, const knz_params* p, const knz_host_stages* hs, const uint8_t* d_in, size_t n, const uint8_t* prologue,
                                uint32_t prologue_bits, int64_t first_block_id, int finish, uint8_t* d_out, size_t out_cap, uint64_t* out_bits)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    CTX_LOCK(c);
    if (!hs) return fail(c, KNZ_ERR_INVALID_PARAM, "no host stage record");
    return encode_impl(c, p, d_in, n, prologue, prologue_bits, 1, finish, first_block_id, d_out, out_cap, out_bits, hs);
}

// ------------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------------
struct WalkResultHost { u64 endBit; int64_t nBlocks; int32_t ended; int32_t error; };

static int decode_impl(Ctx* c, const knz_params* p, const uint8_t* d_in, uint64_t inBits, uint64_t startBit, int64_t maxBlocks,
                       int framing, u32 rawLen, uint8_t* d_out, size_t outCap, uint64_t* outBytes, uint64_t* endBit,
                       int64_t* blocksDone, int32_t* rawDecoded, uint64_t* usedBits, int nHosted = 0, uint32_t* skipOut = nullptr, uint64_t* sumOut = nullptr)
{
    ProfInstall pi_(c);
    HIPCHK(c, hipSetDevice(c->device));
    if ((reinterpret_cast<uintptr_t>(d_in) & 15) || (reinterpret_cast<uintptr_t>(d_out) & 15))
        return fail(c, KNZ_ERR_INVALID_PARAM, "device buffers must be 16-byte aligned");
    const u32 bs = (u32)p->block_size;
    if (framing && (bs < 1024 || bs > (1u << 30) || (bs & 15))) return fail(c, KNZ_ERR_INVALID_PARAM, "invalid block size %u", bs);
    int tok[8];
    const int nTok = count_transforms(p->transform_type, tok);
    if (nHosted < 0 || nHosted > nTok) return fail(c, KNZ_ERR_INVALID_PARAM, "host stage count %d does not fit the chain", nHosted);
    int tokAll[8];                                           // (the whole chain sizes the buffers, as in the encoder: a UTF stage adds 8 KiB of room)
    for (int i = 0; i < nTok; i++) tokAll[i] = tok[i];
    for (int i = 0; i < nTok; i++) {
        if (i < nHosted) {
            // the caller undoes these after the call: for the device they are stages that leave the data alone
            if (!host_stage_id(tok[i])) return fail(c, KNZ_ERR_INVALID_CODEC, "transform id %d is no host stage", tok[i]);
            tok[i] = KNZ_T_NONE;
        } else if (!transform_supported(tok[i])) return fail(c, KNZ_ERR_INVALID_CODEC, "transform id %d not implemented on device", tok[i]);
    }
    if (!entropy_supported(p->entropy_type)) return fail(c, KNZ_ERR_INVALID_CODEC, "entropy id %d not implemented on device", p->entropy_type);
    if (p->checksum_bits != 0 && p->checksum_bits != 32 && p->checksum_bits != 64) return fail(c, KNZ_ERR_INVALID_PARAM, "checksum must be 0, 32 or 64");
    // bitstream version of the blocks (0 = current). Below 6 the Huffman chunks, the BWT block header and the LZ blocks have their
    // old layouts (HuffmanDecoder.cpp:349-459, BWTBlockCodec.cpp:140-164, LZCodec.cpp:614-760)
    const int bsVersion = (p->bs_version == 0) ? 6 : p->bs_version;
    if (bsVersion < 0 || bsVersion > 6) return fail(c, KNZ_ERR_STREAM_VERSION, "cannot read bitstream version %d", bsVersion);
    hipStream_t s = c->stream;

    BitSrc src;
    src.words = reinterpret_cast<const u32*>(d_in);
    src.nBytes = (inBits + 7) >> 3;
    src.nWords = src.nBytes >> 2;
    src.limitBits = inBits;

    int64_t bound = framing ? (int64_t)(outCap / bs) + 2 : 1;
    if (maxBlocks > 0 && maxBlocks < bound) bound = maxBlocks;
    DecBlock* d_blocks; void* d_walk;
    if (int r = ws_get(c, "decBlocks", sizeof(DecBlock) * (size_t)bound, (void**)&d_blocks)) return r;
    if (int r = ws_get(c, "walk", 64, &d_walk)) return r;
    launch_walk_blocks(s, src, startBit, bound, framing, rawLen, p->checksum_bits, bs, d_blocks, d_walk);
    WalkResultHost* h_walk = reinterpret_cast<WalkResultHost*>(c->pinned);
    HIPCHK(c, hipMemcpyAsync(h_walk, d_walk, sizeof(WalkResultHost), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    const WalkResultHost walk = *h_walk;
    if (walk.error) return fail(c, walk.error, "invalid block framing");
    const int nBlocks = (int)walk.nBlocks;
    if (endBit) *endBit = walk.endBit;
    if (blocksDone) *blocksDone = nBlocks;
    if (nBlocks == 0) { if (outBytes) *outBytes = 0; return 0; }
    if (framing && (size_t)nBlocks * bs > outCap + bs) return fail(c, KNZ_ERR_WRITE_FILE, "output buffer too small");
    if (framing && (size_t)nBlocks * bs > (size_t)0x7FFFFFFF) return fail(c, KNZ_ERR_INVALID_PARAM, "batch of %d blocks exceeds the 2 GiB per-call limit (pass max_blocks)", nBlocks);

    // workspace stride: large enough for every valid preTransformLength of this chain
    const u32 unit = framing ? bs : rawLen;
    const int required = seq_required(tokAll, nTok, (int)unit);
    const u64 S = ((u64)required + 255) & ~255ull;
    const u32 maxPre = (u32)S;
    bool realStages = false;
    for (int i = 0; i < nTok; i++) if (tok[i] != KNZ_T_NONE) realStages = true;
    const int maxChunks = (int)((S + ENT_CHUNK - 1) / ENT_CHUNK);
    const u64 outStride = framing ? bs : 0;
    u32 realMask = 0;
    for (int i = 0; i < nTok; i++) if (tok[i] != KNZ_T_NONE) realMask |= 1u << (7 - i);

    // One range of the batch's blocks through entropy decoder and inverse chain, on stream `sp` with the workspaces of lane `lane` (-1: the
    // whole batch on the context's stream, the BWT inverse split into parts as before). The blocks are independent and the walk above has
    // left every block's place in the stream in d_blocks, so a range is a smaller decode of its own: its entries of d_blocks, its part of
    // the caller's output.
    auto issue = [&](int lane, int b0, int nb, hipStream_t sp) -> int {
        const std::string sfxS = lane <= 0 ? std::string() : std::string("#") + std::to_string(lane);
        const char* sfx = sfxS.c_str();
        auto wsName = [&](const char* base) { return std::string(base) + sfx; };
        DecBlock* blk = d_blocks + b0;
        uint8_t* out = d_out + (size_t)b0 * outStride;
        const u64 room = (u64)outCap - (u64)b0 * outStride;
        size_t scratch = 0;
        for (int i = 0; i < nTok; i++) if (tok[i] != KNZ_T_NONE) { const size_t q = stage_scratch_u32(tok[i], nb, (u32)S, false); if (q > scratch) scratch = q; }
        SeqWs w;
        if (int r = seq_alloc(c, nb, S, realStages, scratch, &w, sfx)) return r;
        const size_t nSlots = (size_t)nb * maxChunks;
        // entropy stage decodes into workspace A (or straight into the output when no transform applies)
        // with inverse stages the entropy decoder writes into the workspace (any valid preTransformLength fits);
        // the room in the caller's buffer is enforced where the last inverse stage gets its capacity
        launch_check_prelen(sp, blk, nb, realStages ? maxPre : unit, realStages ? ~0ull : room, outStride);
        launch_seq_inv_entropy_dst(sp, w.a, blk, nb, out, outStride, w.A, S, w.d_entDst, realMask, unit, room);
        if (p->entropy_type == KNZ_E_ANS0) {
            void* d_meta;
            if (int r = ws_get(c, wsName("ansDecChunks").c_str(), ans0_dec_chunk_bytes() * nSlots, &d_meta)) return r;
            launch_ans0_decode(sp, src, blk, nb, maxChunks, d_meta, w.d_entDst);
        } else if (p->entropy_type == KNZ_E_ANS1) {
            const int chunksPerBlock = (int)((S + ANS1_CHUNK - 1) / ANS1_CHUNK);
            const size_t nCh = (size_t)nb * chunksPerBlock;
            Ans1DecWs aw;
            if (int r = ws_get(c, wsName("ans1Meta").c_str(), ans1_meta_bytes(nCh), &aw.meta)) return r;
            if (int r = ws_get(c, wsName("ans1SlotTab").c_str(), ans1_slottab_bytes(nCh), (void**)&aw.slotTab)) return r;
            launch_ans1_decode(sp, src, blk, nb, chunksPerBlock, aw, w.d_entDst);
        } else if (p->entropy_type == KNZ_E_HUFFMAN) {
            void* d_meta;
            if (int r = ws_get(c, wsName("hufDecChunks").c_str(), huffman_dec_chunk_bytes() * nSlots, &d_meta)) return r;
            launch_huffman_decode(sp, src, blk, nb, maxChunks, d_meta, w.d_entDst, bsVersion);
        } else if (p->entropy_type == KNZ_E_FPAQ) {
            launch_fpaq_decode(sp, src, blk, nb, w.d_entDst);
        } else {
            launch_none_decode(sp, src, blk, nb, w.d_entDst);
        }
        // inverse transforms, last stage first (TransformSequence.hpp:197-224)
        if (realStages) {
            const u32 capFinal = framing ? bs : (u32)p->jobs;           // per-stage API passes its capacity in p->jobs
            const u32 blkLenModel = std::max(bs + 512u, bs + (bs >> 4));
            const u32 capMid = framing ? (u32)std::min<u64>(S, blkLenModel) : (u32)p->jobs;
            for (int i = nTok - 1; i >= 0; i--) {
                if (tok[i] == KNZ_T_NONE) continue;
                launch_seq_inv_prepare(sp, w.a, blk, nb, i, out, outStride, w.A, w.B, S, capMid, capFinal, realMask, framing ? room : ~0ull);
                XfStage st;
                st.src = w.a.src; st.dst = w.a.dst; st.len = w.a.alen; st.cap = w.a.cap; st.ok = w.a.ok; st.newLen = w.a.newLen;
                st.nBlocks = nb; st.maxLen = (u32)S; st.scratchU32 = w.scratch; st.entropyType = p->entropy_type; st.bsVersion = bsVersion;
                st.maxCap = std::max(capMid, capFinal);
                if (int r = run_inverse_stage(c, sp, tok[i], st, lane)) return r;
                launch_seq_inv_commit(sp, w.a, blk, nb, i, tok[i]);
            }
        }
        if (p->checksum_bits && framing && nHosted == 0) {
            u64* d_sums;
            if (int r = ws_get(c, wsName("sums").c_str(), sizeof(u64) * nb, (void**)&d_sums)) return r;
            launch_verify_checksums(sp, blk, nb, p->checksum_bits, out, outStride, w.d_viewPtr, w.a.alen, d_sums);
        }
        return 0;
    };

    // Up to three ranges side by side (knob dec_parts / KNZ_DEC_PARTS, default 3, at least KNZ_DEC_PART_MIN = 2 blocks per range -- 7 blocks: decode
    // 4.42 -> 4.04 ms; 4 blocks: no difference --; chains with inverse stages only; not
    // while per-kernel timing is on, not for chains with an LZ stage, whose scratch has one name): the entropy decoders are chains with a few
    // waves per CU and the row ranking of the BWT inverse is latency as well -- they run under the other ranges' bandwidth-bound kernels instead
    // of in front of them. Measured (26 blocks of 8 MiB, 1 / 2 / 3 ranges): decode 9.96 / 9.76 / 9.63 ms on the stand-in, 10.49 / 10.17 / 9.86 on the
    // real files; entropy-only chains (configs 1, 2) lose 2-5 % to the extra launches and stay one range.
    int lanes = 1;
    {
        bool lz = false;
        for (int i = 0; i < nTok; i++) if (tok[i] == KNZ_T_LZ || tok[i] == KNZ_T_LZX) lz = true;
        const int want = dec_parts_knob().load();
        static const int least = [] { const char* e = getenv("KNZ_DEC_PART_MIN"); const int x = e ? atoi(e) : 2; return x < 1 ? 1 : x; }();      // fewest blocks per range
        if (framing && realStages && !c->profiling && !lz && nHosted == 0 && want > 1) {
            lanes = want > 3 ? 3 : want;
            while (lanes > 1 && nBlocks < least * lanes) lanes--;
        }
    }
    if (lanes == 1) {
        if (int r = issue(-1, 0, nBlocks, s)) return r;
    } else {
        for (int k = 1; k < lanes; k++)
            if (c->stream2[k - 1] == nullptr) HIPCHK(c, hipStreamCreateWithFlags(&c->stream2[k - 1], hipStreamNonBlocking));
        if (c->evFork == nullptr) {
            HIPCHK(c, hipEventCreateWithFlags(&c->evFork, hipEventDisableTiming));
            for (int k = 0; k < 3; k++) HIPCHK(c, hipEventCreateWithFlags(&c->evJoin[k], hipEventDisableTiming));
        }
        HIPCHK(c, hipEventRecord(c->evFork, s));
        int first = 0, rc = 0;
        for (int k = 0; k < lanes; k++) {
            const int nb = nBlocks / lanes + (k < nBlocks % lanes ? 1 : 0);
            hipStream_t sp = k ? c->stream2[k - 1] : s;
            if (k) HIPCHK(c, hipStreamWaitEvent(sp, c->evFork, 0));
            if (rc == 0) rc = issue(k, first, nb, sp);
            first += nb;
            if (k) { HIPCHK(c, hipEventRecord(c->evJoin[k - 1], sp)); HIPCHK(c, hipStreamWaitEvent(s, c->evJoin[k - 1], 0)); }
        }
        if (rc) { hipStreamSynchronize(s); return rc; }
    }
    HIPCHK(c, hipGetLastError());
    // results
    std::vector<DecBlock> hb((size_t)nBlocks);
    HIPCHK(c, hipMemcpyAsync(hb.data(), d_blocks, sizeof(DecBlock) * (size_t)nBlocks, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    u64 total = 0;
    for (int b = 0; b < nBlocks; b++) {
        if (hb[b].error) {
            if (getenv("KNZ_DEBUG_ERR")) fprintf(stderr, "DBG block %d error %d used %llu\n", b, hb[b].error, (unsigned long long)hb[b].usedBits);
            if (rawDecoded) { *rawDecoded = -1; if (usedBits) *usedBits = hb[b].usedBits; return 0; }
            return fail(c, hb[b].error, "block %d: decoding failed (code %d)", b + 1, hb[b].error);
        }
        if (framing && b + 1 < nBlocks && hb[b].preLen != bs)
            return fail(c, KNZ_ERR_PROCESS_BLOCK, "block %d: short non-final block (%u bytes) not supported", b + 1, hb[b].preLen);
        total += hb[b].preLen;
    }
    if (outBytes) *outBytes = total;
    if (skipOut) *skipOut = hb[0].copyBlock ? 0xFFu : hb[0].skipFlags;
    if (sumOut) *sumOut = hb[0].checksum;
    if (rawDecoded) *rawDecoded = (int32_t)hb[0].preLen;
    if (usedBits) *usedBits = hb[0].usedBits;
    return 0;
}

int knz_hip_decode_blocks(knz_ctx* ctx, const knz_params* p, const uint8_t* d_in, uint64_t in_bits, uint64_t start_bit,
                          int64_t max_blocks, uint8_t* d_out, size_t out_cap, uint64_t* out_bytes, uint64_t* end_bit,
                          int64_t* blocks_done)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    CTX_LOCK(c);
    return decode_impl(c, p, d_in, in_bits, start_bit, max_blocks, 1, 0, d_out, out_cap, out_bytes, end_bit, blocks_done,
                       nullptr, nullptr);
}

int knz_hip_decode_block_hosted(knz_ctx* ctx, const knz_params* p, int32_t host_stages, const uint8_t* d_in, uint64_t in_bits, uint64_t start_bit,
                                uint8_t* d_out, size_t out_cap, uint64_t* out_bytes, uint64_t* end_bit, uint32_t* skip_flags, uint64_t* checksum, int32_t* done)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    CTX_LOCK(c);
    int64_t nb = 0;
    if (skip_flags) *skip_flags = 0xFF;
    if (checksum) *checksum = 0;
    const int r = decode_impl(c, p, d_in, in_bits, start_bit, 1, 1, 0, d_out, out_cap, out_bytes, end_bit, &nb, nullptr, nullptr, host_stages, skip_flags, checksum);
    if (done) *done = (int32_t)nb;
    return r;
}

// ------------------------------------------------------------------------------------------------
// per-stage (host buffers)
// ------------------------------------------------------------------------------------------------
int knz_hip_entropy_encode(knz_ctx* ctx, int entropy_type, const uint8_t* in, uint32_t n, uint8_t* out, size_t out_cap,
                           uint64_t* out_bits)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    CTX_LOCK(c);
    if (n == 0) { *out_bits = 0; return 0; }
    knz_params p; memset(&p, 0, sizeof(p));
    p.entropy_type = entropy_type; p.block_size = (int32_t)((n + 15) & ~15u); p.transform_type = 0;
    u8 *d_in, *d_out;
    const size_t cap = knz_hip_encode_bound(&p, n);
    if (int r = ws_get(c, "stageIn", (size_t)n + 64, (void**)&d_in)) return r;
    if (int r = ws_get(c, "stageOut", cap, (void**)&d_out)) return r;
    HIPCHK(c, hipMemcpyAsync(d_in, in, n, hipMemcpyHostToDevice, c->stream));
    u64 bits = 0;
    if (int r = encode_impl(c, &p, d_in, n, nullptr, 0, 0, 0, 0, d_out, cap, &bits)) return r;
    const size_t bytes = (size_t)((bits + 7) >> 3);
    if (bytes > out_cap) return fail(c, KNZ_ERR_WRITE_FILE, "output buffer too small");
    HIPCHK(c, hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *out_bits = bits;
    return 0;
}

int knz_hip_entropy_decode(knz_ctx* ctx, int entropy_type, const uint8_t* in, uint64_t in_bits, uint64_t start_bit,
                           uint8_t* out, uint32_t n, int32_t* decoded, uint64_t* used_bits)
{
    return knz_hip_entropy_decode_v(ctx, entropy_type, 0, in, in_bits, start_bit, out, n, decoded, used_bits);
}

int knz_hip_entropy_decode_v(knz_ctx* ctx, int entropy_type, int bs_version, const uint8_t* in, uint64_t in_bits, uint64_t start_bit,
                             uint8_t* out, uint32_t n, int32_t* decoded, uint64_t* used_bits)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    CTX_LOCK(c);
    if (bs_version < 0 || bs_version > 6) return fail(c, KNZ_ERR_STREAM_VERSION, "cannot read bitstream version %d", bs_version);   // (as knz_hip_transform_inverse_v)
    if (n == 0) { *decoded = 0; if (used_bits) *used_bits = 0; return 0; }
    knz_params p; memset(&p, 0, sizeof(p));
    p.entropy_type = entropy_type; p.block_size = (int32_t)((n + 15) & ~15u); p.bs_version = bs_version;
    const size_t inBytes = (size_t)((in_bits + 7) >> 3);
    u8 *d_in, *d_out;
    if (int r = ws_get(c, "stageIn", inBytes + 64, (void**)&d_in)) return r;
    if (int r = ws_get(c, "stageOut", (size_t)n + 64, (void**)&d_out)) return r;
    HIPCHK(c, hipMemcpyAsync(d_in, in, inBytes, hipMemcpyHostToDevice, c->stream));
    u64 ob = 0;
    if (int r = decode_impl(c, &p, d_in, in_bits, start_bit, 1, 0, n, d_out, n, &ob, nullptr, nullptr, decoded, used_bits)) return r;
    if (*decoded == (int32_t)n) {
        HIPCHK(c, hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

static int transform_host(Ctx* c, int t, int forward, const uint8_t* in, int32_t n, uint8_t* out, int32_t dstCap, int etype,
                          int32_t* outLen, int32_t* ok, int bsVersion = 6)
{
    CTX_LOCK(c);
    ProfInstall pi_(c);
    *outLen = 0; *ok = 0;
    if (!transform_supported(t) || t == KNZ_T_NONE) return fail(c, KNZ_ERR_INVALID_CODEC, "transform id %d not implemented on device", t);
    if (n < 0 || dstCap < 0) return fail(c, KNZ_ERR_INVALID_PARAM, "negative size");
    if (n == 0) { *ok = 1; return 0; }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const u32 maxLen = (u32)std::max(n, dstCap) + 2048;
    SeqWs w;
    if (int r = seq_alloc(c, 1, maxLen, false, stage_scratch_u32(t, 1, maxLen, forward != 0), &w)) return r;
    u8 *d_in, *d_out;
    if (int r = ws_get(c, "stageIn", (size_t)n + 64, (void**)&d_in)) return r;
    if (int r = ws_get(c, "stageOut", (size_t)maxLen + 64, (void**)&d_out)) return r;
    HIPCHK(c, hipMemcpyAsync(d_in, in, (size_t)n, hipMemcpyHostToDevice, s));
    struct { const u8* src; u8* dst; u32 len; u32 cap; } h;
    h.src = d_in; h.dst = d_out; h.len = (u32)n; h.cap = (u32)dstCap;
    HIPCHK(c, hipMemcpyAsync(w.a.src, &h.src, 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(w.a.dst, &h.dst, 8, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(w.a.alen, &h.len, 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(w.a.cap, &h.cap, 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemsetAsync(w.a.ok, 0, 1, s));
    HIPCHK(c, hipMemsetAsync(w.a.newLen, 0, 4, s));
    XfStage st;
    st.src = w.a.src; st.dst = w.a.dst; st.len = w.a.alen; st.cap = w.a.cap; st.ok = w.a.ok; st.newLen = w.a.newLen;
    st.nBlocks = 1; st.maxLen = (u32)n; st.scratchU32 = w.scratch; st.entropyType = etype; st.maxCap = (u32)dstCap;
    st.bsVersion = bsVersion;
    if (int r = forward ? run_forward_stage(c, s, t, st) : run_inverse_stage(c, s, t, st)) return r;
    HIPCHK(c, hipGetLastError());
    u8 hok = 0; u32 hlen = 0;
    HIPCHK(c, hipMemcpyAsync(&hok, w.a.ok, 1, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipMemcpyAsync(&hlen, w.a.newLen, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    *ok = hok;
    if (hok) {
        *outLen = (int32_t)hlen;
        HIPCHK(c, hipMemcpyAsync(out, d_out, hlen, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    return 0;
}

int knz_hip_transform_forward(knz_ctx* ctx, int transform_type, const uint8_t* in, int32_t n, uint8_t* out, int32_t dst_cap,
                              int entropy_type, int32_t* out_len, int32_t* ok)
{
    return transform_host(reinterpret_cast<Ctx*>(ctx), transform_type, 1, in, n, out, dst_cap, entropy_type, out_len, ok);
}

int knz_hip_transform_inverse(knz_ctx* ctx, int transform_type, const uint8_t* in, int32_t n, uint8_t* out, int32_t dst_cap,
                              int32_t* out_len, int32_t* ok)
{
    return transform_host(reinterpret_cast<Ctx*>(ctx), transform_type, 0, in, n, out, dst_cap, -1, out_len, ok);
}

int knz_hip_transform_inverse_v(knz_ctx* ctx, int transform_type, int bs_version, const uint8_t* in, int32_t n, uint8_t* out,
                                int32_t dst_cap, int32_t* out_len, int32_t* ok)
{
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    const int v = (bs_version == 0) ? 6 : bs_version;
    if (v < 0 || v > 6) return fail(c, KNZ_ERR_STREAM_VERSION, "cannot read bitstream version %d", v);
    return transform_host(c, transform_type, 0, in, n, out, dst_cap, -1, out_len, ok, v);
}

}  // extern "C"
