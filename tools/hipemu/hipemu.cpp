// TEST INFRASTRUCTURE ONLY -- fiber scheduler behind tools/hipemu/hip/hip_runtime.h (see there).
#include "hip/hip_runtime.h"
#include <sys/mman.h>
#include <algorithm>

extern "C" void hipemu_switch(void** fromSp, void* toSp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {

static const size_t STACK = 256 << 10;
static Block g_blk;
static std::vector<char*> g_stacks;

// per-wave rendezvous state
struct WaveX {
    unsigned long long val[64]; unsigned long long arrived; int gen; unsigned long long snapshot[64]; unsigned long long snapMask; bool spin;
    // a second meeting point for a SUBSET of the lanes (a wave intrinsic inside divergent control flow: on the device the exec mask
    // restricts it to the lanes that are there; here the kernel names them, KNZ_BALLOT_OF)
    unsigned long long subVal[64]; unsigned long long subArrived, subMask; int subGen; unsigned long long subSnapshot[64];
};
static std::vector<WaveX> g_wave;
static int g_barCount, g_barGen;

Block& blk() { return g_blk; }

void yield_to_scheduler() { Fiber* f = g_blk.cur; hipemu_switch(&f->sp, g_blk.schedSp); }

static void fiber_entry()
{
    g_blk.body();
    g_blk.cur->state = 3;
    g_blk.nLive--;
    yield_to_scheduler();
    abort();
}

static unsigned long long live_mask(int wave)
{
    unsigned long long m = 0;
    const int n = (int)g_blk.fibers.size();
    for (int l = 0; l < 64; l++) { const int t = wave * 64 + l; if (t < n && g_blk.fibers[t].state != 3) m |= 1ull << l; }
    return m;
}

void block_barrier()
{
    Fiber* f = g_blk.cur;
    const int gen = g_barGen;
    g_barCount++;
    f->state = 1;
    while (g_barGen == gen) yield_to_scheduler();
    f->state = 0;
}

void wave_exchange(unsigned long long v, unsigned long long out[64], unsigned long long* activeMask)
{
    Fiber* f = g_blk.cur;
    const int t = (int)(f - &g_blk.fibers[0]);
    const int wave = t >> 6, lane = t & 63;
    WaveX& w = g_wave[wave];
    const int gen = w.gen;
    w.val[lane] = v;
    w.arrived |= 1ull << lane;
    f->state = 2;
    while (w.gen == gen) yield_to_scheduler();
    f->state = 0;
    memcpy(out, w.snapshot, sizeof(w.snapshot));
    *activeMask = w.snapMask;
}

void wave_exchange_subset(unsigned long long v, unsigned long long out[64], unsigned long long lanes)
{
    Fiber* f = g_blk.cur;
    const int t = (int)(f - &g_blk.fibers[0]);
    const int wave = t >> 6, lane = t & 63;
    WaveX& w = g_wave[wave];
    if (w.subArrived != 0 && w.subMask != lanes) { fprintf(stderr, "hipemu: two subset rendezvous of one wave at a time\n"); abort(); }
    const int gen = w.subGen;
    w.subMask = lanes;
    w.subVal[lane] = v;
    w.subArrived |= 1ull << lane;
    f->state = 4;
    while (w.subGen == gen) yield_to_scheduler();
    f->state = 0;
    memcpy(out, w.subSnapshot, sizeof(w.subSnapshot));
}

// a wave that polls memory another wave of the block writes: a rendezvous after which the scheduler moves on to the other waves
void wave_spin()
{
    const int t = (int)(g_blk.cur - &g_blk.fibers[0]);
    g_wave[t >> 6].spin = true;
    unsigned long long v[64], act;
    wave_exchange(0, v, &act);
}

static void run_block()
{
    const int n = (int)g_blk.fibers.size();
    const int nWaves = (n + 63) / 64;
    g_wave.assign(nWaves, WaveX());
    g_barCount = 0; g_barGen = 0;
    g_blk.nLive = n;
    for (int t = 0; t < n; t++) {
        Fiber& f = g_blk.fibers[t];
        f.state = 0;
        // initial frame: six callee-saved register slots, then the entry address; 16-byte alignment as after a call
        char* top = f.stack + STACK;
        void** sp = reinterpret_cast<void**>((reinterpret_cast<uintptr_t>(top) & ~(uintptr_t)15) - 8);
        *--sp = reinterpret_cast<void*>(&fiber_entry);
        for (int i = 0; i < 6; i++) *--sp = nullptr;
        f.sp = sp;
    }
    while (g_blk.nLive > 0) {
        bool progress = false;
        // HIPEMU_WAVE_ORDER=1: the waves of a workgroup are visited last to first (kernels whose waves talk through LDS must not
        // depend on who runs first)
        static const bool revWaves = getenv("HIPEMU_WAVE_ORDER") && atoi(getenv("HIPEMU_WAVE_ORDER")) == 1;
        for (int wi = 0; wi < nWaves; wi++) {
            const int wv = revWaves ? nWaves - 1 - wi : wi;
            bool again = true;
            while (again) {
                again = false;
                for (int l = 0; l < 64; l++) {
                    const int t = wv * 64 + l;
                    if (t >= n) break;
                    Fiber& f = g_blk.fibers[t];
                    if (f.state == 0) { g_blk.cur = &f; hipemu_switch(&g_blk.schedSp, f.sp); progress = true; }
                }
                // wave rendezvous complete? (every live lane of the wave has arrived)
                WaveX& w = g_wave[wv];
                const unsigned long long live = live_mask(wv);
                if (w.subArrived != 0 && w.subArrived == (w.subMask & live)) {
                    memcpy(w.subSnapshot, w.subVal, sizeof(w.subVal));
                    const unsigned long long who = w.subArrived;
                    w.subArrived = 0;
                    w.subGen++;
                    for (int l = 0; l < 64; l++) if ((who >> l) & 1) g_blk.fibers[wv * 64 + l].state = 0;
                    again = true; progress = true;
                }
                if (w.arrived != 0 && (w.arrived & live) == live) {
                    bool allWaiting = true;
                    for (int l = 0; l < 64; l++) if ((live >> l) & 1) if (g_blk.fibers[wv * 64 + l].state != 2) allWaiting = false;
                    if (allWaiting) {
                        memcpy(w.snapshot, w.val, sizeof(w.val));
                        w.snapMask = w.arrived;
                        w.arrived = 0;
                        w.gen++;
                        for (int l = 0; l < 64; l++) if ((live >> l) & 1) g_blk.fibers[wv * 64 + l].state = 0;
                        again = !w.spin; progress = true;
                        w.spin = false;
                    }
                }
            }
        }
        // block barrier complete?
        if (g_barCount > 0 && g_barCount == g_blk.nLive) {
            g_barCount = 0; g_barGen++;
            for (int t = 0; t < n; t++) if (g_blk.fibers[t].state == 1) g_blk.fibers[t].state = 0;
            progress = true;
        }
        if (!progress) {
            fprintf(stderr, "hipemu: deadlock in block (%u,%u): %d live threads, %d at the barrier\n", g_blk.bIdx.x, g_blk.bIdx.y, g_blk.nLive, g_barCount);
            abort();
        }
    }
}

void launch(const std::function<void()>& body, dim3 grid, dim3 block)
{
    const size_t n = (size_t)block.x * block.y * block.z;
    while (g_stacks.size() < n) {
        void* p = mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) abort();
        g_stacks.push_back(static_cast<char*>(p));
    }
    g_blk.fibers.assign(n, Fiber());
    for (size_t t = 0; t < n; t++) {
        g_blk.fibers[t].stack = g_stacks[t];
        g_blk.fibers[t].tid = dim3((unsigned)(t % block.x), (unsigned)((t / block.x) % block.y), (unsigned)(t / ((size_t)block.x * block.y)));
    }
    g_blk.body = body;
    g_blk.bDim = block; g_blk.gDim = grid;
    // HIPEMU_ORDER: 0 = workgroups in index order, 1 = reverse, 2 = a fixed pseudo-random permutation. HIP promises no
    // dispatch order; running a test under several orders exposes kernels that depend on one.
    static const int order = getenv("HIPEMU_ORDER") ? atoi(getenv("HIPEMU_ORDER")) : 0;
    const size_t nb = (size_t)grid.x * grid.y * grid.z;
    std::vector<size_t> seq(nb);
    for (size_t i = 0; i < nb; i++) seq[i] = (order == 1) ? nb - 1 - i : i;
    if (order == 2) { unsigned long long st = 0x9E3779B97F4A7C15ull; for (size_t i = nb; i > 1; i--) { st = st * 6364136223846793005ull + 1442695040888963407ull; std::swap(seq[i - 1], seq[(size_t)((st >> 33) % i)]); } }
    for (size_t i = 0; i < nb; i++) {
        const size_t q = seq[i];
        g_blk.bIdx = dim3((unsigned)(q % grid.x), (unsigned)((q / grid.x) % grid.y), (unsigned)(q / ((size_t)grid.x * grid.y)));
        run_block();
    }
}

}  // namespace hipemu
