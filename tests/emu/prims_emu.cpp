// CPU-only check of kanzi-cpp_amd/csrc/prims.hpp (device-wide scans and the segmented LSD radix sort) on the fiber emulation of
// tools/hipemu, against std::stable_sort / plain loops. Built and run by tests/test_emu_kernels.py. Test infrastructure only.
#include "hip/hip_runtime.h"
#include "../../kanzi-cpp_amd/csrc/prims.hpp"

#include <algorithm>
#include <numeric>
#include <random>
#include <vector>
#include <stdio.h>

namespace knz { thread_local ProfHook* g_prof = nullptr; }
using namespace knz;

template <class KEY, bool HAS_VAL>
static int check_sort(std::mt19937_64& rng, const std::vector<u32>& segLens, int loBit, int hiBit, int keyKind)
{
    const int nSeg = (int)segLens.size();
    std::vector<u32> base(nSeg + 1, 0);
    for (int i = 0; i < nSeg; i++) base[i + 1] = base[i] + segLens[i];
    const size_t n = base[nSeg];
    std::vector<KEY> ka(n + 1), kb(n + 1);
    std::vector<u32> va(n + 1), vb(n + 1);
    for (size_t i = 0; i < n; i++) {
        const u64 r = rng();
        ka[i] = (KEY)(keyKind == 0 ? r : keyKind == 1 ? (r & 0x0303030303030303ull) : keyKind == 2 ? 0 : (r | 0xFFFF000000000000ull));
        va[i] = (u32)i;
    }
    std::vector<u8> ws(prims::rs_ws_bytes(n + 1, nSeg) + 256);
    u8* p = reinterpret_cast<u8*>((reinterpret_cast<uintptr_t>(ws.data()) + 255) & ~(uintptr_t)255);
    const prims::RsWs rs = prims::rs_carve(p, n + 1, nSeg, base.data(), nSeg);
    prims::rs_launch_layout(nullptr, rs);
    size_t maxSeg = 1;
    for (u32 l : segLens) maxSeg = std::max<size_t>(maxSeg, l);
    std::vector<KEY> k0 = ka;
    const int r = prims::rs_sort<KEY, HAS_VAL>(nullptr, rs, ka.data(), kb.data(), va.data(), vb.data(), maxSeg, loBit, hiBit);
    const KEY* ko = r ? kb.data() : ka.data();
    const u32* vo = r ? vb.data() : va.data();
    const int nb = hiBit - loBit;
    const KEY mask = (nb >= (int)(8 * sizeof(KEY))) ? (KEY)~(KEY)0 : (KEY)((((KEY)1) << nb) - 1);
    int bad = 0;
    for (int sgm = 0; sgm < nSeg; sgm++) {
        std::vector<u32> idx(segLens[sgm]);
        std::iota(idx.begin(), idx.end(), base[sgm]);
        std::stable_sort(idx.begin(), idx.end(), [&](u32 a, u32 b) { return ((k0[a] >> loBit) & mask) < ((k0[b] >> loBit) & mask); });
        for (u32 i = 0; i < segLens[sgm]; i++) {
            if (ko[base[sgm] + i] != k0[idx[i]] || (HAS_VAL && vo[base[sgm] + i] != idx[i])) { bad++; break; }
        }
    }
    if (bad) printf("FAIL sort: key bytes %d val %d segs %d n %zu bits [%d,%d) kind %d\n", (int)sizeof(KEY), (int)HAS_VAL, nSeg, n, loBit, hiBit, keyKind);
    return bad;
}

template <int OP>
static int check_scan(std::mt19937_64& rng, size_t n, bool dev)
{
    std::vector<u32> in(n + 16), out(n + 16, 0xDEADBEEF), tmp(prims::scan_tmp_bytes(n + 16) / 4 + 64);
    for (size_t i = 0; i < n; i++) in[i] = (u32)(rng() % 1000);
    u32 nDev = (u32)n, total = 0;
    // aligned views
    prims::launch_scan<OP>(nullptr, in.data(), out.data(), dev ? n + 7 : n, dev ? &nDev : nullptr, tmp.data(), OP == prims::SCAN_SUM_EXCL ? &total : nullptr);
    u32 run = OP == prims::SCAN_MIN_INCL ? 0xFFFFFFFFu : 0u;
    for (size_t i = 0; i < n; i++) {
        u32 want;
        if (OP == prims::SCAN_SUM_EXCL) { want = run; run += in[i]; }
        else if (OP == prims::SCAN_MAX_INCL) { run = std::max(run, in[i]); want = run; }
        else { run = std::min(run, in[i]); want = run; }
        if (out[i] != want) { printf("FAIL scan op %d n %zu at %zu: %u vs %u\n", OP, n, i, out[i], want); return 1; }
    }
    if (OP == prims::SCAN_SUM_EXCL && n && total != run) { printf("FAIL scan total n %zu\n", n); return 1; }
    return 0;
}

int main(int argc, char** argv)
{
    std::mt19937_64 rng(12345);
    int bad = 0;
    const bool quick = argc > 1 && argv[1][0] == 'q';        // a short form for the runs under other switches (the sorts only)
    const std::vector<std::vector<u32>> layouts = quick ? std::vector<std::vector<u32>>{ {100, 0, 5000, 4096, 1}, {20000} }
        : std::vector<std::vector<u32>>{ {1}, {4096}, {4097}, {100, 0, 5000, 4096, 1}, {20000}, {70000}, {33000, 33001} };      // (more than 16 tiles: two levels of the column scan)
    for (const auto& lay : layouts) {
        for (int kind = 0; kind < (quick ? 2 : 4); kind++) {
            bad += check_sort<u64, false>(rng, lay, 0, 64, kind);
            bad += check_sort<u64, true>(rng, lay, 3, 45, kind);
            bad += check_sort<u32, true>(rng, lay, 0, 21, kind);
            bad += check_sort<u32, false>(rng, lay, 8, 32, kind);
        }
    }
    if (!quick) for (size_t n : { (size_t)1, (size_t)15, (size_t)16, (size_t)4095, (size_t)4096, (size_t)4097, (size_t)100000, (size_t)300000 })
        for (int dev = 0; dev < 2; dev++) {
            bad += check_scan<prims::SCAN_SUM_EXCL>(rng, n, dev != 0);
            bad += check_scan<prims::SCAN_MAX_INCL>(rng, n, dev != 0);
            bad += check_scan<prims::SCAN_MIN_INCL>(rng, n, dev != 0);
        }
    printf(bad ? "FAILED %d\n" : "OK\n", bad);
    return bad ? 1 : 0;
}
