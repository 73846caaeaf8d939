// Forward Burrows-Wheeler transform (block codec) on gfx950: suffix sorting of all blocks of a batch at once.
//
// Reference being replaced (results are bit-identical; the suffix array of a block is unique, so any correct
// construction matches divsufsort):
//   transform/BWTBlockCodec.cpp:32-87 (header), transform/BWT.cpp:92-134, transform/DivSufSort.cpp:171-295
//   (computeBWT / constructBWT): "proper prefix sorts first", out[0] = in[n-1], then in[SA[r]-1] for ranks with
//   SA[r] != 0, 8 primary indexes rank(suffix k*ceil(n/8)) + 1 (BWT.hpp:40-61).
//
// GPU formulation: prefix doubling with group refinement (Larsson-Sadakane order refinement, made data parallel).
//   State in HBM: SA[slot] (global position ids), ISA[position] = a label of the position's group (a slot inside the group's range), and
//   one bit per slot, gbits, set where a group starts. A position is resolved when its group has one member.
//   Round 0 sorts every block's suffixes on their first 4 symbols with the segmented LSD radix sort of prims.hpp; its first pass reads
//   the text. Then:
//     * the small groups once on the next eight text bytes (k_bwt_f_sort_small_text),
//     * run groups (4 equal symbols) in one round by run length: the runs are sorted, the members follow their runs (k_bwt_f_run_*),
//   and the doubling rounds, h = 4, 8, 16, ..., refine what is left on the key ISA[p + h] (0 past the block end: "shorter sorts first"):
//     * small groups (2..256 members) need no list at all: a kernel sweeps the bit map in windows of 2048 slots, finds the groups that
//       start in its window and ranks every member by counting inside LDS (less / equal / equal-before),
//     * medium groups (257..8192) are (start, length) descriptors (found through staging slots, no atomics); one workgroup sorts a
//       group in LDS: majority-key split or a stable LSD radix sort (8-bit digits, ranking by ballot matching inside a wave),
//     * a group whose majority key is its OWN label (a periodic stretch whose period divides h) is finished in one round by pointer
//       jumping along its chains (k_bwt_f_super),
//     * large groups go through one global radix sort of (descriptor index, key) pairs.
//   A round sorts on the labels as they stood when it began (a refined head read beside an unrefined one would order two suffixes that
//   are still equal): the keys of medium and large groups are gathered by kernels of their own before anything moves; the small groups
//   read theirs in the kernel that sorts them (k_bwt_f_small_fused, round 6), which the VERSIONED labels make safe (lab_old / lab_set below).
#include "common.hpp"
#include "stages.hpp"
#include "bwt_common.hpp"

#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <cmath>
#include "prims.hpp"
#include <chrono>
#include <vector>

namespace knz {

constexpr u32 SM_TS = 1792;        // slots owned by one window
constexpr u32 SM_WIN = 2048;       // slots a window looks at (owned + halo)
constexpr u32 SM_G = 256;          // largest "small" group
// largest "medium" group (one workgroup, LDS resident). Round 6 measured 16,384 with a 1024 x 16 instantiation of k_bwt_f_sort_medium (150 KiB of
// LDS, one workgroup per CU): period-768 blocks 19.0 -> 9.8 ms, but real files 51.7 -> 52.6, the stand-in 26.6 -> 26.9 -- for groups of that
// size the global sort of (descriptor, key) pairs is no slower than one workgroup per CU. Stays 8,192.
constexpr u32 MED_CAP = 8192;
constexpr u32 SUPER_CAP = 8192;    // largest group the chain round and the periodic-stretch probe take
constexpr u32 NO_BIT = 0x7FFFFFFFu;

struct FwdView {
    const u32* base;     // [nBlocks + 1] slot / position base of every block (dense), base[nBlocks] = total
    int nBlocks;
    u32 total;
    u32* SA;             // [total]
    u32* ISA;            // [total] labels as 32-bit words (blocks above 256 MiB), or
    u64* ISA2;           // [total] VERSIONED labels (null: the plain words above): see lab_old / lab_set below
    u32 round;           // number of the doubling round under way (0: in front of the first one)
    u32* K;              // [total] keys of the round, by slot
    u32* gbits;          // group-start bit per slot (bit k of word w = slot 32 w + k); set for every slot >= total
    u32* gnew;           // group starts found in the current round; merged into gbits when the round is over (a window must
                         // not see the subgroups a neighbouring window has just made: their keys belong to the old order)
    u32* counters;       // [0] != 0: small groups are left, [1] medium descriptors, [2] large descriptors, [3] members of large groups
    const u32* ovr;      // [total] by slot: run length R of the members of a group the run round left unresolved (valid where rtbits is set)
    const u32* rtbits;   // bit per slot: the slot belongs to such a group. Its members are c^R followed by different things, so the key that
                         // tells them apart is the label R positions on -- the doubling rounds look there (at max(R, h)) instead of h positions
                         // on, where for h < R they would find the same run again and learn nothing (null: no run round)
    uint2* medStage;     // medium groups found in the current round, one slot per 256 slots of SA (a medium group has more than 256
                         // members, so two of them never start in the same 256): written without atomics, compacted -- in slot order --
                         // into the next round's descriptor list by k_bwt_f_med_compact
};

// ---- labels --------------------------------------------------------------------------------------------------------------------------
// A round must sort on the labels as they stood when it began: a refined label read beside one that is still to be refined would order two
// suffixes that are equal so far (the old label of a group says nothing about where its members go). Rounds 1-5 therefore gathered all keys
// of a round (k_bwt_f_gather_*) before any kernel of it changed a label. With VERSIONED labels a kernel may read keys and write labels in the
// same launch: an entry of ISA2 holds, in one 64-bit word that is read and written whole,
//     bits 0-27 the label, bits 28-55 the label it replaced, bits 56-63 the round in which it was written,
// labels as slots relative to their block's first slot (blocks up to 2^28 bytes; larger ones use the plain words and separate key
// kernels). A reader of round r takes the replaced label when the entry carries r's number -- whatever kernel of the round wrote it, before
// or beside the reader -- and the label otherwise. A position changes its label at most once per round (it is a member of one group).
// Writers outside the rounds (round 0, text round, run round, probe: nobody reads beside them) write label = replaced label, number 0.
constexpr u32 LAB_BITS = 28;
constexpr u64 LAB_MASK = (1ull << LAB_BITS) - 1ull;

// label of position q as of the end of the previous round (global slot); bb = first slot of q's block
__device__ __forceinline__ u32 lab_old(const FwdView& v, u32 q, u32 bb)
{
    if (v.ISA2 == nullptr) return v.ISA[q];
    const u64 e = v.ISA2[q];
    const u32 rel = ((u32)(e >> 56) == v.round && v.round != 0) ? (u32)((e >> LAB_BITS) & LAB_MASK) : (u32)(e & LAB_MASK);
    return bb + rel;
}
// the latest label (for kernels that run where no label changes)
__device__ __forceinline__ u32 lab_cur(const FwdView& v, u32 q, u32 bb)
{
    if (v.ISA2 == nullptr) return v.ISA[q];
    return bb + (u32)(v.ISA2[q] & LAB_MASK);
}
// position p of the block that starts at slot bb goes from label `was` to label `lab` (global slots)
__device__ __forceinline__ void lab_set(const FwdView& v, u32 p, u32 bb, u32 lab, u32 was)
{
    if (v.ISA2 == nullptr) { v.ISA[p] = lab; return; }
    const u64 nw = (u64)(lab - bb), ow = (v.round == 0) ? nw : (u64)(was - bb);
    v.ISA2[p] = nw | (ow << LAB_BITS) | ((u64)v.round << 56);
}

__device__ __forceinline__ u32 gather_key(const FwdView& v, u32 gp, u32 h, u32 blkBase, u32 blkEnd)
{
    const u32 q = gp + h;
    return (q < blkEnd) ? lab_old(v, q, blkBase) - blkBase + 1u : 0u;
}

__device__ __forceinline__ void classify_child(const FwdView& v, uint2* __restrict__ medNext, uint2* __restrict__ largeNext, u32 start, u32 size, u32& surv)
{
    if (size > MED_CAP) {
        const u32 at = atomicAdd(&v.counters[2], 1u);
        largeNext[at] = make_uint2(start, size);
        atomicAdd(&v.counters[3], size);
    } else if (size > SM_G) {
        v.medStage[start >> 8] = make_uint2(start, size);
    } else if (size > 1) {
        surv = 1;
    }
}

// The same through the workgroup: the large / run groups a workgroup finds are counted in LDS first, ONE thread then reserves their
// places in the lists (one global atomic per list and workgroup instead of one per group), and the descriptors are written at
// base + local index. Medium groups -- hundreds of thousands after round 0 of a text, where atomics on ONE counter were two thirds
// of k_bwt_f_r0_place -- use no counter at all: they go to v.medStage (see FwdView).
struct ClassAgg { u32 cnt[3]; u32 elems[3]; u32 base[3]; };        // 0 medium, 1 large, 2 run groups

__device__ __forceinline__ void agg_init(ClassAgg& A) { if (threadIdx.x < 3) { A.cnt[threadIdx.x] = 0; A.elems[threadIdx.x] = 0; A.base[threadIdx.x] = 0; } }

// kind of the group (-1: small or resolved) and its index among the workgroup's groups of that kind
__device__ __forceinline__ int agg_note(ClassAgg& A, u32 size, bool run, u32& surv, u32& local)
{
    int kind;
    if (run) kind = 2;
    else if (size > MED_CAP) kind = 1;
    else if (size > SM_G) kind = 0;
    else { if (size > 1) surv = 1; return -1; }
    if (kind != 0) { local = atomicAdd(&A.cnt[kind], 1u); atomicAdd(&A.elems[kind], size); }
    return kind;
}

// one thread, between two barriers
__device__ __forceinline__ void agg_reserve(ClassAgg& A, const FwdView& v)
{
    if (A.cnt[1]) { A.base[1] = atomicAdd(&v.counters[2], A.cnt[1]); atomicAdd(&v.counters[3], A.elems[1]); }
    if (A.cnt[2]) { A.base[2] = atomicAdd(&v.counters[4], A.cnt[2]); atomicAdd(&v.counters[5], A.elems[2]); }
}

__device__ __forceinline__ void agg_write(const ClassAgg& A, const FwdView& v, int kind, u32 local, u32 start, u32 size, uint2* __restrict__ largeNext,
                                          uint2* __restrict__ runList)
{
    if (kind == 0) { v.medStage[start >> 8] = make_uint2(start, size); return; }        // (medium groups: no counter at all)
    uint2* dst = kind == 1 ? largeNext : runList;
    dst[A.base[kind] + local] = make_uint2(start, size);
}

// ------------------------------------------------------------------------------------------------
// round 0
// ------------------------------------------------------------------------------------------------
// base[b] = sum of active lengths of the blocks before b ; total in base[nBlocks]
__global__ void k_bwt_bases(BwtView v, u32* __restrict__ base, u8* __restrict__ ok, u32* __restrict__ maxLen)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u32 sum = 0, mx = 0;
    for (int b = 0; b < v.nBlocks; b++) {
        base[b] = sum;
        u32 ps;
        const bool a = bwt_fwd_applies(v.len[b], v.cap[b], &ps);
        ok[b] = a ? 1 : 0;
        if (a) { sum += v.len[b]; mx = v.len[b] > mx ? v.len[b] : mx; }
    }
    base[v.nBlocks] = sum;
    *maxLen = mx;                       // longest block the transform applies to
}

// Round-0 key of a suffix: [P bytes of the suffix, zero past the block end | position inside the block in the low pbits bits].
// There is no block field (every block is a segment of the sort with its own digit bins) and no length field: the P - 1 suffixes
// that end inside their own key ("short" suffixes) are handed to the first pass IN FRONT of the others, shortest first, and the
// sort is stable -- so among equal padded keys they come first, shortest first, which is their place ("a proper prefix sorts
// first"); the flag kernel makes each of them a group of its own.
struct TextSrc {
    static constexpr bool STAGED = true;
    const u8* const* src;
    int P, pbits, shift;
    __device__ __forceinline__ u32 pos_of(u32 n, u32 i) const
    {
        const u32 nShort = (u32)(P - 1) < n ? (u32)(P - 1) : n;
        return i < nShort ? n - 1 - i : i - nShort;
    }
    __device__ __forceinline__ u64 load(int sgm, u32, u32 n, u32 i) const
    {
        const u8* t = src[sgm];
        const u32 pos = pos_of(n, i);
        const u64 v = text8(t, pos, n);
        u64 kb = __builtin_bswap64(v) >> (64 - 8 * P);
        const u32 left = n - pos;
        if (left < (u32)P) kb &= ~0ull << (8 * ((u32)P - left));
        return (kb << pbits) | (u64)pos;
    }
    // The tile's stretch of the text in LDS (round 6; the pass took 1.57 ms against 0.97 for a pass over keys: a key was three dword loads
    // that overlap the neighbours'). Elements tile0 .. tile0 + RS_TILE - 1 are consecutive positions from pos0 on (the short suffixes in
    // front of the first tile keep the loads above): the dwords from the one that holds pos0 to the one behind pos0 + RS_TILE + 7, zero from
    // the block's end on, which is what text8 delivers there. A block whose text is not dword aligned is not staged.
    __device__ __forceinline__ bool stage(int sgm, u32 n, u32 tile0, u32* lds, int tid) const
    {
        const u8* t = src[sgm];
        if (reinterpret_cast<uintptr_t>(t) & 3) return false;
        const u32 nShort = (u32)(P - 1) < n ? (u32)(P - 1) : n;
        const u32 pos0 = (tile0 > nShort ? tile0 : nShort) - nShort;
        const u32 a0 = pos0 & ~3u;
        constexpr u32 NW = (prims::RS_TILE + 8u + 3u) / 4u + 2u;
        for (u32 w = (u32)tid; w < NW; w += (u32)prims::RS_THREADS) {
            const u32 q = a0 + 4u * w;
            u32 x = 0;
            if (q + 4u <= n) x = ldg<u32>(t + q);
            else for (u32 j = 0; j < 4u; j++) if (q + j < n) x |= (u32)ldg<u8>(t + q + j) << (8u * j);
            lds[w] = x;
        }
        return true;
    }
    __device__ __forceinline__ u64 load_staged(int sgm, u32 b0, u32 n, u32 i, const u32* lds, u32 tile0) const
    {
        const u32 nShort = (u32)(P - 1) < n ? (u32)(P - 1) : n;
        if (i < nShort) return load(sgm, b0, n, i);
        const u32 pos0 = (tile0 > nShort ? tile0 : nShort) - nShort;
        const u32 pos = i - nShort;
        const u32 o = pos - (pos0 & ~3u);
        const u32 w = o >> 2, sh = (o & 3u) * 8u;
        const u64 lo = (u64)lds[w] | ((u64)lds[w + 1] << 32);
        const u64 v = sh ? ((lo >> sh) | ((u64)lds[w + 2] << (64u - sh))) : lo;
        u64 kb = __builtin_bswap64(v) >> (64 - 8 * P);
        const u32 left = n - pos;
        if (left < (u32)P) kb &= ~0ull << (8 * ((u32)P - left));
        return (kb << pbits) | (u64)pos;
    }
    __device__ __forceinline__ u32 digit(u64 k) const { return (u32)(k >> shift) & 255u; }
    // (the first pass sorts on the last key byte)
    __device__ __forceinline__ u32 digit_at(int sgm, u32, u32 n, u32 i) const
    {
        const u32 q = pos_of(n, i) + (u32)(P - 1);
        return q < n ? (u32)ldg<u8>(src[sgm] + q) : 0u;
    }
};

// Byte histogram of every block (slices of a block by blockIdx.x, summed with atomics into byteHist[block][256], zeroed by the caller).
// Two things are read off it: the key length of round 0 (k_bwt_f_choose_nsym) and the digit counts of all round-0 passes
// (k_bwt_f_r0_counts; prims::k_rs_hist_all would build every key to count its bytes).
__global__ __launch_bounds__(256) void k_bwt_f_bytehist(BwtView bv, const u32* __restrict__ base, u32* __restrict__ byteHist)
{
    // four counter sets per wave (lane & 3 picks one; 257 words apart so that equal symbols of different sets fall into different banks):
    // lanes that show the same symbol in one step serialise on its counter, and text shows its few frequent symbols in most steps
    __shared__ u32 cnt[4][4][257];
    const int sgm = blockIdx.y;
    const u32 n = base[sgm + 1] - base[sgm];
    if (n == 0) return;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int q = tid; q < 4 * 4 * 257; q += 256) (&cnt[0][0][0])[q] = 0;
    __syncthreads();
    const u8* t = bv.src[sgm];
    const u32 per = ((n + gridDim.x - 1) / gridDim.x + 15u) & ~15u;
    const u32 lo = blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
    // 16 bytes per lane and step (the block's text is 16-byte aligned or the slow path below takes it)
    const bool al = (reinterpret_cast<uintptr_t>(t) & 15) == 0;
    u32* mine = cnt[wave][lane & 3];
    for (u32 i0 = lo; i0 < hi; i0 += 16u * 256u) {                 // (uniform trip count: the row ballots want whole waves)
        const u32 i = i0 + 16u * (u32)tid;
        u32 wds[4] = { 0, 0, 0, 0 };
        const u32 m = (i >= hi) ? 0u : ((hi - i < 16u) ? hi - i : 16u);
        if (al && m == 16u) { wds[0] = ldg<u32>(t + i); wds[1] = ldg<u32>(t + i + 4); wds[2] = ldg<u32>(t + i + 8); wds[3] = ldg<u32>(t + i + 12); }
        else for (u32 j = 0; j < m; j++) wds[j >> 2] |= (u32)ldg<u8>(t + i + j) << (8 * (j & 3));
#pragma unroll
        for (u32 j = 0; j < 16; j++) {
            const bool valid = j < m;
            const u32 dg = (wds[j >> 2] >> (8 * (j & 3))) & 255u;
            const u32 d0 = (u32)__builtin_amdgcn_readfirstlane((int)dg);
            const unsigned long long va = __ballot(valid);
            if (__ballot(valid && dg != d0) == 0) { if (lane == 0 && va) mine[d0] += (u32)__popcll(va); }
            else if (valid) atomicAdd(&mine[dg], 1u);
            KNZ_WAVE_ORDER();
        }
    }
    __syncthreads();
    u32 c = 0;
    for (int w = 0; w < 4; w++) for (int q = 0; q < 4; q++) c += cnt[w][q][tid];
    if (c) atomicAdd(&byteHist[(size_t)sgm * 256 + tid], c);
}

// Key length of round 0 from the order-0 entropy H of the blocks: four symbols tell suffixes apart when 4 H bits reach the
// log2(block length) bits a position has (the mixed stand-in: H = 6.8); text (H = 4.1), DNA (2.0) leave most suffixes in groups then,
// and a fifth symbol in the keys is cheaper than the doubling rounds those groups cost (measured on 212 MB of text at 8 MiB blocks:
// suffix sort 33.3 -> 29.6 ms; on the stand-in 27.1 -> 27.3). The host caps the answer by what fits the key beside the position.
// Round 6: H is the MEAN OF THE BLOCKS' entropies, weighted by their lengths, not the entropy of the batch's summed histogram -- a batch of
// different files (shared objects at 6-7 bits beside headers and scripts at 4.3-4.9) has a flat sum (6.1 for the real-file corpus) although
// most of its blocks are text-like: five symbols are worth 1.0 ms of 51.7 there. Batches of more than 64 blocks keep the summed histogram.
__global__ __launch_bounds__(256) void k_bwt_f_choose_nsym(const u32* __restrict__ byteHist, int nBlocks, const u32* __restrict__ longest, u32* __restrict__ out)
{
    __shared__ float part[256];
    __shared__ unsigned long long tot[256];
    const int tid = (int)threadIdx.x;
    float Hsum = 0.0f;                                   // (thread 0: sum of length x entropy over the blocks)
    unsigned long long lenSum = 0;
    const int nParts = nBlocks <= 64 ? nBlocks : 1;
    for (int q = 0; q < nParts; q++) {
        unsigned long long c = 0;
        if (nParts == 1) { for (int b = 0; b < nBlocks; b++) c += byteHist[(size_t)b * 256 + tid]; }
        else c = byteHist[(size_t)q * 256 + tid];
        tot[tid] = c;
        __syncthreads();
        unsigned long long all = 0;
        for (int i = 0; i < 256; i++) all += tot[i];
        part[tid] = (c && all) ? (float)((double)c / (double)all) * log2f((float)((double)all / (double)c)) : 0.0f;
        __syncthreads();
        if (tid == 0) {
            float H = 0.0f;
            for (int i = 0; i < 256; i++) H += part[i];
            Hsum += H * (float)all;
            lenSum += all;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float H = lenSum ? Hsum / (float)lenSum : 8.0f;
        const u32 n = longest[0] > 1 ? longest[0] : 2;
        out[0] = (4.0f * H < log2f((float)n)) ? 5u : 4u;
    }
}

// digit counts of the P round-0 passes of a block: the digit of pass k of the suffix at pos is the text byte at pos + o, o = P - 1 - k,
// or 0 behind the block's end, so its histogram is the block's byte histogram minus the block's first o bytes, plus o zeros
__global__ __launch_bounds__(256) void k_bwt_f_r0_counts(BwtView bv, const u32* __restrict__ base, const u32* __restrict__ byteHist, int P, prims::RsLayout L)
{
    const int sgm = blockIdx.x, tid = (int)threadIdx.x;
    const u32 n = base[sgm + 1] - base[sgm];
    const u32 c = n ? byteHist[(size_t)sgm * 256 + tid] : 0u;
    const u8* t = bv.src[sgm];
    for (int k = 0; k < P; k++) {
        const u32 o = (u32)(P - 1 - k);
        const u32 oo = o < n ? o : n;
        u32 cut = 0;
        for (u32 j = 0; j < oo; j++) cut += ((u32)ldg<u8>(t + j) == (u32)tid) ? 1u : 0u;
        L.histAll[((size_t)k * L.nSeg + sgm) * 256 + tid] = c - cut + ((tid == 0) ? oo : 0u);
    }
}

// group-start flags of the sorted keys as a bit map (one ballot per wave): a slot starts a group when its key bytes differ from
// its left neighbour's, when it is the first slot of a block, or when it or its left neighbour is a short suffix
constexpr u32 R0F_ROWS = 8;       // rows of 256 slots per workgroup, all loads of a thread in flight at once (one row per workgroup ran at 2.5 TB/s)
__global__ __launch_bounds__(256) void k_bwt_f_r0_flags(const u64* __restrict__ keys, const u32* __restrict__ base, int nBlocks, u32 total, int P, int pbits,
                                                        unsigned long long* __restrict__ gbits64)
{
    __shared__ int sBlk;
    const u32 a0 = blockIdx.x * (256u * R0F_ROWS);
    if (threadIdx.x == 0) sBlk = find_block(base, nBlocks, a0 < total ? a0 : (total ? total - 1 : 0));
    __syncthreads();
    u64 kr[R0F_ROWS], kl[R0F_ROWS];
#pragma unroll
    for (u32 r = 0; r < R0F_ROWS; r++) {
        const u32 a = a0 + r * 256u + threadIdx.x;
        kr[r] = (a < total) ? keys[a] : 0ull;
        // the left neighbour's key comes from the lane next door (lane 0 of a wave loads it)
        kl[r] = ((threadIdx.x & 63) == 0 && a > 0 && a < total) ? keys[a - 1] : 0ull;
    }
    int b = sBlk;
    const u64 pm = (1ull << pbits) - 1ull;
#pragma unroll
    for (u32 r = 0; r < R0F_ROWS; r++) {
        const u32 a = a0 + r * 256u + threadIdx.x;
        if (a0 + r * 256u >= total) break;                  // (uniform)
        const u64 k = kr[r];
        const u32 klo = (u32)__shfl_up((int)(u32)k, 1u, 64), khi = (u32)__shfl_up((int)(u32)(k >> 32), 1u, 64);
        u64 kp = ((u64)khi << 32) | klo;
        if ((threadIdx.x & 63) == 0) kp = kl[r];
        bool f = true;
        if (a < total) {
            while (a >= base[b + 1]) b++;
            const u32 bb = base[b], n = base[b + 1] - bb;
            f = (a == bb) || ((k >> pbits) != (kp >> pbits)) || ((u32)(k & pm) + (u32)P > n) || ((u32)(kp & pm) + (u32)P > n);
        }
        const unsigned long long m = __ballot(f);
        if ((threadIdx.x & 63) == 0) gbits64[a >> 6] = m;
    }
}

// per window of 2048 slots: the last group start in it (0 when it has none: slot 0 always starts a group, so 0 is neutral
// for the running maximum) and, mirrored for a running minimum from the right, the first one (`total` when none)
__global__ __launch_bounds__(256) void k_bwt_f_r0_winsum(const u32* __restrict__ gbits, u32 total, u32 nWin, u32* __restrict__ winLast, u32* __restrict__ winFirstRev,
                                                         u32 wordsPerWin)
{
    const u32 w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nWin) return;
    u32 last = 0, first = total;
    bool seen = false;
    for (u32 k = 0; k < wordsPerWin; k++) {
        const u32 x = gbits[w * wordsPerWin + k];
        if (x) {
            const u32 b0 = (w * wordsPerWin + k) * 32;
            if (!seen) { first = b0 + (u32)__ffs((int)x) - 1; seen = true; }
            last = b0 + 31 - (u32)__clz((int)x);
        }
    }
    winLast[w] = last < total ? last : (total ? total - 1 : 0);
    winFirstRev[nWin - 1 - w] = first < total ? first : total;
}

// ------------------------------------------------------------------------------------------------
// small groups: bit-map driven windows
// ------------------------------------------------------------------------------------------------
struct SmWindow {
    u32 bw[64];
    int prevSet[64];     // last set bit in the words before w (-1: none)
    u32 nextSet[64];     // first set bit in the words after w (NO_BIT: none)
};

// wave 0 of the workgroup loads the 64 bit-map words of the window and the two per-word summaries; returns false
// (to every thread) when no group that starts in the window is unresolved
__device__ __forceinline__ bool sm_load_window(const FwdView& v, u32 slot0, SmWindow& W, int* sAny)
{
    const int tid = (int)threadIdx.x;
    if (tid < 64) {
        const u32 w = v.gbits[(slot0 >> 5) + (u32)tid];
        W.bw[tid] = w;
        int last = w ? (tid * 32 + 31 - __clz((int)w)) : -1;
        u32 first = w ? (u32)(tid * 32 + __ffs((int)w) - 1) : NO_BIT;
        // exclusive prefix max / suffix min over the 64 words
        int pm = last;
        u32 sm = first;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(pm, o, 64);
            if (tid >= o) pm = t > pm ? t : pm;
            const u32 u = (u32)__shfl_down((int)sm, o, 64);
            if (tid + o < 64) sm = u < sm ? u : sm;
        }
        int pex = __shfl_up(pm, 1, 64);
        if (tid == 0) pex = -1;
        u32 sex = (u32)__shfl_down((int)sm, 1, 64);
        if (tid == 63) sex = NO_BIT;
        W.prevSet[tid] = pex;
        W.nextSet[tid] = sex;
        // anything to do? a zero bit among the owned slots, or right behind them (a group that starts at the last owned slot)
        const bool zero = (tid < (int)(SM_TS / 32)) ? (w != 0xFFFFFFFFu) : (tid == (int)(SM_TS / 32) ? ((w & 1u) == 0) : false);
        const unsigned long long anyz = __ballot(zero);
        if (tid == 0) *sAny = anyz != 0 ? 1 : 0;
    }
    __syncthreads();
    return *sAny != 0;
}

// group [s, e) of window element i; false when i is not a member of a small unresolved group owned by this window
__device__ __forceinline__ bool sm_group_of(const SmWindow& W, u32 i, u32& s, u32& e)
{
    const u32 w = i >> 5, bit = i & 31;
    const u32 lowmask = (bit == 31) ? 0xFFFFFFFFu : ((2u << bit) - 1u);
    const u32 word = W.bw[w];
    const u32 m = word & lowmask;
    int si = m ? (int)(w * 32 + 31 - __clz((int)m)) : W.prevSet[w];
    if (si < 0 || (u32)si >= SM_TS) return false;
    const u32 m2 = word & ~lowmask;
    const u32 ei = m2 ? (w * 32 + (u32)__ffs((int)m2) - 1) : W.nextSet[w];
    if (ei == NO_BIT) return false;
    const u32 len = ei - (u32)si;
    if (len < 2 || len > SM_G) return false;
    s = (u32)si; e = ei;
    return true;
}

// Round 0: SA, ISA and the descriptors of the groups that are not small, one workgroup per window of 2048 slots. The group
// of a slot starts at the last set bit at or before it: found in the window's 64 bit-map words, else in the running
// maximum over the windows before it (winLastIncl); the length of a group (needed where it starts) ends at the next set bit.
__global__ __launch_bounds__(256) void k_bwt_f_r0_place(FwdView v, const u32* __restrict__ vals, const u32* __restrict__ winLastIncl,
                                                        const u32* __restrict__ winFirstInclRev, u32 nWin, uint2* __restrict__ medNext, uint2* __restrict__ largeNext,
                                                        const u64* __restrict__ keys, int nsym, int pbits, uint2* __restrict__ runList)
{
    __shared__ SmWindow W;
    __shared__ int sBlk;
    __shared__ ClassAgg A;
    const int tid = (int)threadIdx.x;
    const u32 win = blockIdx.x;
    const u32 slot0 = win * SM_WIN;
    if (tid == 0) sBlk = find_block(v.base, v.nBlocks, slot0 < v.total ? slot0 : v.total - 1);
    agg_init(A);
    int hKind[SM_WIN / 256];
    u32 hLocal[SM_WIN / 256], hSize[SM_WIN / 256];
    (void)vals;
    if (tid < 64) {
        const u32 w = v.gbits[(slot0 >> 5) + (u32)tid];
        W.bw[tid] = w;
        int pm = w ? (tid * 32 + 31 - __clz((int)w)) : -1;
        u32 sm = w ? (u32)(tid * 32 + __ffs((int)w) - 1) : NO_BIT;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(pm, (unsigned)o, 64);
            if (tid >= o) pm = t > pm ? t : pm;
            const u32 u = (u32)__shfl_down((int)sm, (unsigned)o, 64);
            if (tid + o < 64) sm = u < sm ? u : sm;
        }
        int pex = __shfl_up(pm, 1u, 64);
        if (tid == 0) pex = -1;
        u32 sex = (u32)__shfl_down((int)sm, 1u, 64);
        if (tid == 63) sex = NO_BIT;
        W.prevSet[tid] = pex;
        W.nextSet[tid] = sex;
    }
    __syncthreads();
    const u32 before = win ? winLastIncl[win - 1] : 0u;
    const u32 after = (win + 1 < nWin) ? winFirstInclRev[nWin - 2 - win] : v.total;
    u32 surv = 0;
#pragma unroll
    for (int k = 0; k < (int)(SM_WIN / 256); k++) {
        const u32 i = (u32)tid + 256u * (u32)k;
        const u32 a = slot0 + i;
        hKind[k] = -1; hLocal[k] = 0; hSize[k] = 0;
        if (a >= v.total) continue;
        const u32 w = i >> 5, bit = i & 31;
        const u32 lowmask = (bit == 31) ? 0xFFFFFFFFu : ((2u << bit) - 1u);
        const u32 word = W.bw[w];
        const u32 m = word & lowmask;
        const int si = m ? (int)(w * 32 + 31 - (u32)__clz((int)m)) : W.prevSet[w];
        const u32 hd = (si >= 0) ? slot0 + (u32)si : before;
        int blk = sBlk;
        while (a >= v.base[blk + 1]) blk++;
        const u64 kk = keys[a];
        const u32 pos = (u32)(kk & ((1ull << pbits) - 1ull));
        const u32 gp = v.base[blk] + pos;
        v.SA[a] = gp;
        lab_set(v, gp, v.base[blk], hd, hd);
        if (hd == a) {
            const u32 m2 = word & ~lowmask;
            const u32 ei = m2 ? (w * 32 + (u32)__ffs((int)m2) - 1) : W.nextSet[w];
            u32 nxt = (ei != NO_BIT) ? slot0 + ei : after;
            if (nxt > v.total) nxt = v.total;
            const u32 size = nxt - a;
            // a group whose nsym key bytes are one and the same byte (and whose suffixes are at least nsym long) sits inside runs
            // of that byte: above the small size it is finished by the run-length round instead of log2(run length) doublings
            bool runGroup = false;
            if (runList != nullptr && size > SM_G) {
                const u64 bytes = kk >> pbits;
                u64 rep = 0;
                for (int q = 0; q < nsym; q++) rep = (rep << 8) | (bytes & 0xFF);
                runGroup = (pos + (u32)nsym <= v.base[blk + 1] - v.base[blk]) && bytes == rep;
            }
            hSize[k] = size;
            hKind[k] = agg_note(A, size, runGroup, surv, hLocal[k]);
        }
    }
    if (__ballot(surv != 0) != 0 && (tid & 63) == 0) v.counters[0] = 1;
    __syncthreads();
    if (tid == 0) agg_reserve(A, v);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (int)(SM_WIN / 256); k++)
        if (hKind[k] >= 0) agg_write(A, v, hKind[k], hLocal[k], slot0 + (u32)tid + 256u * (u32)k, hSize[k], largeNext, runList);
}

// Round 0 and the text round in one sweep (the default; knob bwt_no_text_round = 2 runs k_bwt_f_r0_place and k_bwt_f_sort_small_text one
// after the other instead): windows of SM_TS owned slots that look SM_G slots further, as the small-group kernels do. A slot is placed
// by the window that owns it -- except the members of a SMALL group (2..SM_G members), which are all placed by the window that owns
// the group's first slot: that window has their keys in LDS anyway, sorts them on the seven text bytes behind the round-0 symbols
// before anything is written, and writes position and label ONCE, with the label the text order gives (the separate text round read
// SA back, and wrote SA and the labels of every member that moved a second time).
__global__ __launch_bounds__(256) void k_bwt_f_r0_place_text(BwtView bv, FwdView v, const u32* __restrict__ winLastIncl, const u32* __restrict__ winFirstInclRev, u32 nWin,
                                                             uint2* __restrict__ largeNext, const u64* __restrict__ keys, int nsym, int pbits, uint2* __restrict__ runList)
{
    __shared__ SmWindow W;
    __shared__ int sBlk;
    __shared__ ClassAgg A;
    __shared__ u32 sSA[SM_WIN];
    __shared__ u64 sK[SM_WIN];
    __shared__ u32 sNew[64];
    __shared__ u32 sAfterLocal;           // first group start in the owned part of the NEXT window behind what this one looks at
    constexpr int E = (int)(SM_WIN / 256);
    constexpr u32 TSW = SM_TS / 32;       // words a window owns
    const int tid = (int)threadIdx.x;
    const u32 win = blockIdx.x;
    const u32 slot0 = win * SM_TS;
    if (tid == 0) sBlk = find_block(v.base, v.nBlocks, slot0 < v.total ? slot0 : v.total - 1);
    agg_init(A);
    if (tid < 64) {
        sNew[tid] = 0;
        const u32 w = v.gbits[(slot0 >> 5) + (u32)tid];
        W.bw[tid] = w;
        int pm = w ? (tid * 32 + 31 - __clz((int)w)) : -1;
        u32 sm = w ? (u32)(tid * 32 + __ffs((int)w) - 1) : NO_BIT;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(pm, (unsigned)o, 64);
            if (tid >= o) pm = t > pm ? t : pm;
            const u32 u = (u32)__shfl_down((int)sm, (unsigned)o, 64);
            if (tid + o < 64) sm = u < sm ? u : sm;
        }
        int pex = __shfl_up(pm, 1u, 64);
        if (tid == 0) pex = -1;
        u32 sex = (u32)__shfl_down((int)sm, 1u, 64);
        if (tid == 63) sex = NO_BIT;
        W.prevSet[tid] = pex;
        W.nextSet[tid] = sex;
    } else if (tid < 128) {
        // the rest of the next window's owned words (2 TSW - 64 of them): where a group that leaves this window's view ends, if it ends there
        const int k = tid - 64;
        const u32 w = (k < (int)(2 * TSW - 64)) ? v.gbits[(slot0 >> 5) + 64u + (u32)k] : 0u;
        const unsigned long long any = __ballot(w != 0);
        const int first = any ? __ffsll((long long)any) - 1 : 0;          // (this wave alone writes the word: lane 0 says "none")
        if (k == first) sAfterLocal = any ? SM_WIN + (u32)k * 32u + (u32)__ffs((int)w) - 1u : NO_BIT;
    }
    __syncthreads();
    const u32 before = win ? winLastIncl[win - 1] : 0u;
    u32 after = (sAfterLocal != NO_BIT) ? slot0 + sAfterLocal : ((win + 2 < nWin) ? winFirstInclRev[nWin - 3 - win] : v.total);
    if (after > v.total) after = v.total;
    const u64 pmask = (1ull << pbits) - 1ull;
    // ---- every slot in view: position, group, and -- for the members of small groups this window owns -- the text key
    u32 gp[E], gs[E], ge[E];
    u64 kk[E];
    bool inView[E], mine[E];              // mine: member of a small group whose first slot this window owns
    int blkOf[E];
#pragma unroll
    for (int k = 0; k < E; k++) {
        const u32 i = (u32)tid + 256u * (u32)k;
        const u32 a = slot0 + i;
        inView[k] = a < v.total;
        mine[k] = false; gp[k] = 0; kk[k] = 0; gs[k] = 0; ge[k] = 0; blkOf[k] = 0;
        if (!inView[k]) continue;
        int blk = sBlk;
        while (a >= v.base[blk + 1]) blk++;
        blkOf[k] = blk;
        kk[k] = keys[a];
        gp[k] = v.base[blk] + (u32)(kk[k] & pmask);
        mine[k] = sm_group_of(W, i, gs[k], ge[k]);
        if (mine[k]) {
            const u32 bb = v.base[blk], n = v.base[blk + 1] - bb;
            const u32 q = gp[k] - bb + (u32)nsym;
            const u8* t = bv.src[blk];
            const u64 x = text8(t, q, n);
            sSA[i] = gp[k];
            // (SEVEN text bytes and the member's place in its group below them: the key is unique inside the group, and one compare per
            // pair gives the stable rank, a second one against the key without the place the smaller keys alone -- see k_bwt_f_small_fused)
            sK[i] = (__builtin_bswap64(x) & ~0xFFull) | (u64)(i - gs[k]);
        }
    }
    __syncthreads();
    u32 surv = 0;
    int hKind[E];
    u32 hLocal[E], hSize[E];
#pragma unroll
    for (int k = 0; k < E; k++) {
        const u32 i = (u32)tid + 256u * (u32)k;
        const u32 a = slot0 + i;
        hKind[k] = -1; hLocal[k] = 0; hSize[k] = 0;
        if (!inView[k]) continue;
        if (mine[k]) {
            // a member of a small group of this window: its place and label come from the text order inside the group
            const u64 ki = sK[i], lo = ki & ~0xFFull;
            u32 less = 0, rank = 0;
            for (u32 j = gs[k]; j < ge[k]; j++) {
                const u64 kj = sK[j];
                rank += (kj < ki) ? 1u : 0u;
                less += (kj < lo) ? 1u : 0u;
            }
            const u32 eqBefore = rank - less;
            const u32 headIdx = gs[k] + less;
            v.SA[slot0 + headIdx + eqBefore] = gp[k];
            lab_set(v, gp[k], v.base[blkOf[k]], slot0 + headIdx, slot0 + headIdx);
            if (less != 0 && eqBefore == 0) atomicOr(&sNew[headIdx >> 5], 1u << (headIdx & 31));
            if (eqBefore != 0) surv = 1;
            continue;
        }
        if (i >= SM_TS) continue;                                  // in view only: the next window owns it
        // the group of the slot: starts at the last set bit at or before it
        const u32 w = i >> 5, bit = i & 31;
        const u32 lowmask = (bit == 31) ? 0xFFFFFFFFu : ((2u << bit) - 1u);
        const u32 word = W.bw[w];
        const u32 m = word & lowmask;
        const int si = m ? (int)(w * 32 + 31 - (u32)__clz((int)m)) : W.prevSet[w];
        const u32 hd = (si >= 0) ? slot0 + (u32)si : before;
        if (si < 0) {
            // the group began in front of this window. If it is a small one, the window in front places its members (this one included)
            const u32 m2 = word & ~lowmask;
            const u32 ei = m2 ? (w * 32 + (u32)__ffs((int)m2) - 1) : W.nextSet[w];
            const u32 endSlot = (ei != NO_BIT) ? slot0 + ei : after;
            if (endSlot - hd <= SM_G) continue;
        }
        v.SA[a] = gp[k];
        lab_set(v, gp[k], v.base[blkOf[k]], hd, hd);
        if (hd == a) {
            const u32 m2 = word & ~lowmask;
            const u32 ei = m2 ? (w * 32 + (u32)__ffs((int)m2) - 1) : W.nextSet[w];
            u32 nxt = (ei != NO_BIT) ? slot0 + ei : after;
            if (nxt > v.total) nxt = v.total;
            const u32 size = nxt - a;
            if (size <= SM_G) continue;                            // (a singleton; small groups went the other way)
            bool runGroup = false;
            if (runList != nullptr) {
                const u64 bytes = kk[k] >> pbits;
                u64 rep = 0;
                for (int q = 0; q < nsym; q++) rep = (rep << 8) | (bytes & 0xFF);
                const int blk = blkOf[k];
                runGroup = ((u32)(kk[k] & pmask) + (u32)nsym <= v.base[blk + 1] - v.base[blk]) && bytes == rep;
            }
            hSize[k] = size;
            hKind[k] = agg_note(A, size, runGroup, surv, hLocal[k]);
        }
    }
    if (__ballot(surv != 0) != 0 && (tid & 63) == 0) v.counters[0] = 1;
    __syncthreads();
    if (tid == 0) agg_reserve(A, v);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < E; k++)
        if (hKind[k] >= 0) agg_write(A, v, hKind[k], hLocal[k], slot0 + (u32)tid + 256u * (u32)k, hSize[k], largeNext, runList);
    if (tid < 64 && sNew[tid]) atomicOr(&v.gnew[(slot0 >> 5) + (u32)tid], sNew[tid]);
}

__global__ __launch_bounds__(256) void k_bwt_f_gather_small(FwdView v, u32 h, int stats)
{
    __shared__ SmWindow W;
    __shared__ int sAny;
    __shared__ int sBlk;
    __shared__ u32 sRt[64];
    const u32 slot0 = blockIdx.x * SM_TS;
    if (threadIdx.x >= 64 && threadIdx.x < 128) sRt[threadIdx.x - 64] = v.rtbits ? v.rtbits[(slot0 >> 5) + threadIdx.x - 64] : 0u;
    if (!sm_load_window(v, slot0, W, &sAny)) return;
    if (threadIdx.x == 0) sBlk = find_block(v.base, v.nBlocks, slot0 < v.total ? slot0 : v.total - 1);
    __syncthreads();
    const int b0 = sBlk;
#pragma unroll
    for (int k = 0; k < (int)(SM_WIN / 256); k++) {
        const u32 i = threadIdx.x + 256u * k;
        u32 s, e;
        if (!sm_group_of(W, i, s, e)) continue;
        const u32 slot = slot0 + i;
        int b = b0;
        while (slot >= v.base[b + 1]) b++;
        u32 off = h;
        if ((sRt[i >> 5] >> (i & 31)) & 1u) { const u32 r = v.ovr[slot]; off = r > h ? r : h; }
        v.K[slot] = gather_key(v, v.SA[slot], off, v.base[b], v.base[b + 1]);
        if (stats) { atomicAdd(&v.counters[10], 1u); if (i == s) atomicAdd(&v.counters[11], 1u); }      // (developer statistics, knob bwt_stats)
    }
}

__global__ __launch_bounds__(256) void k_bwt_f_sort_small(FwdView v, u32* __restrict__ survTile)
{
    __shared__ SmWindow W;
    __shared__ int sAny;
    __shared__ u32 sSA[SM_WIN];
    __shared__ u32 sK[SM_WIN];
    __shared__ u32 sNew[64];
    __shared__ u32 sWs[4];
    __shared__ int sBlk;
    const u32 slot0 = blockIdx.x * SM_TS;
    if (!sm_load_window(v, slot0, W, &sAny)) { if (threadIdx.x == 0 && survTile) survTile[blockIdx.x] = 0; return; }
    if (threadIdx.x < 64) sNew[threadIdx.x] = 0;
    if (threadIdx.x == 64) sBlk = find_block(v.base, v.nBlocks, slot0 < v.total ? slot0 : v.total - 1);
    u32 gs[SM_WIN / 256], ge[SM_WIN / 256];
    bool act[SM_WIN / 256];
#pragma unroll
    for (int k = 0; k < (int)(SM_WIN / 256); k++) {
        const u32 i = threadIdx.x + 256u * k;
        act[k] = sm_group_of(W, i, gs[k], ge[k]);
        if (act[k]) { sSA[i] = v.SA[slot0 + i]; sK[i] = v.K[slot0 + i]; }
    }
    __syncthreads();
    const int b0 = sBlk;
    u32 surv = 0;
#pragma unroll
    for (int k = 0; k < (int)(SM_WIN / 256); k++) {
        if (!act[k]) continue;
        const u32 i = threadIdx.x + 256u * k;
        const u32 ki = sK[i];
        u32 less = 0, eq = 0, eqBefore = 0;
        for (u32 j = gs[k]; j < ge[k]; j++) {
            const u32 kj = sK[j];
            less += (kj < ki) ? 1u : 0u;
            const u32 same = (kj == ki) ? 1u : 0u;
            eq += same;
            eqBefore += (j < i) ? same : 0u;
        }
        const u32 gp = sSA[i];
        const u32 headIdx = gs[k] + less;
        v.SA[slot0 + headIdx + eqBefore] = gp;
        if (less != 0) {
            u32 bbase = 0;
            if (v.ISA2) { int b = b0; while (slot0 + i >= v.base[b + 1]) b++; bbase = v.base[b]; }
            lab_set(v, gp, bbase, slot0 + headIdx, slot0 + gs[k]);             // (a small group's label is its first slot)
            if (eqBefore == 0) atomicOr(&sNew[headIdx >> 5], 1u << (headIdx & 31));
        }
        if (eq > 1) surv++;
    }
    {
        // members that are still tied, per window (summed by a scan: the host decides by their number whether the next round looks for
        // links first; one counter for all windows would be an atomic per wave on one address)
        const u32 ws = wave_sum(surv);
        if ((threadIdx.x & 63) == 0) { sWs[threadIdx.x >> 6] = ws; if (ws) v.counters[0] = 1; }
    }
    __syncthreads();
    if (threadIdx.x == 0 && survTile) survTile[blockIdx.x] = sWs[0] + sWs[1] + sWs[2] + sWs[3];
    if (threadIdx.x < 64 && sNew[threadIdx.x]) atomicOr(&v.gnew[(slot0 >> 5) + threadIdx.x], sNew[threadIdx.x]);
}

// Keys and refinement of the small groups in ONE sweep (versioned labels, see lab_old): k_bwt_f_gather_small + k_bwt_f_sort_small without the
// key array in between -- a window's bit-map words, positions and keys are read once and stay in LDS (the two kernels read the bit map and
// SA twice and wrote and read K: 24 of the 40 KiB a dense window moved per round).
// PACKED (blocks of up to 8 MiB: a key -- a label inside the block + 1 -- has 24 bits): the key in LDS is (key << 8 | place in the group), unique
// inside the group, so that ONE compare per pair gives a member its stable rank (smaller keys + equal keys in front of it) and a second one
// against (key << 8) the smaller keys alone; the plain form needs "smaller", "equal" and "equal and in front" -- 9.4 vector instructions
// per pair against 4, and a member of a group of g compares with all g (round 6: groups of 33..256 members are a third of the small-group
// visits of the first rounds on real files).
template <bool PACKED>
__global__ __launch_bounds__(256) void k_bwt_f_small_fused(FwdView v, u32 h, u32* __restrict__ survTile, int stats)
{
    __shared__ SmWindow W;
    __shared__ int sAny;
    __shared__ int sBlk;
    __shared__ u32 sRt[64];
    __shared__ u32 sSA[SM_WIN];
    __shared__ u32 sK[SM_WIN];
    __shared__ u32 sNew[64];
    __shared__ u32 sWs[4];
    const u32 slot0 = blockIdx.x * SM_TS;
    if (threadIdx.x >= 64 && threadIdx.x < 128) sRt[threadIdx.x - 64] = v.rtbits ? v.rtbits[(slot0 >> 5) + threadIdx.x - 64] : 0u;
    if (!sm_load_window(v, slot0, W, &sAny)) { if (threadIdx.x == 0 && survTile) survTile[blockIdx.x] = 0; return; }
    if (threadIdx.x < 64) sNew[threadIdx.x] = 0;
    if (threadIdx.x == 64) sBlk = find_block(v.base, v.nBlocks, slot0 < v.total ? slot0 : v.total - 1);
    __syncthreads();
    const int b0 = sBlk;
    u32 gs[SM_WIN / 256], ge[SM_WIN / 256], bb[SM_WIN / 256];
    bool act[SM_WIN / 256];
#pragma unroll
    for (int k = 0; k < (int)(SM_WIN / 256); k++) {
        const u32 i = threadIdx.x + 256u * k;
        act[k] = sm_group_of(W, i, gs[k], ge[k]);
        bb[k] = 0;
        if (!act[k]) continue;
        const u32 slot = slot0 + i;
        int b = b0;
        while (slot >= v.base[b + 1]) b++;
        bb[k] = v.base[b];
        u32 off = h;
        if ((sRt[i >> 5] >> (i & 31)) & 1u) { const u32 r = v.ovr[slot]; off = r > h ? r : h; }
        const u32 gp = v.SA[slot];
        sSA[i] = gp;
        const u32 key = gather_key(v, gp, off, bb[k], v.base[b + 1]);
        sK[i] = PACKED ? ((key << 8) | (i - gs[k])) : key;
        if (stats) { atomicAdd(&v.counters[10], 1u); if (i == gs[k]) atomicAdd(&v.counters[11], 1u); }      // (developer statistics, knob bwt_stats)
    }
    __syncthreads();
    u32 surv = 0;
#pragma unroll
    for (int k = 0; k < (int)(SM_WIN / 256); k++) {
        if (!act[k]) continue;
        const u32 i = threadIdx.x + 256u * k;
        const u32 ki = sK[i];
        u32 less = 0, eqBefore = 0, tied = 0;
        if (PACKED) {
            const u32 lo = ki & ~0xFFu;
            u32 rank = 0;
            for (u32 j = gs[k]; j < ge[k]; j++) {
                const u32 kj = sK[j];
                rank += (kj < ki) ? 1u : 0u;
                less += (kj < lo) ? 1u : 0u;
            }
            eqBefore = rank - less;
            // (members that are still tied: in a subgroup of m equal keys m - 1 have an equal key in front of them, and exactly one of
            // those has exactly one)
            tied = (eqBefore != 0 ? 1u : 0u) + (eqBefore == 1 ? 1u : 0u);
        } else {
            u32 eq = 0;
            for (u32 j = gs[k]; j < ge[k]; j++) {
                const u32 kj = sK[j];
                less += (kj < ki) ? 1u : 0u;
                const u32 same = (kj == ki) ? 1u : 0u;
                eq += same;
                eqBefore += (j < i) ? same : 0u;
            }
            tied = (eq > 1) ? 1u : 0u;
        }
        const u32 gp = sSA[i];
        const u32 headIdx = gs[k] + less;
        v.SA[slot0 + headIdx + eqBefore] = gp;
        if (less != 0) {
            lab_set(v, gp, bb[k], slot0 + headIdx, slot0 + gs[k]);             // (a small group's label is its first slot)
            if (eqBefore == 0) atomicOr(&sNew[headIdx >> 5], 1u << (headIdx & 31));
        }
        surv += tied;
    }
    {
        const u32 ws = wave_sum(surv);
        if ((threadIdx.x & 63) == 0) { sWs[threadIdx.x >> 6] = ws; if (ws) v.counters[0] = 1; }
    }
    __syncthreads();
    if (threadIdx.x == 0 && survTile) survTile[blockIdx.x] = sWs[0] + sWs[1] + sWs[2] + sWs[3];
    if (threadIdx.x < 64 && sNew[threadIdx.x]) atomicOr(&v.gnew[(slot0 >> 5) + threadIdx.x], sNew[threadIdx.x]);
}

// The small groups of round 0 once more, on the EIGHT TEXT BYTES behind the sorted symbols (zero past the block end): the keys come
// from the text, which nobody changes, so gathering and sorting are one kernel (no key array, no ordering constraint between
// windows), and a member pays one random read and one label write for eight symbols of depth where a doubling round at h = 4 buys
// four. Most small groups of round 0 are resolved here and never enter a doubling round; ties (also: a suffix that ends inside the
// eight bytes against one that continues with zeros) stay groups and go on as usual.
__global__ __launch_bounds__(256) void k_bwt_f_sort_small_text(BwtView bv, FwdView v, u32 off)
{
    __shared__ SmWindow W;
    __shared__ int sAny;
    __shared__ int sBlk;
    __shared__ u32 sSA[SM_WIN];
    __shared__ u64 sK[SM_WIN];
    __shared__ u32 sNew[64];
    const u32 slot0 = blockIdx.x * SM_TS;
    if (!sm_load_window(v, slot0, W, &sAny)) return;
    if (threadIdx.x < 64) sNew[threadIdx.x] = 0;
    if (threadIdx.x == 64) sBlk = find_block(v.base, v.nBlocks, slot0 < v.total ? slot0 : v.total - 1);
    __syncthreads();
    const int b0 = sBlk;
    u32 gs[SM_WIN / 256], ge[SM_WIN / 256];
    bool act[SM_WIN / 256];
#pragma unroll
    for (int k = 0; k < (int)(SM_WIN / 256); k++) {
        const u32 i = threadIdx.x + 256u * k;
        act[k] = sm_group_of(W, i, gs[k], ge[k]);
        if (!act[k]) continue;
        const u32 slot = slot0 + i;
        int b = b0;
        while (slot >= v.base[b + 1]) b++;
        const u32 bb = v.base[b], n = v.base[b + 1] - bb;
        const u32 gp = v.SA[slot];
        const u32 q = gp - bb + off;
        const u8* t = bv.src[b];
        const u64 x = text8(t, q, n);
        sSA[i] = gp;
        sK[i] = __builtin_bswap64(x);
    }
    __syncthreads();
    u32 surv = 0;
#pragma unroll
    for (int k = 0; k < (int)(SM_WIN / 256); k++) {
        if (!act[k]) continue;
        const u32 i = threadIdx.x + 256u * k;
        const u64 ki = sK[i];
        u32 less = 0, eq = 0, eqBefore = 0;
        for (u32 j = gs[k]; j < ge[k]; j++) {
            const u64 kj = sK[j];
            less += (kj < ki) ? 1u : 0u;
            const u32 same = (kj == ki) ? 1u : 0u;
            eq += same;
            eqBefore += (j < i) ? same : 0u;
        }
        const u32 gp = sSA[i];
        const u32 headIdx = gs[k] + less;
        v.SA[slot0 + headIdx + eqBefore] = gp;
        if (less != 0) {
            int b = b0;
            while (slot0 + i >= v.base[b + 1]) b++;
            lab_set(v, gp, v.base[b], slot0 + headIdx, slot0 + gs[k]);
            if (eqBefore == 0) atomicOr(&sNew[headIdx >> 5], 1u << (headIdx & 31));
        }
        if (eq > 1) surv = 1;
    }
    if (__ballot(surv != 0) != 0 && (threadIdx.x & 63) == 0) v.counters[0] = 1;
    __syncthreads();
    if (threadIdx.x < 64 && sNew[threadIdx.x]) atomicOr(&v.gnew[(slot0 >> 5) + threadIdx.x], sNew[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// small groups inside long repeats: the link step (round 5)
// ------------------------------------------------------------------------------------------------
// A repeat of length L is L tied groups {p + i, q + i, ...}, i = 0 .. L - 1, and prefix doubling visits each of them log2(L - i) times:
// real files (the same licence text in front of thousands of headers, string tables, duplicate objects) spend most of the suffix sort
// there. All those groups are ordered alike -- by what follows the repeat's end. A small group G is LINKED when the successors p + 1 of
// all its members lie in one group (which may hold other suffixes as well). Every member of a linked group is flagged at its own
// position in a bit map over the positions, so the groups of one repeat are runs of consecutive set bits, and a run's end -- the first
// position that is not flagged -- is c positions on, the same c for every member of G (they stay together all the way): the members of
// G share c symbols more than their depth says, and the group at the run's end is what tells them apart. The existing "look max(R, h)
// positions on" of the key kernels (FwdView::ovr, rtbits) does the rest: ovr = c + h for the members of G, and G is sorted in this
// round by the labels c + h positions on -- together with the group at the end of its run, instead of log2 c rounds later. Exact: a
// link means the successors are in ONE group, i.e. equal in at least their first symbol; c links in a row mean c equal symbols, and the
// group at the end has depth h. Only groups whose positions lie below posEnd take part (0: below the end of the first block -- the trial
// of the host policy covers that block only, and its bit map ends there: the window of the next block that rides along must neither
// flag nor look up anything).
// a trial looks at up to four whole blocks of the batch (a run never leaves its block, so the blocks' runs are complete)
struct LinkTrial { int n; int blk[4]; };

__device__ __forceinline__ bool link_tile(const FwdView& v, const LinkTrial& tr, u32 tileStep, u32& tile, u32& posLo, u32& posHi)
{
    if (tr.n == 0) { tile = blockIdx.x * tileStep; posLo = 0; posHi = v.total; return true; }
    const int b = tr.blk[blockIdx.y];
    posLo = v.base[b]; posHi = v.base[b + 1];
    tile = posLo / SM_TS + blockIdx.x;                        // (the window that holds the block's first slot, and the ones behind it)
    return tile * SM_TS < posHi;
}

__global__ __launch_bounds__(256) void k_bwt_f_link_small(FwdView v, u32* __restrict__ posflag, u32* __restrict__ linked, int stats, u32 tileStep,
                                                          u32* __restrict__ linkedTile, LinkTrial tr)
{
    __shared__ SmWindow W;
    __shared__ int sAny;
    __shared__ int sBlk;
    __shared__ u32 sP[SM_WIN];
    __shared__ u32 sK[SM_WIN];
    __shared__ u32 sLinked[64];
    __shared__ u32 sCnt[4];
    u32 tile, posLo, posHi;
    if (!link_tile(v, tr, tileStep, tile, posLo, posHi)) return;
    const u32 slot0 = tile * SM_TS;
    if (!sm_load_window(v, slot0, W, &sAny)) { if (threadIdx.x == 0) linkedTile[tile] = 0; return; }
    if (threadIdx.x < 64) sLinked[threadIdx.x] = 0;
    if (threadIdx.x == 64) sBlk = find_block(v.base, v.nBlocks, slot0 < v.total ? slot0 : v.total - 1);
    __syncthreads();
    const int b0 = sBlk;
    u32 gs[SM_WIN / 256], ge[SM_WIN / 256];
    bool act[SM_WIN / 256];
#pragma unroll
    for (int k = 0; k < (int)(SM_WIN / 256); k++) {
        const u32 i = threadIdx.x + 256u * k;
        act[k] = sm_group_of(W, i, gs[k], ge[k]);
        if (!act[k]) continue;
        const u32 slot = slot0 + i;
        int b = b0;
        while (slot >= v.base[b + 1]) b++;
        const u32 gp = v.SA[slot];
        sP[i] = gp;
        sK[i] = (gp + 1 < v.base[b + 1]) ? lab_old(v, gp + 1, v.base[b]) : 0xFFFFFFFFu;      // the label of the successor; none at the block's end
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (int)(SM_WIN / 256); k++) {
        const u32 i = threadIdx.x + 256u * k;
        if (!act[k] || i != gs[k]) continue;
        const u32 lab = sK[i];
        bool ok = lab != 0xFFFFFFFFu && sP[i] >= posLo && sP[i] < posHi;         // (one group, one block: its first member speaks for all)
        for (u32 j = gs[k] + 1; j < ge[k] && ok; j++) ok = sK[j] == lab;
        if (ok) atomicOr(&sLinked[i >> 5], 1u << (i & 31));
    }
    __syncthreads();
    u32 mine = 0;
    // every member of a linked group is flagged at its own position: the group one position on may hold other suffixes as well (they do
    // not matter: what counts is that THESE members stay together), so a run is followed member by member, not group by group
#pragma unroll
    for (int k = 0; k < (int)(SM_WIN / 256); k++) {
        if (!act[k]) continue;
        if (!((sLinked[gs[k] >> 5] >> (gs[k] & 31)) & 1u)) continue;
        const u32 gp = sP[threadIdx.x + 256u * k];
        atomicOr(&posflag[gp >> 5], 1u << (gp & 31));
        mine++;
    }
    (void)stats;
    {
        const u32 ws = wave_sum(mine);
        if ((threadIdx.x & 63) == 0) sCnt[threadIdx.x >> 6] = ws;
    }
    __syncthreads();
    if (threadIdx.x == 0) linkedTile[tile] = sCnt[0] + sCnt[1] + sCnt[2] + sCnt[3];       // members in linked groups: what the step may pay for
    if (threadIdx.x < 64 && sLinked[threadIdx.x]) atomicOr(&linked[(slot0 >> 5) + threadIdx.x], sLinked[threadIdx.x]);
}

// what the link step bought: members it linked against members still tied after the round, over the windows it was applied to. A trial
// on sample blocks pays when it pays in EVERY one of them (a batch of different files: the first block of the real-file corpus is one
// shared object whose groups are whole repeats, the header files behind it are not -- applied to the whole batch the step cost 3.5 ms
// there and bought 1.7): a block that does not pay is reported as "everything still tied".
__global__ __launch_bounds__(256) void k_bwt_f_link_payoff(FwdView v, const u32* __restrict__ linkedTile, const u32* __restrict__ survTile, u32 nTiles, u32 tileStep,
                                                           u32* __restrict__ out, LinkTrial tr)
{
    __shared__ u32 sA[256], sB[256];
    u32 totA = 0, totB = 0;
    bool every = true;
    const int nPart = tr.n ? tr.n : 1;
    for (int q = 0; q < nPart; q++) {
        u32 t0 = 0, t1 = nTiles, step = tileStep;
        if (tr.n) { t0 = v.base[tr.blk[q]] / SM_TS; t1 = (v.base[tr.blk[q] + 1] + SM_TS - 1) / SM_TS; step = 1; if (t1 > nTiles) t1 = nTiles; }
        u32 a = 0, b = 0;
        for (u32 t = t0 + threadIdx.x * step; t < t1; t += 256u * step) { a += linkedTile[t]; b += survTile[t]; }
        sA[threadIdx.x] = a; sB[threadIdx.x] = b;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) { sA[threadIdx.x] += sA[threadIdx.x + o]; sB[threadIdx.x] += sB[threadIdx.x + o]; } __syncthreads(); }
        totA += sA[0]; totB += sB[0];
        if (tr.n && !(sA[0] >= 1024u && sB[0] < sA[0] / 2)) every = false;
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = totA; out[1] = every ? totB : (totA > totB ? totA : totB); }
}

// first zero bit of every word of the link bit map, by word, in reversed order (an inclusive minimum scan over it gives, for every word,
// the first zero bit at or behind it)
__global__ __launch_bounds__(256) void k_bwt_f_link_zeros(const u32* __restrict__ posflag, u32 nW, u32* __restrict__ rev)
{
    const u32 w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nW) return;
    const u32 x = ~posflag[w];
    rev[nW - 1 - w] = x ? 32u * w + (u32)__ffs((int)x) - 1u : 0xFFFFFFFFu;
}

__device__ __forceinline__ u32 link_next_zero(const u32* __restrict__ posflag, const u32* __restrict__ sufRev, u32 nW, u32 p)
{
    const u32 w = p >> 5;
    const u32 x = ~posflag[w] & (0xFFFFFFFFu << (p & 31));
    if (x) return 32u * w + (u32)__ffs((int)x) - 1u;
    return (w + 1 < nW) ? sufRev[nW - 2 - w] : 0xFFFFFFFFu;
}

__global__ __launch_bounds__(256) void k_bwt_f_link_apply(FwdView v, const u32* __restrict__ posflag, const u32* __restrict__ sufRev, u32 nW,
                                                          const u32* __restrict__ linked, u32 h, u32* __restrict__ ovr, u32* __restrict__ rtbits, u32 tileStep, LinkTrial tr)
{
    __shared__ SmWindow W;
    __shared__ int sAny;
    __shared__ u32 sLinked[64];
    __shared__ u32 sRt[64];
    __shared__ u32 sOff[SM_WIN];
    u32 tile, posLo, posHi;
    if (!link_tile(v, tr, tileStep, tile, posLo, posHi)) return;
    const u32 slot0 = tile * SM_TS;
    if (threadIdx.x >= 64 && threadIdx.x < 128) { sLinked[threadIdx.x - 64] = linked[(slot0 >> 5) + threadIdx.x - 64]; sRt[threadIdx.x - 64] = 0; }
    if (!sm_load_window(v, slot0, W, &sAny)) return;
    __syncthreads();
    u32 gs[SM_WIN / 256], ge[SM_WIN / 256];
    bool act[SM_WIN / 256];
#pragma unroll
    for (int k = 0; k < (int)(SM_WIN / 256); k++) {
        const u32 i = threadIdx.x + 256u * k;
        act[k] = sm_group_of(W, i, gs[k], ge[k]) && ((sLinked[gs[k] >> 5] >> (gs[k] & 31)) & 1u);
        if (act[k] && i == gs[k]) {
            const u32 gp = v.SA[slot0 + i];                                      // (any member: the run is as long for all of them)
            const u32 e = link_next_zero(posflag, sufRev, nW, gp);
            sOff[i] = (e != 0xFFFFFFFFu && e > gp) ? (e - gp) + h : 0u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (int)(SM_WIN / 256); k++) {
        if (!act[k]) continue;
        const u32 i = threadIdx.x + 256u * k;
        const u32 off = sOff[gs[k]];
        if (off == 0) continue;
        const u32 slot = slot0 + i;
        const bool had = (rtbits[slot >> 5] >> (slot & 31)) & 1u;
        const u32 old = had ? ovr[slot] : 0u;
        if (off > old) ovr[slot] = off;
        if (!had) atomicOr(&sRt[i >> 5], 1u << (i & 31));
    }
    __syncthreads();
    if (threadIdx.x < 64 && sRt[threadIdx.x]) atomicOr(&rtbits[(slot0 >> 5) + threadIdx.x], sRt[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// medium groups: one workgroup per descriptor
// ------------------------------------------------------------------------------------------------
// The descriptors arrive sorted by start slot: neighbours in the list are neighbouring groups, and in periodic data (the
// members of group v+1 are the members of group v moved by one position) their members look at neighbouring words of ISA.
// The 32 workgroups that run on one XCD (workgroup index mod 8 -- an observed placement, used for speed only) walk through one
// contiguous eighth of the list side by side, so that a line of ISA fetched for one group is found in that XCD's L2 by the
// 31 groups next to it, instead of being fetched from memory once per group.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_bwt_f_gather_desc(FwdView v, const uint2* __restrict__ desc, u32 nDesc, u32 h, uint2* __restrict__ descInfo, int stats)
{
    __shared__ int sBlk;
    const u32 xcd = blockIdx.x & 7, lanesPerXcd = gridDim.x >> 3, slot = blockIdx.x >> 3;      // the grid is a multiple of 8 workgroups
    const u32 per = (nDesc + 7) / 8;
    const u32 lo = xcd * per, hi = (lo + per < nDesc) ? lo + per : nDesc;
    for (u32 g = lo + slot; g < hi; g += lanesPerXcd) {
        const uint2 d = desc[g];
        if (threadIdx.x == 0) sBlk = find_block(v.base, v.nBlocks, d.x);
        __syncthreads();
        const u32 bb = v.base[sBlk], be = v.base[sBlk + 1];
        u32 off = h;                                        // (uniform for the group)
        if (v.rtbits && ((v.rtbits[d.x >> 5] >> (d.x & 31)) & 1u)) { const u32 r = v.ovr[d.x]; off = r > h ? r : h; }
        // (block base, label the members carry): what the sorting kernel needs per group without a chain of dependent loads of its own
        if (threadIdx.x == 0) { descInfo[g] = make_uint2(bb, lab_old(v, v.SA[d.x], bb)); if (stats) atomicAdd(&v.counters[12], d.y); }
        // eight members per thread at a time: all position loads, then all key loads, then the stores -- two memory
        // latencies per batch instead of two per member
        for (u32 i0 = 0; i0 < d.y; i0 += 8 * THREADS) {
            u32 gp[8], key[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { const u32 i = i0 + (u32)k * THREADS + threadIdx.x; gp[k] = (i < d.y) ? v.SA[d.x + i] : bb; }
#pragma unroll
            for (int k = 0; k < 8; k++) { const u32 i = i0 + (u32)k * THREADS + threadIdx.x; key[k] = (i < d.y) ? gather_key(v, gp[k], off, bb, be) : 0u; }
#pragma unroll
            for (int k = 0; k < 8; k++) { const u32 i = i0 + (u32)k * THREADS + threadIdx.x; if (i < d.y) v.K[d.x + i] = key[k]; }
        }
        __syncthreads();
    }
}

// LDS of one sorting workgroup of THREADS threads: up to ROWS elements per thread
template <int THREADS, int ROWS>
struct MedLds {
    static constexpr int WAVES = THREADS / 64;
    static constexpr u32 CAP = (u32)ROWS * THREADS;
    u32 oK[CAP];
    u32 oV[CAP];
    u32 cnt[WAVES * 256];
    u32 fb[CAP / 32];
    int pm[CAP / 32];
    u32 pn[CAP / 32];
    u32 wtot[WAVES];
    u32 wtot2[WAVES];
    u32 oldLab;                 // the label the group's members carry
    u32 blkBase;                // first slot of the group's block (labels are stored relative to it: lab_set)
};

template <int THREADS, int ROWS>
__device__ __forceinline__ u32 med_block_sum(MedLds<THREADS, ROWS>& L, u32 x)
{
    const u32 s = wave_sum(x);
    if ((threadIdx.x & 63) == 0) L.wtot[threadIdx.x >> 6] = s;
    __syncthreads();
    u32 tot = 0;
    for (int w = 0; w < THREADS / 64; w++) tot += L.wtot[w];
    __syncthreads();
    return tot;
}

// Stable LSD radix sort of the pairs (oK, oV)[0, n) in LDS, 8-bit digits. Wave w owns the R*64 consecutive elements from
// w*R*64 (R = rows of 64 per wave); the rank of an element among the elements of its wave with the same digit comes from
// ballot matching (the lanes of a row that agree on all 8 digit bits), rows in order; per-wave digit counters are then
// scanned in (digit, wave) order. Entered and left with the workgroup in step.
template <int THREADS, int ROWS>
__device__ __forceinline__ void med_radix_sort(MedLds<THREADS, ROWS>& L, u32 n, int npass)
{
    constexpr int WAVES = THREADS / 64;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    u32* cntw = L.cnt + wave * 256;
    const int R = (int)((n + THREADS - 1) / THREADS);
    const u32 waveBase = (u32)wave * (u32)R * 64u;
    for (int pass = 0; pass < npass; pass++) {
        const int shift = 8 * pass;
        for (int q = lane; q < 256; q += 64) cntw[q] = 0;
        u32 key[ROWS], val[ROWS], pre[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            key[r] = 0xFFFFFFFFu; val[r] = 0; pre[r] = 0;
            if (r < R) {
                const u32 idx = waveBase + (u32)r * 64u + (u32)lane;
                if (idx < n) { key[r] = L.oK[idx]; val[r] = L.oV[idx]; }
                const u32 dg = (key[r] >> shift) & 255u;
                const unsigned long long peers = prims::digit_peers_fast(true, dg);
                const u32 pop = (u32)__popcll(peers);
                const u32 rnk = (u32)__popcll(peers & ltMask);
                const int leader = __ffsll((long long)peers) - 1;
                u32 old = 0;
                if (lane == leader) { old = cntw[dg]; cntw[dg] = old + pop; }
                old = (u32)__shfl((int)old, leader, 64);
                pre[r] = old + rnk;
            }
        }
        __syncthreads();
        // exclusive scan of the counters in (digit, wave) order: a thread owns one digit and four consecutive waves
        {
            constexpr int TPD = THREADS / 256;
            const int dg = tid / TPD, w0 = (tid % TPD) * 4;
            const u32 c0 = L.cnt[(w0 + 0) * 256 + dg], c1 = L.cnt[(w0 + 1) * 256 + dg], c2 = L.cnt[(w0 + 2) * 256 + dg], c3 = L.cnt[(w0 + 3) * 256 + dg];
            const u32 sum = c0 + c1 + c2 + c3;
            const u32 incl = wave_incl_scan(sum);
            if (lane == 63) L.wtot[wave] = incl;
            __syncthreads();
            u32 wbase = 0;
            for (int w = 0; w < wave; w++) wbase += L.wtot[w];
            const u32 excl = wbase + incl - sum;
            L.cnt[(w0 + 0) * 256 + dg] = excl;
            L.cnt[(w0 + 1) * 256 + dg] = excl + c0;
            L.cnt[(w0 + 2) * 256 + dg] = excl + c0 + c1;
            L.cnt[(w0 + 3) * 256 + dg] = excl + c0 + c1 + c2;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            if (r < R) {
                const u32 dg = (key[r] >> shift) & 255u;
                const u32 pos = cntw[dg] + pre[r];
                if (pos < n) {                         // the padding of the last row sorts behind everything: not stored
                    L.oK[pos] = key[r];
                    L.oV[pos] = val[r];
                }
            }
        }
        __syncthreads();
    }
    (void)WAVES;
}

// Write-back of a group sorted in LDS (oK = keys in order, oV = positions): subgroup boundaries, SA, labels, the round's bit map
// and the children's descriptors. Entered and left with the workgroup in step.
template <int THREADS, int ROWS>
__device__ __forceinline__ void med_write_back(MedLds<THREADS, ROWS>& L, const FwdView& v, u32 gs, u32 n, uint2* __restrict__ medNext, uint2* __restrict__ largeNext)
{
    constexpr u32 CAP = (u32)ROWS * THREADS;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    // ---- subgroup boundaries of the sorted keys as a bit map in LDS + per-word summaries
    const int nWords = (int)((n + 31) >> 5);
    if (tid < (int)(CAP / 32)) {
        u32 bits = 0;
        if (tid < nWords) {
            const u32 i0 = (u32)tid * 32u;
            u32 prev = (i0 == 0) ? 0u : L.oK[i0 - 1];
            for (u32 k = 0; k < 32; k++) {
                const u32 i = i0 + k;
                if (i >= n) break;
                const u32 cur = L.oK[i];
                if (i == 0 || cur != prev) bits |= 1u << k;
                prev = cur;
            }
        }
        L.fb[tid] = bits;
        L.pm[tid] = bits ? (tid * 32 + 31 - __clz((int)bits)) : -1;
        L.pn[tid] = bits ? (u32)(tid * 32 + __ffs((int)bits) - 1) : n;
    }
    __syncthreads();
    // inclusive prefix max of pm / suffix min of pn over the words, by the first wave (WPL consecutive words per lane)
    if (tid < 64) {
        constexpr int WPL = (int)(CAP / 32) / 64;
        int run = -1;
        for (int k = 0; k < WPL; k++) { const int x = L.pm[tid * WPL + k]; run = x > run ? x : run; }
        int incl = run;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, (unsigned)o, 64); if (tid >= o) incl = t > incl ? t : incl; }
        int carry = __shfl_up(incl, 1u, 64);
        if (tid == 0) carry = -1;
        for (int k = 0; k < WPL; k++) { const int x = L.pm[tid * WPL + k]; carry = x > carry ? x : carry; L.pm[tid * WPL + k] = carry; }
        u32 runn = n;
        for (int k = WPL - 1; k >= 0; k--) { const u32 x = L.pn[tid * WPL + k]; runn = x < runn ? x : runn; }
        u32 incn = runn;
        for (int o = 1; o < 64; o <<= 1) { const u32 t = (u32)__shfl_down((int)incn, (unsigned)o, 64); if (tid + o < 64) incn = t < incn ? t : incn; }
        u32 carn = (u32)__shfl_down((int)incn, 1u, 64);
        if (tid == 63) carn = n;
        for (int k = WPL - 1; k >= 0; k--) { const u32 x = L.pn[tid * WPL + k]; carn = x < carn ? x : carn; L.pn[tid * WPL + k] = carn; }
    }
    __syncthreads();
    // pm[w] = last boundary at or before the end of word w, pn[w] = first boundary at or after the start of word w
    u32 surv = 0;
    const u32 oldLab = L.oldLab;
    __shared__ ClassAgg A;
    agg_init(A);
    __syncthreads();
    int hKind[ROWS];
    u32 hLocal[ROWS], hSize[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++) hKind[r] = -1;
    int row = 0;
    for (u32 i = (u32)tid; i < n; i += THREADS, row++) {
        const u32 w = i >> 5, bit = i & 31;
        const u32 lowmask = (bit == 31) ? 0xFFFFFFFFu : ((2u << bit) - 1u);
        const u32 word = L.fb[w];
        const u32 mm = word & lowmask;
        const u32 hd = mm ? (w * 32 + 31 - (u32)__clz((int)mm)) : (u32)L.pm[w - 1];      // word 0 always has bit 0
        const u32 gp = L.oV[i];
        v.SA[gs + i] = gp;
        const u32 m2 = word & ~lowmask;
        const u32 e = m2 ? (w * 32 + (u32)__ffs((int)m2) - 1) : ((w + 1 < (u32)(CAP / 32)) ? L.pn[w + 1] : n);     // end of my subgroup
        // The label of a group only has to be a slot inside the group's range (ranges are disjoint, so labels stay unique and
        // ordered). Small children take their first slot, as every kernel that works on small groups expects; a larger child
        // keeps the parent's label when that slot lies inside its range, else it takes its middle slot -- which the part that
        // holds the majority in the rounds to come will still contain, so that its thousands of members are not relabelled
        // (a scattered 4-byte store each) round after round just because a few members left in front of them.
        const u32 size = e - hd;
        const u32 rel = oldLab - gs;
        const u32 lab = (size <= SM_G) ? gs + hd : ((rel >= hd && rel < e) ? oldLab : gs + hd + (size >> 1));
        if (lab != oldLab) lab_set(v, gp, L.blkBase, lab, oldLab);
        if (hd == i) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) if (r == row) { hSize[r] = size; hKind[r] = agg_note(A, size, false, surv, hLocal[r]); }
        }
    }
    if (__ballot(surv != 0) != 0 && lane == 0) v.counters[0] = 1;
    __syncthreads();
    if (tid == 0) agg_reserve(A, v);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROWS; r++)
        if (hKind[r] >= 0) agg_write(A, v, hKind[r], hLocal[r], gs + (u32)tid + (u32)r * THREADS, hSize[r], largeNext, (uint2*)nullptr);
    // new group starts into the round's bit map (bit 0 of the group is set already)
    if (tid < nWords && L.fb[tid]) {
        const u32 off = gs + (u32)tid * 32u;
        const u32 sh = off & 31;
        atomicOr(&v.gnew[off >> 5], L.fb[tid] << sh);
        if (sh && (L.fb[tid] >> (32 - sh))) atomicOr(&v.gnew[(off >> 5) + 1], L.fb[tid] >> (32 - sh));
    }
}

// Write-back of a group whose majority key m is held by c members, with the nOth <= THREADS others sorted in oK / oV [0, nOth) and the
// majority's members behind them in their old (position) order: the result is [others < m][majority][others > m], `less` others in front.
template <int THREADS, int ROWS>
__device__ __forceinline__ void med_majority_write_back(MedLds<THREADS, ROWS>& L, const FwdView& v, u32 gs, u32 n, u32 nOth, u32 c, u32 less, u32 m)
{
    const int tid = (int)threadIdx.x;
    const u32 oldLab = L.oldLab;
    (void)n; (void)m;
    // ---- the majority: c members from slot gs + less on. It keeps the group's label when that slot is inside its range (see med_write_back)
    const u32 mStart = gs + less;
    const u32 rel = oldLab - gs;
    const u32 mLab = (c <= SM_G) ? mStart : ((rel >= less && rel < less + c) ? oldLab : mStart + (c >> 1));
    for (u32 i = (u32)tid; i < c; i += THREADS) {
        const u32 gp = L.oV[nOth + i];
        v.SA[mStart + i] = gp;
        if (mLab != oldLab) lab_set(v, gp, L.blkBase, mLab, oldLab);
    }
    u32 surv = 0;
    if (tid == 0) {
        if (less != 0) atomicOr(&v.gnew[mStart >> 5], 1u << (mStart & 31));
        if (c > SM_G) v.medStage[mStart >> 8] = make_uint2(mStart, c);      // (c <= n <= MED_CAP)
        else if (c > 1) surv = 1;
    }
    // ---- the others: one thread each
    if ((u32)tid < nOth) {
        const u32 t = (u32)tid;
        const u32 key = L.oK[t];
        u32 hd = t, e = t + 1;
        while (hd > 0 && L.oK[hd - 1] == key) hd--;
        while (e < nOth && L.oK[e] == key) e++;
        const u32 size = e - hd;
        const u32 shift = (t < less) ? 0u : c;                          // (equal keys are on one side of the majority)
        const u32 slot = gs + t + shift, hdSlot = gs + hd + shift;
        const u32 relO = rel - shift;                                   // the parent's label in the others' own index space (wraps when it is not there)
        const u32 lab = (size <= SM_G) ? hdSlot : ((oldLab >= hdSlot && oldLab < hdSlot + size) ? oldLab : hdSlot + (size >> 1));
        (void)relO;
        const u32 gp = L.oV[t];
        v.SA[slot] = gp;
        if (lab != oldLab) lab_set(v, gp, L.blkBase, lab, oldLab);
        if (t == hd) {
            if (hdSlot != gs) atomicOr(&v.gnew[hdSlot >> 5], 1u << (hdSlot & 31));
            if (size > SM_G) v.medStage[hdSlot >> 8] = make_uint2(hdSlot, size);
            else if (size > 1) surv = 1;
        }
    }
    if (__ballot(surv != 0) != 0 && (tid & 63) == 0) v.counters[0] = 1;
}

// One workgroup refines one group of 257..ROWS*THREADS members (descriptors of other sizes are left to the other
// instantiations of the kernel): keys and positions into LDS, sort, subgroup boundaries, SA / ISA / bit map / children.
// A group in which one key holds the majority (periodic stretches and runs: every member but the ones near the end of the
// stretch looks at the same group h further on) is first split, stably, into "that key" and "the others"; only the
// others are sorted.
// (Round 6 measured the keys fetched inside this kernel, versioned labels as in k_bwt_f_small_fused: gather + sort of the medium groups
// 9.1 -> 11.0 ms on the real files, 5.5 -> 6.9 on the stand-in -- a sorting workgroup waits for its own three dependent loads, the gather kernel
// keeps eight groups per CU in flight and walks the list XCD by XCD. The keys stay a kernel of their own.)
template <int THREADS, int ROWS>
__global__ __launch_bounds__(THREADS, (THREADS >= 1024 ? 4 : THREADS / 128)) void k_bwt_f_sort_medium(FwdView v, const uint2* __restrict__ desc, u32 nDesc, int npass, u32 minLen,
                                                               uint2* __restrict__ medNext, uint2* __restrict__ largeNext, const uint2* __restrict__ descInfo,
                                                               uint4* __restrict__ superList)
{
    constexpr int WAVES = THREADS / 64;
    constexpr u32 CAP = (u32)ROWS * THREADS;
    __shared__ MedLds<THREADS, ROWS> L;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    for (u32 g = blockIdx.x; g < nDesc; g += gridDim.x) {
        const uint2 d = desc[g];
        const u32 gs = d.x, n = d.y;
        if (n <= minLen || n > CAP) continue;               // uniform for the workgroup
        const uint2 info = descInfo[g];                     // (block base, the members' label)
        {
            // all loads of the group in flight before the first one is used
            u32 k[ROWS], p[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) { const u32 i = (u32)tid + (u32)r * THREADS; k[r] = 0; p[r] = 0; if (i < n) { k[r] = v.K[gs + i]; p[r] = v.SA[gs + i]; } }
#pragma unroll
            for (int r = 0; r < ROWS; r++) { const u32 i = (u32)tid + (u32)r * THREADS; if (i < n) { L.oK[i] = k[r]; L.oV[i] = p[r]; } }
        }
        __syncthreads();
        if (tid == 0) { L.oldLab = info.y; L.blkBase = info.x; }
        // majority candidate: the key two of three probes agree on, else the middle one
        const u32 ka = L.oK[n >> 2], kb = L.oK[n >> 1], kc = L.oK[(n >> 2) * 3];
        const u32 m = (ka == kc) ? ka : kb;
        u32 c = 0;
        for (u32 i = (u32)tid; i < n; i += THREADS) c += (L.oK[i] == m) ? 1u : 0u;
        c = med_block_sum(L, c);
        if (c < n && 2 * c >= n && n <= SUPER_CAP && superList != nullptr && m == info.y - info.x + 1u) {
            // the majority looks at the group itself (a periodic stretch whose period divides h): k_bwt_f_super finishes it in one round
            if (tid == 0) { const u32 at = atomicAdd(&v.counters[7], 1u); superList[at] = make_uint4(gs, n, info.x, info.y); }
            __syncthreads();
            continue;
        }
        if (c < n) {
            if (2 * c >= n) {
                // ---- stable split: [others (nOth)][members with key m (c)]
                const int R = (int)((n + THREADS - 1) / THREADS);
                const u32 waveBase = (u32)wave * (u32)R * 64u;
                u32 key[ROWS], val[ROWS], pos[ROWS];
                u32 runO = 0, runE = 0;
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    key[r] = m; val[r] = 0; pos[r] = 0xFFFFFFFFu;
                    if (r < R) {
                        const u32 idx = waveBase + (u32)r * 64u + (u32)lane;
                        const bool valid = idx < n;
                        if (valid) { key[r] = L.oK[idx]; val[r] = L.oV[idx]; }
                        const bool oth = valid && key[r] != m, same = valid && key[r] == m;
                        const unsigned long long bo = __ballot(oth), be = __ballot(same);
                        if (oth) pos[r] = runO + (u32)__popcll(bo & ltMask);
                        if (same) pos[r] = 0x80000000u | (runE + (u32)__popcll(be & ltMask));
                        runO += (u32)__popcll(bo);
                        runE += (u32)__popcll(be);
                    }
                }
                if (lane == 0) { L.wtot[wave] = runO; L.wtot2[wave] = runE; }
                __syncthreads();
                u32 baseO = 0, baseE = 0, nOth = 0;
                for (int w = 0; w < WAVES; w++) { if (w < wave) { baseO += L.wtot[w]; baseE += L.wtot2[w]; } nOth += L.wtot[w]; }
#pragma unroll
                for (int r = 0; r < ROWS; r++) {
                    if (r < R && pos[r] != 0xFFFFFFFFu) {
                        const u32 at = (pos[r] & 0x80000000u) ? (nOth + baseE + (pos[r] & 0x7FFFFFFFu)) : (baseO + pos[r]);
                        L.oK[at] = key[r];
                        L.oV[at] = val[r];
                    }
                }
                __syncthreads();
                if (nOth <= (u32)THREADS) {
                    // few others: rank by counting (stable: smaller keys, then equal keys that come earlier)
                    u32 kk = 0, vv = 0, at = 0;
                    if ((u32)tid < nOth) {
                        kk = L.oK[tid]; vv = L.oV[tid];
                        for (u32 j = 0; j < nOth; j++) { const u32 kj = L.oK[j]; at += (kj < kk || (kj == kk && j < (u32)tid)) ? 1u : 0u; }
                    }
                    __syncthreads();
                    if ((u32)tid < nOth) { L.oK[at] = kk; L.oV[at] = vv; }
                    __syncthreads();
                    // A group that keeps its majority and loses a handful of members (round after round, for a periodic stretch): the
                    // majority is written where it goes straight from the split order and keeps its label, each of the few others finds the
                    // members it stays with by looking left and right among the sorted others -- no rearranging of the whole group, no bit map
                    // over it, no scans (med_write_back does all that for the general case)
                    u32 lessQ = 0;
                    if ((u32)tid < nOth) lessQ = (L.oK[tid] < m) ? 1u : 0u;
                    lessQ = med_block_sum(L, lessQ);
                    if (tid == 0) { L.oldLab = info.y; L.blkBase = info.x; }
                    __syncthreads();
                    med_majority_write_back<THREADS, ROWS>(L, v, gs, n, nOth, c, lessQ, m);
                    __syncthreads();
                    continue;
                } else {
                    med_radix_sort<THREADS, ROWS>(L, nOth, npass);
                }
                u32 less = 0;
                for (u32 i = (u32)tid; i < nOth; i += THREADS) less += (L.oK[i] < m) ? 1u : 0u;
                less = med_block_sum(L, less);
                // ---- [others < m][key m][others > m]
#pragma unroll
                for (int k = 0; k < ROWS; k++) {
                    const u32 i = (u32)tid + (u32)k * THREADS;
                    if (i < n) { key[k] = L.oK[i]; val[k] = L.oV[i]; }
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < ROWS; k++) {
                    const u32 i = (u32)tid + (u32)k * THREADS;
                    if (i < n) {
                        const u32 at = (i < less) ? i : (i < nOth ? i + c : less + (i - nOth));
                        L.oK[at] = key[k];
                        L.oV[at] = val[k];
                    }
                }
                __syncthreads();
            } else {
                med_radix_sort<THREADS, ROWS>(L, n, npass);
            }
        }
        med_write_back<THREADS, ROWS>(L, v, gs, n, medNext, largeNext);
        __syncthreads();
    }
}

// A group whose majority key is the group's OWN label: for most members p the suffix p + h is a member too, i.e. the members form
// chains p, p + h, p + 2h, ... inside the group (a periodic stretch whose period divides h) that end in a member e whose key is another
// label. With c(p) = steps from p to the end e(p) of its chain, two members compare like this: all of them start with the group's
// h symbols B, so suffix p = B^(c+1) followed by what the key of e(p) stands for. The one with the shorter chain meets its foreign
// key while the other still shows B (label = the group's own): it is the smaller one when that key is below the own label,
// the larger one when above; equal chain lengths leave the two foreign keys to decide. One sort on
//     hi = key(e) < own ? c : 2 CAP - c,   lo = rank of key(e) among the chain ends
// therefore does what log2(longest chain) doubling rounds would do; ties share (c + 1) h symbols and the depth of the end's label
// (at least 2h altogether) and go on as ordinary groups. The members are in position order (every sort of the pipeline is stable),
// so p + h is found by binary search in the group itself and c, e come from pointer jumping in LDS.
__global__ __launch_bounds__(1024) void k_bwt_f_super(FwdView v, const uint4* __restrict__ superList, u32 h, int npass, uint2* __restrict__ medNext, uint2* __restrict__ largeNext)
{
    constexpr int THREADS = 1024, ROWS = 8;                // (126 KB of LDS: one workgroup per CU, so as many waves in it as a workgroup can have)
    constexpr u32 CAP = (u32)ROWS * THREADS;
    __shared__ MedLds<THREADS, ROWS> L;
    __shared__ u16 nxt[CAP];
    __shared__ u16 cn[CAP];
    __shared__ u16 trEnd[CAP];
    __shared__ u32 sEnds;
    __shared__ u32 sMoved[2];
    __shared__ u32 sCmax;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const unsigned long long ltMask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const u32 nList = v.counters[7];
    for (u32 g = blockIdx.x; g < nList; g += gridDim.x) {
        const uint4 d = superList[g];
        const u32 gs = d.x, n = d.y;                       // 256 < n <= CAP
        {
            u32 k[ROWS], p[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) { const u32 i = (u32)tid + (u32)r * THREADS; k[r] = 0; p[r] = 0; if (i < n) { k[r] = v.K[gs + i]; p[r] = v.SA[gs + i]; } }
#pragma unroll
            for (int r = 0; r < ROWS; r++) { const u32 i = (u32)tid + (u32)r * THREADS; if (i < n) { L.oK[i] = k[r]; L.oV[i] = p[r]; } }
        }
        if (tid == 0) { sEnds = 0; sCmax = 0; L.oldLab = d.w; L.blkBase = d.z; }
        __syncthreads();
        const u32 m = d.w - d.z + 1u;
        // the offset the group's keys were gathered at (k_bwt_f_gather_desc): h, or what the group is known to share beyond it
        u32 link = h;
        if (v.rtbits && ((v.rtbits[gs >> 5] >> (gs & 31)) & 1u)) { const u32 r = v.ovr[gs]; link = r > h ? r : h; }
        // ---- chain links: the member `link` further on
#pragma unroll 4
        for (int r = 0; r < ROWS; r++) {
            const u32 i = (u32)tid + (u32)r * THREADS;
            if (i >= n) continue;
            u32 to = i, one = 0;
            if (L.oK[i] == m) {
                const u32 target = L.oV[i] + link;
                u32 lo = i + 1, hi = n;                     // first index with position >= target
                while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (L.oV[mid] < target) lo = mid + 1; else hi = mid; }
                if (lo < n && L.oV[lo] == target) { to = lo; one = 1; }
            }
            nxt[i] = (u16)to; cn[i] = (u16)one;
        }
        __syncthreads();
        // ---- pointer jumping: chain end and chain length of every member (until no link moves any more)
        for (u32 span = 1, it = 0; span < n; span <<= 1, it ^= 1) {
            u32 a[ROWS], b[ROWS];
            bool moved = false;
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const u32 i = (u32)tid + (u32)r * THREADS;
                a[r] = 0; b[r] = 0;
                if (i < n) { const u32 j = nxt[i]; a[r] = nxt[j]; b[r] = (u32)cn[i] + (u32)cn[j]; moved = moved || (a[r] != j); }
            }
            if (tid == 0) sMoved[it] = 0;                          // (two flags in turn: the other one may still be read by a slow thread)
            __syncthreads();
            if (moved) sMoved[it] = 1;
#pragma unroll
            for (int r = 0; r < ROWS; r++) { const u32 i = (u32)tid + (u32)r * THREADS; if (i < n) { nxt[i] = (u16)a[r]; cn[i] = (u16)b[r]; } }
            __syncthreads();
            if (!sMoved[it]) break;
        }
        u32 e[ROWS], cc[ROWS], tail[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const u32 i = (u32)tid + (u32)r * THREADS;
            e[r] = 0; cc[r] = 0; tail[r] = 0;
            if (i < n) { e[r] = nxt[i]; cc[r] = cn[i]; tail[r] = L.oK[e[r]]; }
        }
        __syncthreads();
        // ---- the chain ends, compacted (any order) and sorted by key
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const u32 i = (u32)tid + (u32)r * THREADS;
            const bool isEnd = (i < n) && cc[r] == 0;
            const unsigned long long bal = __ballot(isEnd);
            u32 base0 = 0;
            if (lane == 0 && bal) base0 = atomicAdd(&sEnds, (u32)__popcll(bal));
            base0 = (u32)__shfl((int)base0, 0, 64);
            if (isEnd) { const u32 at = base0 + (u32)__popcll(bal & ltMask); L.oK[at] = tail[r]; L.oV[at] = i; }
        }
        __syncthreads();
        const u32 nEnds = sEnds;
        if (nEnds <= (u32)THREADS) {
            u32 kk = 0, vv = 0, at = 0;
            if ((u32)tid < nEnds) {
                kk = L.oK[tid]; vv = L.oV[tid];
                for (u32 j = 0; j < nEnds; j++) { const u32 kj = L.oK[j]; at += (kj < kk || (kj == kk && j < (u32)tid)) ? 1u : 0u; }
            }
            __syncthreads();
            if ((u32)tid < nEnds) { L.oK[at] = kk; L.oV[at] = vv; }
            __syncthreads();
        } else {
            med_radix_sort<THREADS, ROWS>(L, nEnds, npass);
        }
        // rank of an end's key = first index with that key
        for (u32 j = (u32)tid; j < nEnds; j += THREADS) {
            const u32 key = L.oK[j];
            u32 lo = 0, hi = j;
            while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (L.oK[mid] < key) lo = mid + 1; else hi = mid; }
            trEnd[L.oV[j]] = (u16)lo;
        }
        __syncthreads();
        // ---- one sort of all members on (hi, rank of the end's key); the key is as wide as the longest chain and the number of ends ask for
        {
            u32 cm = 0;
#pragma unroll
            for (int r = 0; r < ROWS; r++) cm = cc[r] > cm ? cc[r] : cm;
            cm = wave_max(cm);
            if (lane == 0 && cm) atomicMax(&sCmax, cm);
        }
        __syncthreads();
        const u32 cmax = sCmax;
        const u32 trBits = bitlen_u32(nEnds ? nEnds - 1 : 0), hiBits = bitlen_u32(2u * cmax + 1u);
        u32 fk[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const u32 i = (u32)tid + (u32)r * THREADS;
            fk[r] = 0;
            if (i < n) { const u32 hiKey = (tail[r] < m) ? cc[r] : (2u * cmax + 1u - cc[r]); fk[r] = (hiKey << trBits) | (u32)trEnd[e[r]]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ROWS; r++) { const u32 i = (u32)tid + (u32)r * THREADS; if (i < n) { L.oK[i] = fk[r]; L.oV[i] = i; } }
        __syncthreads();
        med_radix_sort<THREADS, ROWS>(L, n, (int)((hiBits + trBits + 7) / 8));
        {
            u32 p[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; r++) { const u32 i = (u32)tid + (u32)r * THREADS; p[r] = 0; if (i < n) p[r] = v.SA[gs + L.oV[i]]; }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ROWS; r++) { const u32 i = (u32)tid + (u32)r * THREADS; if (i < n) L.oV[i] = p[r]; }
        }
        __syncthreads();
        med_write_back<THREADS, ROWS>(L, v, gs, n, medNext, largeNext);
        __syncthreads();
    }
}

constexpr u32 PROBE_PMAX = 2048;
// Periodic stretches of ANY period, found by looking at the text (once, in front of the first doubling round). The members of a medium group are
// in position order; when most neighbours are the same distance P apart the group is mostly made of chains p, p + P, p + 2P, ...
// through stretches of period P, and doubling would refine it log2(stretch / h) times, peeling off only the members near the ends of
// the stretches each time, before the chain round can see the group look at itself (which it only does once P divides h). Here every
// member is compared with ONE member's text over the P symbols behind the ones the group shares: members that differ leave, ordered by
// where they differ and how (an ordinary refinement: a member that falls below the pattern earlier is smaller, one that rises above it
// earlier is larger, same place and byte stay together); the members that equal the pattern stay one group that is now KNOWN to share P
// symbols, which is recorded like a run length (FwdView::ovr / rtbits): its keys are gathered P positions on, where most members find
// the group itself, and the chain round finishes it in the first doubling round whatever P is (a block of text(2000) + 550 copies of a
// random 700-byte unit: 16 doubling rounds without this, 1 with it).
// candidates: medium groups with three equal distances around their middle member, a distance that is no power of two
// (a period that is a power of two is left to the doubling rounds: the chain round takes the group as soon as h reaches it, and the rounds up
// to there are cheap -- the majority of the members looks at one and the same group and is split off unsorted; measured on the stand-in's
// period-256 stretches the text comparison costs 3.4 ms and saves 1.7. Any other period never meets h.)
__global__ __launch_bounds__(256) void k_bwt_f_probe_scan(FwdView v, const uint2* __restrict__ desc, u32 depth, const u32* __restrict__ rtbits, u32* __restrict__ cand)
{
    const u32 nDesc = v.counters[1];
    for (u32 g = blockIdx.x * 256 + threadIdx.x; g < nDesc; g += gridDim.x * 256) {
        const uint2 d = desc[g];
        const u32 gs = d.x, n = d.y;
        if (n <= SM_G || n > SUPER_CAP) continue;
        if (rtbits != nullptr && ((rtbits[gs >> 5] >> (gs & 31)) & 1u)) continue;         // (groups of the run round know their offset already)
        const u32 a0 = v.SA[gs + (n >> 1) - 2], a1 = v.SA[gs + (n >> 1) - 1], a2 = v.SA[gs + (n >> 1)], a3 = v.SA[gs + (n >> 1) + 1];
        const u32 pd = a2 - a1;
        if ((a1 - a0 == pd) && (a3 - a2 == pd) && (pd > depth) && (pd <= 2048u) && ((pd & (pd - 1)) != 0)) cand[atomicAdd(&v.counters[14], 1u)] = g;
    }
}

__global__ __launch_bounds__(512, 4) void k_bwt_f_probe(BwtView bv, FwdView v, uint2* __restrict__ desc, const u32* __restrict__ cand, u32 nCand, u32 depth, uint2* __restrict__ medNext,
                                                        uint2* __restrict__ largeNext, u32* __restrict__ ovr, u32* __restrict__ rtbits)
{
    constexpr int THREADS = 512, ROWS = 16;
    constexpr u32 CAP = (u32)ROWS * THREADS;
    __shared__ MedLds<THREADS, ROWS> L;
    __shared__ u32 patw[PROBE_PMAX / 4];
    u8* pat = reinterpret_cast<u8*>(patw);
    __shared__ u32 sInfo[4];                 // block base, block end, label of the group, -
    const int tid = (int)threadIdx.x;
    for (u32 cg = blockIdx.x; cg < nCand; cg += gridDim.x) {
        const u32 g = cand[cg];
        const uint2 d = desc[g];
        const u32 gs = d.x, n = d.y;
        bool take = n > SM_G && n <= CAP;
        __syncthreads();                                     // (LDS of the group before)
        if (take) {
            for (u32 i = (u32)tid; i < n; i += THREADS) L.oV[i] = v.SA[gs + i];
            if (tid == 0) {
                const int b = find_block(v.base, v.nBlocks, gs);
                sInfo[0] = v.base[b]; sInfo[1] = v.base[b + 1]; sInfo[3] = (u32)b;
            }
        }
        __syncthreads();
        u32 P = 0, pr = 0;
        if (take) {
            pr = L.oV[n >> 1];
            P = pr - L.oV[(n >> 1) - 1];
            take = P > depth && P <= PROBE_PMAX && pr + P <= sInfo[1];
        }
        if (take) {
            u32 c = 0;
            for (u32 i = (u32)tid; i + 1 < n; i += THREADS) c += (L.oV[i + 1] - L.oV[i] == P) ? 1u : 0u;
            c = med_block_sum(L, c);
            take = 2 * c >= n;
        }
        if (!take) continue;                                  // (uniform) the group goes on as it is
        const u32 bb = sInfo[0], be = sInfo[1];
        const u8* t = bv.src[sInfo[3]];
        for (u32 j = (u32)tid; j < P; j += THREADS) pat[j] = ldg<u8>(t + (pr - bb + j));
        if (tid == 0) sInfo[2] = lab_cur(v, pr, sInfo[0]);
        __syncthreads();
        // ---- every member against the pattern: where and how it differs
        constexpr u32 MID = 1u << 21;
        u32 nEq = 0, nBelow = 0;
        for (u32 i = (u32)tid; i < n; i += THREADS) {
            const u32 q0 = L.oV[i] - bb;
            u32 l = depth;
            // sixteen bytes at a time while they agree: the member's from five aligned dwords (all loads issued together), the pattern's
            // as dwords from LDS (`depth` and the steps are multiples of four)
            if ((depth & 3u) == 0) {
                const u32* pw = patw;
                while (l + 16 <= P && q0 + l + 20 <= be - bb) {
                    const uintptr_t ad = reinterpret_cast<uintptr_t>(t + q0 + l);
                    const u8* w = reinterpret_cast<const u8*>(ad & ~(uintptr_t)3);
                    const u32 sh = (u32)(ad & 3) * 8;
                    const u32 w0 = ldg<u32>(w), w1 = ldg<u32>(w + 4), w2 = ldg<u32>(w + 8), w3 = ldg<u32>(w + 12), w4 = ldg<u32>(w + 16);
                    const u32 x0 = sh ? ((w0 >> sh) | (w1 << (32 - sh))) : w0, x1 = sh ? ((w1 >> sh) | (w2 << (32 - sh))) : w1;
                    const u32 x2 = sh ? ((w2 >> sh) | (w3 << (32 - sh))) : w2, x3 = sh ? ((w3 >> sh) | (w4 << (32 - sh))) : w3;
                    const u32 d0 = x0 ^ pw[l >> 2], d1 = x1 ^ pw[(l >> 2) + 1], d2 = x2 ^ pw[(l >> 2) + 2], d3 = x3 ^ pw[(l >> 2) + 3];
                    if ((d0 | d1 | d2 | d3) != 0) break;
                    l += 16;
                }
            }
            u32 code = MID;
            while (l < P) {
                if (q0 + l >= be - bb) { code = l * 512u; break; }                        // the suffix ends here: a proper prefix sorts first
                const u32 x = ldg<u8>(t + q0 + l), y = pat[l];
                if (x != y) { code = (x < y) ? (l * 512u + x + 1u) : (MID + 1u + (P - l) * 512u + x + 1u); break; }
                l++;
            }
            L.oK[i] = code;
            nEq += (code == MID) ? 1u : 0u;
            nBelow += (code < MID) ? 1u : 0u;
        }
        nEq = med_block_sum(L, nEq);
        nBelow = med_block_sum(L, nBelow);
        if (2 * nEq < n) continue;                            // the pattern was not the stretches' (or there are none): nothing was written
        // the parts of the group are staged for the next round's list (med_write_back); this round's descriptor is void
        if (tid == 0) { L.oldLab = sInfo[2]; L.blkBase = sInfo[0]; desc[g].y = 0; }
        med_radix_sort<THREADS, ROWS>(L, n, 3);
        med_write_back<THREADS, ROWS>(L, v, gs, n, medNext, largeNext);
        // what every member is now known to share with the members it stays together with: the ones that equal the pattern P symbols
        // (slots [gs + nBelow, + nEq)), a member that left at symbol l with the ones that left there with the same byte l + 1
        for (u32 i = (u32)tid; i < n; i += THREADS) {
            const u32 code = L.oK[i];
            const u32 lshare = (code == MID) ? P : ((code < MID) ? (code >> 9) : (P - ((code - MID - 1u) >> 9))) + 1u;
            ovr[gs + i] = lshare;
            atomicOr(&rtbits[(gs + i) >> 5], 1u << ((gs + i) & 31));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// large groups: one global sort of (descriptor index, key) pairs
// ------------------------------------------------------------------------------------------------
// loff[i] = sum of the lengths of the descriptors before i; loff[n] = total
// (+ lbase[i] = first slot of the block descriptor i lies in: the placing kernels store labels relative to it, lab_set)
__global__ __launch_bounds__(1024) void k_bwt_f_large_prefix(const uint2* __restrict__ desc, u32 nDesc, u32* __restrict__ loff, const u32* __restrict__ base, int nBlocks,
                                                             u32* __restrict__ lbase)
{
    __shared__ u32 part[1024];
    const u32 per = (nDesc + 1023) / 1024;
    const u32 lo = threadIdx.x * per, hi = (lo + per < nDesc) ? lo + per : nDesc;
    u32 sum = 0;
    for (u32 i = lo; i < hi; i++) { sum += desc[i].y; lbase[i] = base[find_block(base, nBlocks, desc[i].x)]; }
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) { u32 run = 0; for (int t = 0; t < 1024; t++) { const u32 x = part[t]; part[t] = run; run += x; } loff[nDesc] = run; }
    __syncthreads();
    u32 run = part[threadIdx.x];
    for (u32 i = lo; i < hi; i++) { loff[i] = run; run += desc[i].y; }
}

// KEY = u32 when descriptor index and key fit 32 bits together (two thirds of the sort's traffic), else u64
template <class KEY>
__global__ __launch_bounds__(256) void k_bwt_f_large_keys(FwdView v, const uint2* __restrict__ desc, u32 nDesc, const u32* __restrict__ loff,
                                                          u32 L, u32 h, int kbits, KEY* __restrict__ keys, u32* __restrict__ vals)
{
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= L) return;
    u32 lo = 0, hi = nDesc;
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (loff[mid] <= j) lo = mid; else hi = mid; }
    const uint2 d = desc[lo];
    const u32 slot = d.x + (j - loff[lo]);
    const int b = find_block(v.base, v.nBlocks, d.x);
    const u32 gp = v.SA[slot];
    u32 off = h;
    if (v.rtbits && ((v.rtbits[d.x >> 5] >> (d.x & 31)) & 1u)) { const u32 r = v.ovr[d.x]; off = r > h ? r : h; }
    const u32 key = gather_key(v, gp, off, v.base[b], v.base[b + 1]);
    keys[j] = (KEY)(((u64)lo << kbits) | (u64)key);
    vals[j] = gp;
}

template <class KEY>
__global__ __launch_bounds__(256) void k_bwt_f_large_flags(const KEY* __restrict__ keys, u32 L, u32* __restrict__ headIdx, u32* __restrict__ nextIdxRev)
{
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= L) return;
    const bool f = (j == 0) || (keys[j] != keys[j - 1]);
    headIdx[j] = f ? j : 0u;
    nextIdxRev[L - 1 - j] = f ? j : L;
}

template <class KEY>
__global__ __launch_bounds__(256) void k_bwt_f_large_place(FwdView v, const uint2* __restrict__ desc, const u32* __restrict__ loff, u32 L, int kbits,
                                                           const KEY* __restrict__ keys, const u32* __restrict__ vals, const u32* __restrict__ head,
                                                           const u32* __restrict__ nextRev, uint2* __restrict__ medNext, uint2* __restrict__ largeNext,
                                                           const u32* __restrict__ memberR, u32* __restrict__ ovr, u32* __restrict__ rtbits, const u32* __restrict__ lbase)
{
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    u32 surv = 0;
    u32 mySlot = 0;
    bool setBit = false;
    __shared__ ClassAgg A;
    agg_init(A);
    __syncthreads();
    int hKind = -1;
    u32 hLocal = 0, hSize = 0;
    if (j < L) {
        const u32 di = (u32)(keys[j] >> kbits);
        const u32 gs = desc[di].x, off = loff[di];
        const u32 gp = vals[j];
        const u32 nh = head[j];
        v.SA[gs + (j - off)] = gp;
        // (a large group's label is its first slot: round 0 and this kernel are the only ones that make large groups)
        if (nh != off) lab_set(v, gp, lbase[di], gs + (nh - off), gs);
        mySlot = gs + (j - off);
        if (memberR != nullptr) ovr[mySlot] = memberR[j];          // the run round: what the doubling rounds need to look behind the run
        if (nh == j) {
            const u32 nxt = (j + 1 < L) ? nextRev[L - 2 - j] : L;
            setBit = (j != off);
            hSize = nxt - j;
            hKind = agg_note(A, hSize, false, surv, hLocal);
        }
    }
    if (__ballot(surv != 0) != 0 && (threadIdx.x & 63) == 0) v.counters[0] = 1;
    __syncthreads();
    if (threadIdx.x == 0) agg_reserve(A, v);
    __syncthreads();
    if (hKind >= 0) agg_write(A, v, hKind, hLocal, mySlot, hSize, largeNext, (uint2*)nullptr);
    // new group starts: the lanes of a wave mostly hold consecutive slots (one group), so their bits are put together with a
    // ballot and leave as at most three word-wide ORs per wave; lanes that are out of line (a group border inside the wave) OR
    // their own bit
    const int lane = (int)(threadIdx.x & 63);
    const u32 s0 = (u32)__shfl((int)mySlot, 0, 64);
    const bool inLine = (j < L) && (mySlot == s0 + (u32)lane);
    if (rtbits != nullptr) {
        const unsigned long long rm = __ballot(inLine);
        if (j < L && !inLine) atomicOr(&rtbits[mySlot >> 5], 1u << (mySlot & 31));
        if (rm != 0 && lane == 0) {
            const u32 w0 = s0 >> 5, sh = s0 & 31;
            const u32 p0 = (u32)(rm << sh);
            const u32 p1 = sh ? (u32)(rm >> (32 - sh)) : (u32)(rm >> 32);
            const u32 p2 = sh ? (u32)(rm >> (64 - sh)) : 0u;
            if (p0) atomicOr(&rtbits[w0], p0);
            if (p1) atomicOr(&rtbits[w0 + 1], p1);
            if (p2) atomicOr(&rtbits[w0 + 2], p2);
        }
    }
    const unsigned long long mask = __ballot(setBit && inLine);
    if (setBit && !inLine) atomicOr(&v.gnew[mySlot >> 5], 1u << (mySlot & 31));
    if (mask != 0 && lane == 0) {
        const u32 w0 = s0 >> 5, sh = s0 & 31;
        const u32 p0 = (u32)(mask << sh);
        const u32 p1 = sh ? (u32)(mask >> (32 - sh)) : (u32)(mask >> 32);
        const u32 p2 = sh ? (u32)(mask >> (64 - sh)) : 0u;
        if (p0) atomicOr(&v.gnew[w0], p0);
        if (p1) atomicOr(&v.gnew[w0 + 1], p1);
        if (p2) atomicOr(&v.gnew[w0 + 2], p2);
    }
}

// ------------------------------------------------------------------------------------------------
// runs of one byte: finished in one round by their length
// ------------------------------------------------------------------------------------------------
// Members of a run group all start with nsym copies of a byte c. With R = length of the run of c that starts at the position, a =
// the byte behind the run (or the block end, which sorts first), two members compare like (R, a): the one whose run ends first
// is smaller when its a is below c, larger when above; equal (R, a) leave the suffixes behind the runs to decide. So ONE sort on
//   hi = (a < c) ? R : 2^(kbits+1) - 1 - R,   lo = ISA[p + R] (rank of the suffix behind the run, 0 at the block end)
// replaces the log2(R / nsym) doubling rounds such a group would otherwise take; members that still tie share R + nsym symbols.

// run-end flags in position order: bit set where the next byte differs or the block ends (and for everything past the end);
// a thread makes the flag byte of 8 positions, a workgroup the 2048 positions of one window
__global__ __launch_bounds__(256) void k_bwt_f_run_ends(BwtView bv, FwdView v, u8* __restrict__ ebits8)
{
    __shared__ int sBlk;
    const u32 pos0 = blockIdx.x * SM_WIN;
    if (threadIdx.x == 0) sBlk = find_block(v.base, v.nBlocks, pos0 < v.total ? pos0 : v.total - 1);
    __syncthreads();
    const u32 g0 = pos0 + 8u * threadIdx.x;
    u32 flags = 0xFF;
    if (g0 < v.total) {
        int b = sBlk;
        while (g0 >= v.base[b + 1]) b++;
        u32 bb = v.base[b], be = v.base[b + 1];
        const u8* t = bv.src[b];
        u32 cur = t[g0 - bb];
        flags = 0;
#pragma unroll
        for (u32 k = 0; k < 8; k++) {
            const u32 gp = g0 + k;
            if (gp >= v.total) { flags |= 1u << k; continue; }
            if (gp + 1 >= be) {                               // last byte of a block: the run ends here
                flags |= 1u << k;
                if (gp + 1 < v.total) { b++; while (v.base[b + 1] <= gp + 1) b++; bb = be; be = v.base[b + 1]; t = bv.src[b]; cur = t[0]; }   // (blocks the BWT skips are empty here)
                continue;
            }
            const u32 nxt = t[gp + 1 - bb];
            flags |= (cur != nxt ? 1u : 0u) << k;
            cur = nxt;
        }
    }
    ebits8[g0 >> 3] = (u8)flags;
}

// R[gp] = distance to the end of the run that holds gp (1 = the run ends here); one workgroup per window of 2048 positions
__global__ __launch_bounds__(256) void k_bwt_f_run_len(BwtView bv, FwdView v, const u32* __restrict__ ebits, const u32* __restrict__ winFirstInclRev, u32 nWin,
                                                       u32* __restrict__ R, const u32* __restrict__ classTab, u32 nsym, unsigned long long* __restrict__ startBits64,
                                                       u32* __restrict__ startCount)
{
    __shared__ u32 bw[64];
    __shared__ u32 nextSet[64];
    __shared__ int sBlk;
    const int tid = (int)threadIdx.x;
    const u32 win = blockIdx.x, pos0 = win * SM_WIN;
    if (tid == 64) sBlk = find_block(v.base, v.nBlocks, pos0 < v.total ? pos0 : v.total - 1);
    if (tid < 64) {
        const u32 w = ebits[(pos0 >> 5) + (u32)tid];
        bw[tid] = w;
        u32 sm = w ? (u32)(tid * 32 + __ffs((int)w) - 1) : NO_BIT;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 u = (u32)__shfl_down((int)sm, (unsigned)o, 64);
            if (tid + o < 64) sm = u < sm ? u : sm;
        }
        u32 sex = (u32)__shfl_down((int)sm, 1u, 64);
        if (tid == 63) sex = NO_BIT;
        nextSet[tid] = sex;
    }
    __syncthreads();
    const u32 after = (win + 1 < nWin) ? winFirstInclRev[nWin - 2 - win] : v.total - 1;
    const u32 prevWord = pos0 ? ebits[(pos0 >> 5) - 1] : 0xFFFFFFFFu;      // (the run-end bit of the position in front of the window)
    u32 rmax = 0;
#pragma unroll
    for (int k = 0; k < (int)(SM_WIN / 256); k++) {
        const u32 i = (u32)tid + 256u * (u32)k;
        const u32 gp = pos0 + i;
        bool start = false;
        if (gp < v.total) {
            const u32 w = i >> 5, bit = i & 31;
            const u32 m = bw[w] & (0xFFFFFFFFu << bit);              // this position or later
            const u32 ei = m ? (w * 32 + (u32)__ffs((int)m) - 1) : nextSet[w];
            const u32 e = (ei != NO_BIT) ? pos0 + ei : after;
            const u32 r = e - gp + 1;
            R[gp] = r;
            rmax = r > rmax ? r : rmax;
            // a run starts here when a run ends right in front (block ends are run ends); it takes part in the run-length round when it
            // is at least nsym long and its byte has a run group
            const bool endBefore = (i == 0) ? ((prevWord >> 31) & 1u) != 0 : ((bw[(i - 1) >> 5] >> ((i - 1) & 31)) & 1u) != 0;
            if (endBefore && r >= nsym && classTab != nullptr) {
                int b = sBlk;
                while (gp >= v.base[b + 1]) b++;
                start = classTab[(u32)b * 256u + (u32)bv.src[b][gp - v.base[b]]] != 0xFFFFFFFFu;
            }
        }
        const unsigned long long m64 = __ballot(start);
        if ((tid & 63) == 0 && startBits64 != nullptr) {
            startBits64[(pos0 + 256u * (u32)k + (u32)(tid & ~63)) >> 6] = m64;
            startCount[((pos0 + 256u * (u32)k + (u32)(tid & ~63)) >> 5)] = (u32)__popc((u32)m64);
            startCount[((pos0 + 256u * (u32)k + (u32)(tid & ~63)) >> 5) + 1] = (u32)__popc((u32)(m64 >> 32));
        }
    }
    rmax = wave_max(rmax);
    // longest run: sizes the key of the run-length sort (read first: one atomic per wave on one address would serialise the launch)
    if ((tid & 63) == 0 && rmax > __atomic_load_n(&v.counters[6], __ATOMIC_RELAXED)) atomicMax(&v.counters[6], rmax);
}

// the run groups as ordinary groups (when the run-length round cannot take them)
__global__ __launch_bounds__(256) void k_bwt_f_run_fallback(FwdView v, const uint2* __restrict__ runList, u32 nRun, uint2* __restrict__ medNext, uint2* __restrict__ largeNext)
{
    const u32 g = blockIdx.x * 256 + threadIdx.x;
    u32 surv = 0;
    if (g < nRun) classify_child(v, medNext, largeNext, runList[g].x, runList[g].y, surv);
}

// ---- the run-length round without a sort of (key, position) pairs over 48 bits ----------------------------------------------
// A member's key (run length R, what follows the run) is a property of its RUN except for R itself: a run of length L of a
// run-group byte has the members R = nsym .. L, and all of them share the direction bit and the rank T of the suffix behind the
// run. So the runs (a few per cent of the members in number) are sorted on (class, direction, T) first; the members are then
// generated run by run in that order and sorted, stably, on (class, hi(R)) alone -- three passes over 8-byte elements for 24 key
// bits, where the pair sort took six passes over 12-byte elements: the order of equal (class, hi) is the order of the runs, i.e.
// T. What k_bwt_f_large_flags / _place want -- the full key and the position per slot -- is rebuilt from the run table.

// class of a run-group byte: classTab[block * 256 + byte] = index of the group's descriptor
__global__ __launch_bounds__(256) void k_bwt_f_run_classes(BwtView bv, FwdView v, const uint2* __restrict__ runList, u32 nRun, u32* __restrict__ classTab)
{
    const u32 d = blockIdx.x * 256 + threadIdx.x;
    if (d >= nRun) return;
    const u32 gp = v.SA[runList[d].x];
    const int b = find_block(v.base, v.nBlocks, gp);
    classTab[(u32)b * 256u + (u32)bv.src[b][gp - v.base[b]]] = d;
}

__global__ __launch_bounds__(256) void k_bwt_f_run_compact(const u32* __restrict__ bits, const u32* __restrict__ wprefix, u32 nWords, u32* __restrict__ runPos)
{
    const u32 w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nWords) return;
    u32 word = bits[w];
    u32 at = wprefix[w];
    while (word) {
        const u32 k = (u32)__ffs((int)word) - 1;
        runPos[at++] = 32u * w + k;
        word &= word - 1;
    }
}

// run k: end, length, and the sort key [class | above? | T | k]
__global__ __launch_bounds__(256) void k_bwt_f_run_table(BwtView bv, FwdView v, const u32* __restrict__ runPos, u32 nRuns, const u32* __restrict__ R,
                                                         const u32* __restrict__ classTab, int kbits, int idxBits, u64* __restrict__ runKeys,
                                                         u32* __restrict__ runE, u32* __restrict__ runL)
{
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nRuns) return;
    const u32 p = runPos[k];
    const int b = find_block(v.base, v.nBlocks, p);
    const u32 bb = v.base[b], be = v.base[b + 1];
    const u8* t = bv.src[b];
    const u32 c = t[p - bb], L = R[p], e = p + L;
    const bool below = (e >= be) || (t[e - bb] < c);                 // the block end sorts in front of every byte
    const u64 T = (e < be) ? (u64)(lab_cur(v, e, bb) - bb + 1u) : 0ull;
    const u64 cls = classTab[(u32)b * 256u + c];
    runKeys[k] = ((((cls << 1) | (below ? 0ull : 1ull)) << kbits | T) << idxBits) | (u64)k;
    runE[k] = e;
    runL[k] = L;
}

// the runs in sorted order: end, (class, direction, T), number of members
__global__ __launch_bounds__(256) void k_bwt_f_run_sorted(const u64* __restrict__ runKeys, u32 nRuns, int idxBits, const u32* __restrict__ runE,
                                                          const u32* __restrict__ runL, u32 nsym, u32* __restrict__ sE, u64* __restrict__ sKey, u32* __restrict__ cnt)
{
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nRuns) return;
    const u64 key = runKeys[k];
    const u32 idx = (u32)(key & ((1ull << idxBits) - 1ull));
    sE[k] = runE[idx];
    sKey[k] = key >> idxBits;
    cnt[k] = runL[idx] - nsym + 1u;
}

// member j (members of run 0 first, then of run 1, ...): [class | hi(R) | run index]; a workgroup makes 2048 consecutive members
__global__ __launch_bounds__(256) void k_bwt_f_run_members(const u32* __restrict__ moff, u32 nRuns, u32 M, const u64* __restrict__ sKey, int kbits, int hbits,
                                                           u32 nsym, u64* __restrict__ mkeys)
{
    __shared__ u32 sOff[2048 + 1];
    __shared__ u32 sFirst;
    const u32 j0 = blockIdx.x * 2048u;
    if (j0 >= M) return;
    if (threadIdx.x == 0) {
        u32 lo = 0, hi = nRuns;                                      // last run with moff <= j0
        while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (moff[mid] <= j0) lo = mid; else hi = mid; }
        sFirst = lo;
    }
    __syncthreads();
    const u32 k0 = sFirst;
    // every run has at least one member: the members of this tile belong to at most 2048 runs
    for (u32 i = threadIdx.x; i <= 2048u; i += 256) sOff[i] = (k0 + i < nRuns) ? moff[k0 + i] : 0xFFFFFFFFu;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const u32 j = j0 + (u32)q * 256u + threadIdx.x;
        if (j >= M) continue;
        u32 lo = 0, hi = 2048;
        while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (sOff[mid] <= j) lo = mid; else hi = mid; }
        if (sOff[hi] <= j) lo = hi;
        const u32 k = k0 + lo;
        const u64 key = sKey[k];
        const u64 cls = key >> (kbits + 1);
        const bool above = (key >> kbits) & 1ull;
        const u64 Rr = (u64)nsym + (u64)(j - sOff[lo]);
        const u64 hiK = above ? ((1ull << hbits) - 1ull - Rr) : Rr;
        mkeys[j] = (((cls << hbits) | hiK) << 32) | (u64)k;
    }
}

// sorted members -> the (key, position) pairs k_bwt_f_large_flags / k_bwt_f_large_place work on: key = [class | hi | T]
__global__ __launch_bounds__(256) void k_bwt_f_run_expand(const u64* __restrict__ mkeys, u32 M, const u64* __restrict__ sKey, const u32* __restrict__ sE,
                                                          int kbits, int hbits, u64* __restrict__ keys, u32* __restrict__ vals, u32* __restrict__ Rout)
{
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const u64 mk = mkeys[j];
    const u32 k = (u32)mk;
    const u64 ch = mk >> 32;                                         // class | hi
    const u64 hiK = ch & ((1ull << hbits) - 1ull);
    const u64 rk = sKey[k];
    const bool above = (rk >> kbits) & 1ull;
    const u64 T = rk & ((1ull << kbits) - 1ull);
    const u32 Rr = (u32)(above ? ((1ull << hbits) - 1ull - hiK) : hiK);
    keys[j] = (ch << kbits) | T;
    vals[j] = sE[k] - Rr;
    if (Rout != nullptr) Rout[j] = Rr;
}

// ---- the run round's last step without index arrays: where a tie starts among the sorted members is ONE BIT per member (a ballot per
// wave), the windows of 2048 members find their heads and sizes the way round 0 does (k_bwt_f_r0_winsum + two scans over the windows),
// and the placing kernel reads keys, positions and 64 words of bits -- where k_bwt_f_large_flags wrote two index arrays of the members'
// size and two device-wide scans ran over them.
__global__ __launch_bounds__(256) void k_bwt_f_run_flags(const u64* __restrict__ keys, u32 M, unsigned long long* __restrict__ bits64)
{
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    const u64 k = (j < M) ? keys[j] : 0ull;
    u32 klo = (u32)__shfl_up((int)(u32)k, 1u, 64), khi = (u32)__shfl_up((int)(u32)(k >> 32), 1u, 64);
    u64 kp = ((u64)khi << 32) | klo;
    if ((threadIdx.x & 63) == 0 && j > 0 && j < M) kp = keys[j - 1];
    const bool f = (j >= M) || (j == 0) || (k != kp);                  // (every bit from M on is set: the end of the last tie)
    const unsigned long long m = __ballot(f);
    if ((threadIdx.x & 63) == 0) bits64[j >> 6] = m;
}

__global__ __launch_bounds__(256) void k_bwt_f_run_place(FwdView v, const uint2* __restrict__ desc, const u32* __restrict__ loff, u32 M, int kbits, const u64* __restrict__ keys,
                                                         const u32* __restrict__ vals, const u32* __restrict__ bits, const u32* __restrict__ winLastIncl,
                                                         const u32* __restrict__ winFirstInclRev, u32 nWin, uint2* __restrict__ largeNext, const u32* __restrict__ memberR,
                                                         u32* __restrict__ ovr, u32* __restrict__ rtbits, const u32* __restrict__ lbase)
{
    // one member per thread (the kernel is a scatter of labels: occupancy is what hides its latency); the 64 words of the member's window of
    // 2048 are looked at by every one of the window's eight workgroups
    __shared__ SmWindow W;
    __shared__ ClassAgg A;
    const int tid = (int)threadIdx.x;
    const u32 jb = blockIdx.x * 256u;
    const u32 win = jb / SM_WIN, j0 = win * SM_WIN;
    agg_init(A);
    if (tid < 64) {
        const u32 w = bits[(j0 >> 5) + (u32)tid];
        W.bw[tid] = w;
        int pm = w ? (tid * 32 + 31 - __clz((int)w)) : -1;
        u32 sm = w ? (u32)(tid * 32 + __ffs((int)w) - 1) : NO_BIT;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(pm, (unsigned)o, 64);
            if (tid >= o) pm = t > pm ? t : pm;
            const u32 u = (u32)__shfl_down((int)sm, (unsigned)o, 64);
            if (tid + o < 64) sm = u < sm ? u : sm;
        }
        int pex = __shfl_up(pm, 1u, 64);
        if (tid == 0) pex = -1;
        u32 sex = (u32)__shfl_down((int)sm, 1u, 64);
        if (tid == 63) sex = NO_BIT;
        W.prevSet[tid] = pex;
        W.nextSet[tid] = sex;
    }
    __syncthreads();
    const u32 j = jb + (u32)tid;
    u32 surv = 0, mySlot = 0, hLocal = 0, hSize = 0;
    int hKind = -1;
    bool setBit = false;
    if (j < M) {
        const u32 i = j - j0;
        const u32 w = i >> 5, bit = i & 31;
        const u32 lowmask = (bit == 31) ? 0xFFFFFFFFu : ((2u << bit) - 1u);
        const u32 word = W.bw[w];
        const u32 m = word & lowmask;
        const int si = m ? (int)(w * 32 + 31 - (u32)__clz((int)m)) : W.prevSet[w];
        const u32 nh = (si >= 0) ? j0 + (u32)si : (win ? winLastIncl[win - 1] : 0u);       // first member of my tie
        const u32 di = (u32)(keys[j] >> kbits);
        const u32 gs = desc[di].x, off = loff[di];
        const u32 gp = vals[j];
        mySlot = gs + (j - off);
        v.SA[mySlot] = gp;
        if (nh != off) lab_set(v, gp, lbase[di], gs + (nh - off), gs);      // (the first tie of a group keeps the group's label)
        if (memberR != nullptr) ovr[mySlot] = memberR[j];
        if (nh == j) {
            const u32 m2 = word & ~lowmask;
            const u32 ei = m2 ? (w * 32 + (u32)__ffs((int)m2) - 1) : W.nextSet[w];
            u32 nxt = (ei != NO_BIT) ? j0 + ei : ((win + 1 < nWin) ? winFirstInclRev[nWin - 2 - win] : M);
            if (nxt > M) nxt = M;
            setBit = (j != off);
            hSize = nxt - j;
            hKind = agg_note(A, hSize, false, surv, hLocal);
        }
    }
    if (__ballot(surv != 0) != 0 && (tid & 63) == 0) v.counters[0] = 1;
    __syncthreads();
    if (tid == 0) agg_reserve(A, v);
    __syncthreads();
    if (hKind >= 0) agg_write(A, v, hKind, hLocal, mySlot, hSize, largeNext, (uint2*)nullptr);
    // bits: the lanes of a wave mostly hold consecutive slots (one group), so their bits leave as at most three word-wide ORs per wave
    const int lane = tid & 63;
    const bool live = j < M;
    const u32 s0 = (u32)__shfl((int)mySlot, 0, 64);
    const bool inLine = live && (mySlot == s0 + (u32)lane);
    const unsigned long long hm = __ballot(setBit && inLine), rm = __ballot(inLine);
    if (live && !inLine) {
        if (setBit) atomicOr(&v.gnew[mySlot >> 5], 1u << (mySlot & 31));
        if (rtbits != nullptr) atomicOr(&rtbits[mySlot >> 5], 1u << (mySlot & 31));
    }
    if (lane == 0) {
        const u32 w0 = s0 >> 5, sh = s0 & 31;
        for (int which = 0; which < 2; which++) {
            const unsigned long long mask = which ? rm : hm;
            u32* dst = which ? rtbits : v.gnew;
            if (mask == 0 || dst == nullptr) continue;
            const u32 p0 = (u32)(mask << sh);
            const u32 p1 = sh ? (u32)(mask >> (32 - sh)) : (u32)(mask >> 32);
            const u32 p2 = sh ? (u32)(mask >> (64 - sh)) : 0u;
            if (p0) atomicOr(&dst[w0], p0);
            if (p1) atomicOr(&dst[w0 + 1], p1);
            if (p2) atomicOr(&dst[w0 + 2], p2);
        }
    }
}

// end of a round: the group starts found in it become visible
__global__ __launch_bounds__(256) void k_bwt_f_merge_bits(u32* __restrict__ gbits, u32* __restrict__ gnew, u32 nWords)
{
    const u32 w = blockIdx.x * 256 + threadIdx.x;
    if (w >= nWords) return;
    const u32 x = gnew[w];
    if (x) { gbits[w] |= x; gnew[w] = 0; }
}

// ------------------------------------------------------------------------------------------------
// final: BWT bytes + header (BWTBlockCodec.cpp:58-86)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bwt_f_emit(BwtView v, const u32* __restrict__ base, const u8* __restrict__ ok, const u32* __restrict__ SA,
                                                    FwdView fv, u32* __restrict__ newLen)
{
    const int b = blockIdx.y;
    if (!ok[b]) return;
    const u32 n = v.len[b];
    u32 pIndexSize = 0;
    bwt_fwd_applies(n, v.cap[b], &pIndexSize);
    const int chunks = bwt_chunks(n);
    const u32 hdr = 1 + (u32)chunks * pIndexSize;
    const u8* s = v.src[b];
    u8* d = v.dst[b];
    const u32 bb = base[b];
    const u32 r0 = lab_cur(fv, bb, bb) - bb;               // rank of suffix 0
    // output byte q (hdr + 1 <= q < hdr + n) is the symbol in front of the suffix of rank r' = q - hdr - 1, ranks from r0 on moved
    // up by one (suffix 0 has nothing in front of it); a thread gathers the four bytes of one aligned dword of the output
    const u32 qLo = hdr + 1, qHi = hdr + n;
    const bool al = (reinterpret_cast<uintptr_t>(d) & 3) == 0;
    for (u32 w = (qLo >> 2) + blockIdx.x * 256 + threadIdx.x; 4 * w < qHi; w += gridDim.x * 256) {
        u32 word = 0, have = 0;
#pragma unroll
        for (u32 k = 0; k < 4; k++) {
            const u32 q = 4 * w + k;
            if (q >= qLo && q < qHi) {
                const u32 rr = q - qLo;
                const u32 r = rr + (rr >= r0 ? 1u : 0u);
                const u32 p = SA[bb + r] - bb;
                word |= (u32)s[p - 1] << (8 * k);
                have |= 1u << k;
            }
        }
        if (have == 0xF && al) reinterpret_cast<u32*>(d)[w] = word;
        else for (u32 k = 0; k < 4; k++) if ((have >> k) & 1) d[4 * w + k] = (u8)(word >> (8 * k));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d[hdr] = s[n - 1];
        const u32 st = n / (u32)chunks;
        const u32 step = ((u32)chunks * st == n) ? st : st + 1;
        const u32 logNbChunks = (u32)ilog2_u32((u32)chunks);
        d[0] = (u8)((logNbChunks << 2) | (pIndexSize - 1));
        u32 idx = 1;
        for (int k = 0; k < chunks; k++) {
            const u32 prim = lab_cur(fv, bb + (u32)k * step, bb) - bb;   // primaryIndex - 1
            for (int sh = (int)(pIndexSize - 1) * 8; sh >= 0; sh -= 8) d[idx++] = (u8)(prim >> sh);
        }
        newLen[b] = hdr + n;
    }
}

// ------------------------------------------------------------------------------------------------
// medium groups of the round that ends: the staging slots in order -> flags, (scan), descriptor list; the slots are cleared for the next round
__global__ __launch_bounds__(256) void k_bwt_f_med_flags(const uint2* __restrict__ stage, u32 nSlots, u32* __restrict__ flags)
{
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i < nSlots) flags[i] = stage[i].y ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_bwt_f_med_compact(uint2* __restrict__ stage, u32 nSlots, const u32* __restrict__ prefix, uint2* __restrict__ medNext)
{
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nSlots) return;
    const uint2 d = stage[i];
    if (d.y) { medNext[prefix[i]] = d; stage[i] = make_uint2(0u, 0u); }
}

// knobs (tests, tuning): read from the environment once per process, or set through knz_hip_tune()
struct FwdTuning { int nsym; int noRunRound; int runFallback; int noSuper; int noTextRound; int stats; int noRunOffsets; int noProbe; int link; int gatherWg; int plainLabels; int noFuse; int largeOs; int noPack; };
static FwdTuning& fwd_tuning()
{
    static FwdTuning t = [] {
        FwdTuning x; x.nsym = 0; x.noRunRound = 0; x.runFallback = 0; x.noSuper = 0; x.noTextRound = 0; x.stats = 0; x.noRunOffsets = 0; x.noProbe = 0; x.link = 1; x.gatherWg = 512; x.plainLabels = 0; x.noFuse = 0; x.largeOs = 1000000; x.noPack = 0;     // link: 0 off, 1 on (from h = 32), n > 1: from h = n
        if (getenv("KNZ_BWT_NO_PROBE")) x.noProbe = 1;
        if (getenv("KNZ_BWT_PLAIN_LABELS")) x.plainLabels = 1;
        if (getenv("KNZ_BWT_NO_FUSE")) x.noFuse = 1;
        if (getenv("KNZ_BWT_NO_PACK")) x.noPack = 1;
        if (const char* e = getenv("KNZ_BWT_LINK")) x.link = atoi(e);
        if (getenv("KNZ_BWT_NO_RUN_OFFSETS")) x.noRunOffsets = 1;
        if (getenv("KNZ_BWT_STATS")) x.stats = 1;
        if (const char* e = getenv("KNZ_BWT_NSYM")) x.nsym = atoi(e);
        if (getenv("KNZ_BWT_NO_RUN_ROUND")) x.noRunRound = 1;
        if (getenv("KNZ_BWT_RUN_FALLBACK")) x.runFallback = 1;
        if (getenv("KNZ_BWT_NO_SUPER")) x.noSuper = 1;
        if (const char* e = getenv("KNZ_BWT_NO_TEXT_ROUND")) x.noTextRound = atoi(e);
        return x;
    }();
    return t;
}
int bwt_forward_tune(const char* key, int value)
{
    FwdTuning& t = fwd_tuning();
    if (!strcmp(key, "rs_onesweep")) { prims::rs_onesweep_knob().store(value); return 0; }
    if (!strcmp(key, "bwt_nsym")) t.nsym = value;
    else if (!strcmp(key, "bwt_no_run_round")) t.noRunRound = value;
    else if (!strcmp(key, "bwt_run_fallback")) t.runFallback = value;
    else if (!strcmp(key, "bwt_no_super")) t.noSuper = value;
    else if (!strcmp(key, "bwt_no_text_round")) t.noTextRound = value;
    else if (!strcmp(key, "bwt_stats")) t.stats = value;
    else if (!strcmp(key, "bwt_no_run_offsets")) t.noRunOffsets = value;
    else if (!strcmp(key, "bwt_no_probe")) t.noProbe = value;
    else if (!strcmp(key, "bwt_link")) t.link = value;
    else if (!strcmp(key, "bwt_gather_wg")) t.gatherWg = value;
    else if (!strcmp(key, "bwt_plain_labels")) t.plainLabels = value;
    else if (!strcmp(key, "bwt_no_fuse")) t.noFuse = value;
    else if (!strcmp(key, "bwt_no_pack")) t.noPack = value;      // small groups ranked on plain keys (three counts per pair) also where the packed keys fit
    else if (!strcmp(key, "bwt_large_os")) t.largeOs = value;
    else return -1;
    return 0;
}

struct FwdScratch {
    u64* keysA; u64* keysB;
    u32* valsA; u32* valsB;
    u32* SA; u32* ISA; u32* K; u64* ISA2;
    u32* t0; u32* t1; u32* t2; u32* t3;
    u32* gbits; u32* gnew; u32* rtbits; u32* ovr; size_t gbitsWords;
    uint2* med[2]; uint2* medStage; u32* medFlags; u32* medPrefix; size_t medSlots; uint2* descInfo; uint2* large[2]; uint2* runList; uint4* superList; u32* ebits;
    u32* loff; u32* lbase;
    u32* survTile;       // members still tied after a round's small-group sort, per window
    u32* linkedTile;     // members the link step linked, per window
    u32* base;
    u32* counters;
    u32* seg2;
    void* scanTmp;
    void* rsMem;
    u32* byteHist;       // [nBlocks][256]
    // run round: class table, run-start bit map (+ counts, prefix), run tables
    u32* classTab; u32* rbits; u32* rcount; u32* rprefix;
    u32* runPos; u32* runE; u32* runL; u32* sE; u32* rcnt; u32* moff;
    u64* runKeysA; u64* runKeysB; u64* sKey;
    size_t maxRuns;
};

static size_t fwd_align(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t fwd_carve(u8* p, int nBlocks, size_t total, FwdScratch* w, u32 VS)
{
    u8* q = p;
    auto take = [&](size_t sz) { u8* r = q; q += fwd_align(sz); return r; };
    // (twice the medium groups there can be: the list of the first round may hold groups k_bwt_f_probe took apart, void, beside their parts)
    const size_t maxMed = 2 * (total / (SM_G + 1) + 2), maxLarge = total / (MED_CAP + 1) + 2;
    w->gbitsWords = ((total + 64) / 64 + SM_WIN / 64 + 4) * 2 + 128;       // (the placement kernel looks 2 SM_TS slots behind a window's start)
    w->keysA = (u64*)take(8 * total); w->keysB = (u64*)take(8 * total);
    w->valsA = (u32*)take(4 * total); w->valsB = (u32*)take(4 * total);
    w->SA = (u32*)take(4 * total); w->ISA = (u32*)take(4 * total); w->K = (u32*)take(4 * total);
    w->ISA2 = (VS <= (1u << LAB_BITS)) ? (u64*)take(8 * total) : (u64*)nullptr;     // versioned labels (lab_old / lab_set; blocks up to 2^LAB_BITS bytes)
    w->t0 = (u32*)take(4 * total); w->t1 = (u32*)take(4 * total); w->t2 = (u32*)take(4 * total); w->t3 = (u32*)take(4 * total);
    w->gbits = (u32*)take(4 * w->gbitsWords);
    w->gnew = (u32*)take(4 * w->gbitsWords);
    w->rtbits = (u32*)take(4 * w->gbitsWords);
    w->ovr = (u32*)take(4 * total);
    w->med[0] = (uint2*)take(8 * maxMed); w->med[1] = (uint2*)take(8 * maxMed);
    w->medSlots = total / 256 + 2;
    w->medStage = (uint2*)take(8 * w->medSlots); w->medFlags = (u32*)take(4 * w->medSlots + 64); w->medPrefix = (u32*)take(4 * w->medSlots + 64);
    w->large[0] = (uint2*)take(8 * maxLarge); w->large[1] = (uint2*)take(8 * maxLarge);
    w->runList = (uint2*)take(8 * maxMed);
    w->superList = (uint4*)take(16 * maxMed);
    w->descInfo = (uint2*)take(8 * maxMed);
    w->ebits = (u32*)take(4 * w->gbitsWords);
    w->loff = (u32*)take(4 * (maxMed + 1));
    w->lbase = (u32*)take(4 * (maxMed + 1));
    w->survTile = (u32*)take(4 * (total / SM_TS + 64));
    w->linkedTile = (u32*)take(4 * (total / SM_TS + 64));
    w->base = (u32*)take(4ull * (nBlocks + 2));
    w->counters = (u32*)take(256);
    w->byteHist = (u32*)take(1024ull * (nBlocks + 1));
    w->seg2 = (u32*)take(64);
    w->scanTmp = take(prims::scan_tmp_bytes(total + 16));
    w->rsMem = take(prims::rs_ws_bytes(total, nBlocks + 1));
    w->maxRuns = total / 4 + 64;                                   // (runs of at least 4 symbols; with a shorter round-0 key there can be more: fallback)
    w->classTab = (u32*)take(1024ull * (size_t)nBlocks);
    const size_t rw = total / 32 + SM_WIN / 32 + 136;           // (+ the link step's bit map over the slots, whole windows)
    w->rbits = (u32*)take(4 * rw); w->rcount = (u32*)take(4 * rw); w->rprefix = (u32*)take(4 * rw);
    w->runPos = (u32*)take(4 * w->maxRuns); w->runE = (u32*)take(4 * w->maxRuns); w->runL = (u32*)take(4 * w->maxRuns);
    w->sE = (u32*)take(4 * w->maxRuns); w->rcnt = (u32*)take(4 * w->maxRuns); w->moff = (u32*)take(4 * w->maxRuns + 64);
    w->runKeysA = (u64*)take(8 * w->maxRuns); w->runKeysB = (u64*)take(8 * w->maxRuns); w->sKey = (u64*)take(8 * w->maxRuns);
    return (size_t)(q - p);
}

size_t bwt_forward_scratch_bytes(int nBlocks, u32 VS, size_t total)
{
    FwdScratch w;
    return fwd_carve(nullptr, nBlocks, total, &w, VS) + 4096;
}

#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, s

// Returns 0 or a negative HIP error. Synchronises the stream (the sizes of the work lists are read back per round).
int launch_bwt_forward(hipStream_t s, const XfStage& st, void* scratch, size_t scratchBytes, u32* h_pinned)
{
    BwtView bv; bv.src = st.src; bv.dst = st.dst; bv.len = st.len; bv.cap = st.cap; bv.VS = st.maxLen; bv.nBlocks = st.nBlocks;
    const size_t maxTotal = (size_t)st.nBlocks * bv.VS;
    FwdScratch w;
    if (fwd_carve(reinterpret_cast<u8*>(scratch), st.nBlocks, maxTotal, &w, bv.VS) > scratchBytes) return -2;
    const FwdTuning tune = fwd_tuning();
    { KScope ks_("k_bwt_f_bases"); hipLaunchKernelGGL(k_bwt_bases, dim3(1), dim3(64), 0, s, bv, w.base, st.ok, w.counters + 32); }
    hipMemsetAsync(st.newLen, 0, sizeof(u32) * st.nBlocks, s);
    { KScope ks_("k_bwt_f_r0_hist");                                // byte histograms: the key length of round 0, later its digit counts
      hipMemsetAsync(w.byteHist, 0, 1024ull * st.nBlocks, s);
      hipLaunchKernelGGL(k_bwt_f_bytehist, dim3(64, (unsigned)st.nBlocks), dim3(256), 0, s, bv, w.base, w.byteHist);
      hipLaunchKernelGGL(k_bwt_f_choose_nsym, dim3(1), dim3(256), 0, s, w.byteHist, st.nBlocks, w.counters + 32, w.counters + 33); }
    if (hipMemcpyAsync(h_pinned, w.base + st.nBlocks, 4, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    if (hipMemcpyAsync(h_pinned + 1, w.counters + 32, 8, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    const u32 total = h_pinned[0];
    if (total == 0) return 0;
    FwdView v; v.base = w.base; v.nBlocks = st.nBlocks; v.total = total; v.SA = w.SA; v.ISA = w.ISA; v.K = w.K; v.ISA2 = (w.ISA2 != nullptr && !tune.plainLabels) ? w.ISA2 : (u64*)nullptr; v.round = 0; v.gbits = w.gbits; v.gnew = w.gnew; v.counters = w.counters; v.medStage = w.medStage; v.ovr = nullptr; v.rtbits = nullptr;
    const u32 medSlots = (u32)((size_t)total / 256 + 1);
    hipMemsetAsync(w.medStage, 0, 8ull * w.medSlots, s);
    // the medium groups staged by the kernels of a round -> descriptor list `dst` (in slot order) and counters[1]
    auto compactMedium = [&](uint2* dst) {
        KScope ks_("k_bwt_f_med_compact");
        hipLaunchKernelGGL(k_bwt_f_med_flags, GRID1(medSlots), w.medStage, medSlots, w.medFlags);
        prims::launch_scan<prims::SCAN_SUM_EXCL>(s, w.medFlags, w.medPrefix, medSlots, nullptr, w.scanTmp, w.counters + 1);
        hipLaunchKernelGGL(k_bwt_f_med_compact, GRID1(medSlots), w.medStage, medSlots, w.medPrefix, dst);
    };

    // ---- round 0: the suffixes of every block sorted by their first nsym symbols. Keys = [nsym bytes | position in the block];
    // one stable LSD pass per symbol, the first one reads the text, the blocks are the segments of the sort.
    int pbits = 1;
    while ((1ull << pbits) < (u64)h_pinned[1]) pbits++;           // positions inside the longest block the transform applies to
    // Four or five symbols (as many as fit the key beside the position when the block is larger than 8 / 16 MiB), by the batch's order-0
    // entropy (k_bwt_f_choose_nsym). Measured on 212 MB with 8 MiB blocks, MB/s of the whole round trip, round 3: 4 symbols 4437 (mixed
    // stand-in) / 3859 (text); 5 symbols with h = 5, 10, ...: 4243 / 4155; 5 symbols with h = 4, 8, ...: 4319 / 3970. Round 4 (the
    // doubling offsets stay 4, 8, ...): 4 symbols 5034 / 4314, 5 symbols 4961 / 4564 -- text likes the deeper first round, the stand-in
    // (periodic and sparse stretches, noise) gains nothing from it and pays the fifth pass.
    const int fit = (64 - pbits) / 8;
    const int want = (h_pinned[2] == 5u) ? 5 : 4;                   // (k_bwt_f_choose_nsym: the batch's order-0 entropy)
    int nsym = fit < want ? fit : want;
    if (tune.nsym >= 1) nsym = tune.nsym < fit ? tune.nsym : fit;   // tuning knob: other round-0 key lengths
    if (nsym < 1) return -4;
    prims::RsWs rs = prims::rs_carve(w.rsMem, maxTotal, st.nBlocks + 1, w.base, st.nBlocks);
    { KScope ks_("k_bwt_f_r0_layout"); prims::rs_launch_layout(s, rs); }
    u64* kin = w.keysA; u64* kout = w.keysB;
    const bool oneRead = prims::rs_onesweep_knob().load() != 0 && nsym <= prims::RS_MAXPASS
                         && (size_t)bv.VS < (size_t)prims::RS_VAL_MASK;   // 30-bit counts in the look-back words (a 1 GiB block: count + scatter)
    if (oneRead) {                                                // the digits of all passes counted from the text, once
        KScope ks_("k_bwt_f_r0_sort");
        TextSrc src; src.src = bv.src; src.P = nsym; src.pbits = pbits; src.shift = pbits;
        (void)src;
        hipLaunchKernelGGL(k_bwt_f_r0_counts, dim3((unsigned)rs.L.nSeg), dim3(256), 0, s, bv, w.base, w.byteHist, nsym, rs.L);
        hipLaunchKernelGGL(prims::k_rs_digit_bases, dim3((unsigned)rs.L.nSeg, (unsigned)nsym), dim3(256), 0, s, rs.L);
    }
    for (int pass = 0; pass < nsym; pass++) {
        KScope ks_("k_bwt_f_r0_sort");
        if (pass == 0) {
            TextSrc src; src.src = bv.src; src.P = nsym; src.pbits = pbits; src.shift = pbits;
            if (oneRead) prims::rs_launch_pass_os<u64, false>(s, rs, src, (const u32*)nullptr, kout, (u32*)nullptr, (size_t)bv.VS, pass);
            else prims::rs_launch_pass<u64, false>(s, rs, src, (const u32*)nullptr, kout, (u32*)nullptr, (size_t)bv.VS);
        } else {
            prims::DigitOfKey<u64> src; src.keys = kin; src.shift = pbits + 8 * pass; src.mask = 255u;
            if (oneRead) prims::rs_launch_pass_os<u64, false>(s, rs, src, (const u32*)nullptr, kout, (u32*)nullptr, (size_t)bv.VS, pass);
            else prims::rs_launch_pass<u64, false>(s, rs, src, (const u32*)nullptr, kout, (u32*)nullptr, (size_t)bv.VS);
        }
        std::swap(kin, kout);
    }
    const u64* sortedKeys = kin;                                  // (the last pass wrote into what is now `kin`)
    u64* keysFree = kout;                                         // scratch for the rounds that follow
    u64* keysFree2 = const_cast<u64*>(sortedKeys);                // free again once r0_place has read it
    // every bit from `total` on is set (end sentinel, and windows may look past the end)
    hipMemsetAsync(w.gbits, 0xFF, 4 * w.gbitsWords, s);
    hipMemsetAsync(w.gnew, 0, 4 * w.gbitsWords, s);
    hipMemsetAsync(w.counters, 0, 64, s);
    { KScope ks_("k_bwt_f_r0_flags"); hipLaunchKernelGGL(k_bwt_f_r0_flags, dim3((total + 256 * R0F_ROWS - 1) / (256 * R0F_ROWS)), dim3(256), 0, s, sortedKeys, w.base, st.nBlocks, total, nsym, pbits,
                                                         reinterpret_cast<unsigned long long*>(w.gbits)); }
    // group starts before / after every window of 2048 slots: two scans over ~total/2048 values
    const u32 nWin = (total + SM_WIN - 1) / SM_WIN;
    // (windows of SM_TS slots for the fused placement + text round, of SM_WIN slots for the separate kernels)
    const bool fusedText = tune.noTextRound == 0;
    const u32 nWinP = fusedText ? (total + SM_TS - 1) / SM_TS : nWin;
    { KScope ks_("k_bwt_f_r0_winsum"); hipLaunchKernelGGL(k_bwt_f_r0_winsum, GRID1(nWinP), w.gbits, total, nWinP, w.t0, w.t2, fusedText ? SM_TS / 32 : SM_WIN / 32); }
    { KScope ks_("k_bwt_f_scan_max"); prims::launch_scan<prims::SCAN_MAX_INCL>(s, w.t0, w.t1, nWinP, nullptr, w.scanTmp); }
    { KScope ks_("k_bwt_f_scan_min"); prims::launch_scan<prims::SCAN_MIN_INCL>(s, w.t2, w.t3, nWinP, nullptr, w.scanTmp); }
    int cur = 0;
    int kbits = 1;
    while ((1ull << kbits) < (u64)bv.VS + 2) kbits++;
    // the run-length round needs descriptor index + (kbits + 1) + kbits bits in one 64-bit key
    const bool runRound = (2 * kbits + 1) < 64 && !tune.noRunRound;
    const bool runOffsets = !tune.noRunOffsets;
    if (fusedText) {
        KScope ks_("k_bwt_f_r0_place");
        hipLaunchKernelGGL(k_bwt_f_r0_place_text, dim3(nWinP), dim3(256), 0, s, bv, v, w.t1, w.t3, nWinP, w.large[cur], sortedKeys, nsym, pbits, runRound ? w.runList : (uint2*)nullptr);
        hipLaunchKernelGGL(k_bwt_f_merge_bits, GRID1(total / 32 + 2), w.gbits, w.gnew, total / 32 + 2);
    } else {
        KScope ks_("k_bwt_f_r0_place");
        hipLaunchKernelGGL(k_bwt_f_r0_place, dim3(nWin), dim3(256), 0, s, v, (const u32*)nullptr, w.t1, w.t3, nWin, w.med[cur], w.large[cur],
                           sortedKeys, nsym, pbits, runRound ? w.runList : (uint2*)nullptr);
    }
    if (tune.noTextRound == 2) {
        const u32 nTiles0 = (total + SM_TS - 1) / SM_TS;
        { KScope ks_("k_bwt_f_sort_small_text"); hipLaunchKernelGGL(k_bwt_f_sort_small_text, dim3(nTiles0), dim3(256), 0, s, bv, v, (u32)nsym); }
        { KScope ks_("k_bwt_f_merge_bits"); hipLaunchKernelGGL(k_bwt_f_merge_bits, GRID1(total / 32 + 2), w.gbits, w.gnew, total / 32 + 2); }
    }
    if (hipMemcpyAsync(h_pinned, w.counters, 64, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    u32 nRun = h_pinned[4], runElems = h_pinned[5];
    bool medCompacted = false;
    const int maxKeyBits = 2 * kbits + 1;
    if (nRun && ((u64)nRun > (1ull << (64 - maxKeyBits)) || tune.runFallback)) {     // (the knob: tests force this path)
        // more run groups than the key has index bits left for (thousands of blocks in one batch): they go the ordinary way
        { KScope ks_("k_bwt_f_run_fallback"); hipLaunchKernelGGL(k_bwt_f_run_fallback, GRID1(nRun), v, w.runList, nRun, w.med[cur], w.large[cur]); }
        if (hipMemcpyAsync(h_pinned, w.counters, 64, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
        if (hipStreamSynchronize(s) != hipSuccess) return -1;
        nRun = 0;
    }
    // candidates for k_bwt_f_probe among the medium groups of the list just compacted (their number comes back with the counters)
    u32* probeCand = w.medFlags;                                  // (free between two compactions)
    auto probeScan = [&]() {
        if (tune.noProbe) return;
        KScope ks_("k_bwt_f_probe");
        hipLaunchKernelGGL(k_bwt_f_probe_scan, dim3(256), dim3(256), 0, s, v, w.med[cur], (u32)nsym, v.rtbits, probeCand);
    };
    prims::RsWs rs1 = prims::rs_carve(w.rsMem, maxTotal, st.nBlocks + 1, w.seg2, 1);     // single-segment sorts of the rounds: [0, seg2[1])
    if (nRun) {
        // run lengths of every position (text order), then one sort of the run groups' members on (run length, what follows)
        { KScope ks_("k_bwt_f_run_ends"); hipLaunchKernelGGL(k_bwt_f_run_ends, dim3((total + SM_WIN - 1) / SM_WIN), dim3(256), 0, s, bv, v, reinterpret_cast<u8*>(w.ebits)); }
        { KScope ks_("k_bwt_f_r0_winsum"); hipLaunchKernelGGL(k_bwt_f_r0_winsum, GRID1(nWin), w.ebits, total, nWin, w.t0, w.t2, SM_WIN / 32); }
        { KScope ks_("k_bwt_f_scan_min"); prims::launch_scan<prims::SCAN_MIN_INCL>(s, w.t2, w.t3, nWin, nullptr, w.scanTmp); }
        // run lengths of every position, and the starts of the runs of run-group bytes as a bit map
        hipMemsetAsync(w.classTab, 0xFF, 1024ull * (size_t)st.nBlocks, s);
        { KScope ks_("k_bwt_f_run_classes"); hipLaunchKernelGGL(k_bwt_f_run_classes, GRID1(nRun), bv, v, w.runList, nRun, w.classTab); }
        { KScope ks_("k_bwt_f_run_len"); hipLaunchKernelGGL(k_bwt_f_run_len, dim3(nWin), dim3(256), 0, s, bv, v, w.ebits, w.t3, nWin, w.K, w.classTab, (u32)nsym,
                                                            reinterpret_cast<unsigned long long*>(w.rbits), w.rcount); }
        // the runs themselves, compacted in position order (the windows of k_bwt_f_run_len cover whole words up to nWin * 2048)
        const u32 nRW = nWin * (SM_WIN / 32);
        { KScope ks_("k_bwt_f_scan_sum"); prims::launch_scan<prims::SCAN_SUM_EXCL>(s, w.rcount, w.rprefix, nRW, nullptr, w.scanTmp, w.counters + 8); }
        { KScope ks_("k_bwt_f_run_compact"); hipLaunchKernelGGL(k_bwt_f_run_compact, GRID1(nRW), w.rbits, w.rprefix, nRW, w.runPos); }
        if (hipMemcpyAsync(h_pinned + 8, w.counters + 6, 12, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;     // longest run, -, number of runs
        if (hipStreamSynchronize(s) != hipSuccess) return -1;
        const u32 nRuns = h_pinned[10];
        int hbits = 1;
        while ((1ull << hbits) < (u64)h_pinned[8] + 1) hbits++;
        hbits++;                                                  // both halves of the order: R and 2^hbits - 1 - R
        const int keyBits = hbits + kbits;
        int rbits = 0;
        while ((1u << rbits) < nRun) rbits++;
        int idxBits = 1;
        while ((1ull << idxBits) < (u64)nRuns) idxBits++;
        if (nRuns == 0 || nRuns > w.maxRuns || rbits + hbits > 32 || idxBits + kbits + 1 + rbits > 64) {
            // more runs than the tables hold (a round-0 key shorter than 4 symbols) or fields that do not fit their words: the run groups
            // go the ordinary way
            { KScope ks_("k_bwt_f_run_fallback"); hipLaunchKernelGGL(k_bwt_f_run_fallback, GRID1(nRun), v, w.runList, nRun, w.med[cur], w.large[cur]); }
            if (hipMemcpyAsync(h_pinned, w.counters, 64, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
            if (hipStreamSynchronize(s) != hipSuccess) return -1;
            nRun = 0;
        }
      if (nRun) {
        { KScope ks_("k_bwt_f_large_prefix"); hipLaunchKernelGGL(k_bwt_f_large_prefix, dim3(1), dim3(1024), 0, s, w.runList, nRun, w.loff, w.base, st.nBlocks, w.lbase); }
        { KScope ks_("k_bwt_f_run_table"); hipLaunchKernelGGL(k_bwt_f_run_table, GRID1(nRuns), bv, v, w.runPos, nRuns, w.K, w.classTab, kbits, idxBits, w.runKeysA, w.runE, w.runL); }
        const u64* rkSorted;
        { KScope ks_("k_bwt_f_sort_runs");
          hipLaunchKernelGGL(prims::k_rs_one_segment, dim3(1), dim3(64), 0, s, w.seg2, nRuns);
          prims::rs_launch_layout(s, rs1);
          const int r = prims::rs_sort<u64, false>(s, rs1, w.runKeysA, w.runKeysB, (u32*)nullptr, (u32*)nullptr, (size_t)nRuns, idxBits, idxBits + kbits + 1 + rbits);
          rkSorted = r ? w.runKeysB : w.runKeysA; }
        { KScope ks_("k_bwt_f_run_sorted"); hipLaunchKernelGGL(k_bwt_f_run_sorted, GRID1(nRuns), rkSorted, nRuns, idxBits, w.runE, w.runL, (u32)nsym, w.sE, w.sKey, w.rcnt); }
        { KScope ks_("k_bwt_f_scan_sum"); prims::launch_scan<prims::SCAN_SUM_EXCL>(s, w.rcnt, w.moff, nRuns, nullptr, w.scanTmp); }
        { KScope ks_("k_bwt_f_run_members"); hipLaunchKernelGGL(k_bwt_f_run_members, dim3((runElems + 2047) / 2048), dim3(256), 0, s, w.moff, nRuns, runElems, w.sKey, kbits, hbits,
                                                                (u32)nsym, keysFree); }
        u64* rk; u32* rv = w.valsA;
        { KScope ks_("k_bwt_f_sort_members");
          hipLaunchKernelGGL(prims::k_rs_one_segment, dim3(1), dim3(64), 0, s, w.seg2, runElems);
          prims::rs_launch_layout(s, rs1);
          const int r = prims::rs_sort<u64, false>(s, rs1, keysFree, keysFree2, (u32*)nullptr, (u32*)nullptr, (size_t)runElems, 32, 32 + hbits + rbits);
          const u64* sortedM = r ? keysFree2 : keysFree;
          rk = r ? keysFree : keysFree2;
          hipLaunchKernelGGL(k_bwt_f_run_expand, GRID1(runElems), sortedM, runElems, w.sKey, w.sE, kbits, hbits, rk, rv, runOffsets ? w.valsB : (u32*)nullptr); }
        {
            // ties among the sorted members as a bit map, heads and sizes per window of 2048 members (the bit map reuses the run-end map,
            // which nobody reads any more)
            u32* mbits = w.ebits;
            const u32 nWinM = (runElems + SM_WIN - 1) / SM_WIN;
            { KScope ks_("k_bwt_f_large_flags"); hipLaunchKernelGGL(k_bwt_f_run_flags, dim3(nWinM * (SM_WIN / 256)), dim3(256), 0, s, rk, runElems, reinterpret_cast<unsigned long long*>(mbits)); }
            { KScope ks_("k_bwt_f_r0_winsum"); hipLaunchKernelGGL(k_bwt_f_r0_winsum, GRID1(nWinM), mbits, runElems, nWinM, w.t0, w.t2, SM_WIN / 32); }
            { KScope ks_("k_bwt_f_scan_max"); prims::launch_scan<prims::SCAN_MAX_INCL>(s, w.t0, w.t1, nWinM, nullptr, w.scanTmp); }
            { KScope ks_("k_bwt_f_scan_min"); prims::launch_scan<prims::SCAN_MIN_INCL>(s, w.t2, w.t3, nWinM, nullptr, w.scanTmp); }
            { KScope ks_("k_bwt_f_large_place");
              if (runOffsets) hipMemsetAsync(w.rtbits, 0, 4 * w.gbitsWords, s);
              hipLaunchKernelGGL(k_bwt_f_run_place, dim3((runElems + 255) / 256), dim3(256), 0, s, v, w.runList, w.loff, runElems, keyBits, rk, rv, mbits, w.t1, w.t3, nWinM, w.large[cur],
                                 runOffsets ? (const u32*)w.valsB : (const u32*)nullptr, runOffsets ? w.ovr : (u32*)nullptr, runOffsets ? w.rtbits : (u32*)nullptr, w.lbase);
              if (runOffsets) { v.ovr = w.ovr; v.rtbits = w.rtbits; } }
        }
        { KScope ks_("k_bwt_f_merge_bits"); hipLaunchKernelGGL(k_bwt_f_merge_bits, GRID1(total / 32 + 2), w.gbits, w.gnew, total / 32 + 2); }
        compactMedium(w.med[cur]);
        probeScan();
        medCompacted = true;
        if (hipMemcpyAsync(h_pinned, w.counters, 64, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
        if (hipStreamSynchronize(s) != hipSuccess) return -1;
      }
    }
    if (!medCompacted) {
        compactMedium(w.med[cur]);
        probeScan();
        if (hipMemcpyAsync(h_pinned, w.counters, 64, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
        if (hipStreamSynchronize(s) != hipSuccess) return -1;
    }
    u32 surv = h_pinned[0], nMed = h_pinned[1], nLarge = h_pinned[2], largeElems = h_pinned[3];
    if (const u32 nCand = tune.noProbe ? 0u : h_pinned[14]) {
        // Periodic stretches of a period the doubling offsets never meet, by looking at the text (k_bwt_f_probe), before the first round: the
        // groups it takes apart are void in the list (length 0), their parts that are medium groups are appended to it
        if (!v.rtbits) { hipMemsetAsync(w.rtbits, 0, 4 * w.gbitsWords, s); v.ovr = w.ovr; v.rtbits = w.rtbits; }
        hipMemsetAsync(w.counters, 0, 8, s);
        { KScope ks_("k_bwt_f_probe");
          hipLaunchKernelGGL(k_bwt_f_probe, dim3(std::min<u32>(nCand, 4096)), dim3(512), 0, s, bv, v, w.med[cur], probeCand, nCand, (u32)nsym, w.med[cur ^ 1], w.large[cur], w.ovr, w.rtbits);
          hipLaunchKernelGGL(k_bwt_f_merge_bits, GRID1(total / 32 + 2), w.gbits, w.gnew, total / 32 + 2); }
        compactMedium(w.med[cur] + nMed);
        if (hipMemcpyAsync(h_pinned, w.counters, 8, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
        if (hipStreamSynchronize(s) != hipSuccess) return -1;
        surv |= h_pinned[0];
        nMed += h_pinned[1];
    }
    if (tune.stats) fprintf(stderr, "after round 0 (nsym %d, total %u): run groups %u (%u members); small left %u, medium %u, large %u (%u members)\n",
                            nsym, total, nRun, runElems, surv, nMed, nLarge, largeElems);
    std::chrono::steady_clock::time_point statT = std::chrono::steady_clock::now();

    const int npass = (kbits + 7) / 8;
    const u32 nTiles = (total + SM_TS - 1) / SM_TS;
    // The doubling starts at the largest power of two the round-0 depth covers (groups and labels of depth nsym >= h are what a round
    // with offset h needs): h = 4, 8, 16, ... meets the periods real data has (record and row sizes are powers of two more often than
    // not), which is what lets the chain round (k_bwt_f_super) see a group look at itself.
    u32 h = 1;
    while (2 * h <= (u32)nsym) h <<= 1;
    u32 survMembers = total;                                        // (not counted before the first round: assume many)
    int linkMode = 0;                                               // 0 not tried, 1 sampled, 2 on, 3 applied (to be judged), 4 off
    u32 linkStepUsed = 1, linkRetryH = 0xFFFFFFFFu;
    LinkTrial linkTr; linkTr.n = 0;
    int linkTrials = 0;
    while (surv || nMed || nLarge) {
        if (h > bv.VS) return -5;                                    // cannot happen: suffixes of one block differ in length
        { KScope ks_("k_bwt_f_round"); hipMemsetAsync(w.counters, 0, 64, s); }      // (the scope counts the doubling rounds for the profile)
        v.round++;                                                   // (the number the round's label writes carry; at most log2(block) + a few)
        if (v.round > 255) return -5;
        const int nxt = cur ^ 1;
        // -- small groups inside long repeats: links first (k_bwt_f_link_small), once the rounds are past the depth where most ties are chance
        // Whether it pays is a property of the data (it does where groups are whole repeats: copied spans, files that hold a part twice;
        // it does not where every group has members that leave it one by one, as in the file mix of config 9): the first application, at
        // h = 32, is a trial; what it linked against what was still tied after the round decides whether the rounds that follow apply it
        // too, and every application is judged again. The trial is made on the windows of up to THREE WHOLE BLOCKS of the batch, the first,
        // the middle and the last one (a run never leaves its block, so a block's runs are complete; a sample of windows all over the
        // batch would cut every run), and must pay in each of them: 0.15 ms per 8 MiB where the step applied to a 212 MB batch costs
        // 2-4 ms. (Round 5 looked at the first block only: in the real-file corpus that is one shared object, the trial said yes, and
        // the application to the whole batch cost 3.5 ms for 1.7 ms of later rounds.)
        bool linkNow = false;
        u32 linkStep = 1;
        if (surv && tune.link && h >= (u32)(tune.link > 1 ? tune.link : 32)) {
            if (linkMode == 1 || linkMode == 3) {                    // judge the application of the round before
                const u32 linkedN = h_pinned[14], tiedN = h_pinned[15];
                const bool paid = linkedN >= 4096 && tiedN < linkedN / 2;
                linkMode = paid ? 2 : 4;                              // 2: apply, 4: not now
                if (!paid) linkRetryH = (linkTrials < 2) ? h * 8 : 0xFFFFFFFFu;    // (chance ties of the early rounds may have hidden the repeats: once more, three rounds on)
                if (tune.stats) fprintf(stderr, "link step: %u members linked, %u of the windows' members still tied after the round -> %s\n", linkedN, tiedN, paid ? "on" : "off");
            }
            if (linkMode == 4 && h >= linkRetryH) linkMode = 0;
            if (linkMode == 0 && survMembers >= total / 8) { linkNow = true; linkMode = 1; linkTrials++; }              // the trial
            else if (linkMode == 2 && survMembers >= total / 64) { linkNow = true; linkMode = 3; }
        }
        if (linkNow) {
            KScope ks_("k_bwt_f_link");
            u32* posflag = w.ebits; u32* linked = w.rbits; u32* rev = w.rcount; u32* sufRev = w.rprefix;
            // the trial looks at whole blocks: the first, the middle and the last one of the batch (blocks next to each other would share a window)
            linkTr.n = 0;
            if (linkMode == 1) {
                linkTr.blk[linkTr.n++] = 0;
                if (st.nBlocks >= 5) linkTr.blk[linkTr.n++] = st.nBlocks / 2;
                if (st.nBlocks >= 3) linkTr.blk[linkTr.n++] = st.nBlocks - 1;
            }
            const u32 nW = total / 32 + 1;
            hipMemsetAsync(posflag, 0, 4 * (size_t)nW, s);
            hipMemsetAsync(linked, 0, 4 * ((size_t)nTiles * (SM_TS / 32) + 64), s);
            if (!v.rtbits) { hipMemsetAsync(w.rtbits, 0, 4 * w.gbitsWords, s); v.ovr = w.ovr; v.rtbits = w.rtbits; }
            // (a block covers at most VS / SM_TS + 2 windows)
            const u32 nTrial = (u32)std::min<u64>((u64)nTiles, ((u64)bv.VS + SM_TS - 1) / SM_TS + 2);
            const dim3 gridL(linkMode == 1 ? nTrial : (nTiles + linkStep - 1) / linkStep, (unsigned)(linkTr.n ? linkTr.n : 1));
            hipLaunchKernelGGL(k_bwt_f_link_small, gridL, dim3(256), 0, s, v, posflag, linked, tune.stats, linkStep, w.linkedTile, linkTr);
            hipLaunchKernelGGL(k_bwt_f_link_zeros, GRID1(nW), posflag, nW, rev);
            prims::launch_scan<prims::SCAN_MIN_INCL>(s, rev, sufRev, nW, nullptr, w.scanTmp);
            hipLaunchKernelGGL(k_bwt_f_link_apply, gridL, dim3(256), 0, s, v, posflag, sufRev, nW, linked, h, w.ovr, w.rtbits, linkStep, linkTr);
            linkStepUsed = linkStep;
        }
        // -- all keys first (with versioned labels the small groups fetch theirs in the kernel that sorts them: k_bwt_f_small_fused)
        const bool fused = v.ISA2 != nullptr && !tune.noFuse;
        if (surv && !fused) { KScope ks_("k_bwt_f_gather_small"); hipLaunchKernelGGL(k_bwt_f_gather_small, dim3(nTiles), dim3(256), 0, s, v, h, tune.stats); }
        if (nMed) {
            // (the list is in slot order: k_bwt_f_med_compact)
            KScope ks_("k_bwt_f_gather_desc");
            // Threads per workgroup (knob bwt_gather_wg, default 512): a group's keys are three dependent loads, and more, smaller workgroups keep
            // more groups in flight per CU. Real files 4.13 -> 2.99 ms (512) / 3.16 (256), text 3.26 -> 2.39 / 2.46, the stand-in 2.54 -> 2.56 / 2.71.
            if (tune.gatherWg == 256) hipLaunchKernelGGL(k_bwt_f_gather_desc<256>, dim3(2048), dim3(256), 0, s, v, w.med[cur], nMed, h, w.descInfo, tune.stats);
            else if (tune.gatherWg == 1024) hipLaunchKernelGGL(k_bwt_f_gather_desc<1024>, dim3(256), dim3(1024), 0, s, v, w.med[cur], nMed, h, w.descInfo, tune.stats);
            else hipLaunchKernelGGL(k_bwt_f_gather_desc<512>, dim3(1024), dim3(512), 0, s, v, w.med[cur], nMed, h, w.descInfo, tune.stats);
        }
        int lbits = 0;
        bool small32 = false;
        u64* lkA = keysFree; u64* lkB = keysFree2;
        if (nLarge) {
            while ((1u << lbits) < nLarge) lbits++;
            small32 = (kbits + lbits) <= 32;
            { KScope ks_("k_bwt_f_large_prefix"); hipLaunchKernelGGL(k_bwt_f_large_prefix, dim3(1), dim3(1024), 0, s, w.large[cur], nLarge, w.loff, w.base, st.nBlocks, w.lbase); }
            KScope ks_("k_bwt_f_large_keys");
            if (small32) hipLaunchKernelGGL(k_bwt_f_large_keys<u32>, GRID1(largeElems), v, w.large[cur], nLarge, w.loff, largeElems, h, kbits, reinterpret_cast<u32*>(lkA), w.valsA);
            else hipLaunchKernelGGL(k_bwt_f_large_keys<u64>, GRID1(largeElems), v, w.large[cur], nLarge, w.loff, largeElems, h, kbits, lkA, w.valsA);
        }
        // -- then the refinements
        if (surv) { KScope ks_("k_bwt_f_sort_small");
                    if (fused && pbits <= 23 && !tune.noPack) hipLaunchKernelGGL(k_bwt_f_small_fused<true>, dim3(nTiles), dim3(256), 0, s, v, h, tune.link ? w.survTile : (u32*)nullptr, tune.stats);
                    else if (fused) hipLaunchKernelGGL(k_bwt_f_small_fused<false>, dim3(nTiles), dim3(256), 0, s, v, h, tune.link ? w.survTile : (u32*)nullptr, tune.stats);
                    else hipLaunchKernelGGL(k_bwt_f_sort_small, dim3(nTiles), dim3(256), 0, s, v, tune.link ? w.survTile : (u32*)nullptr);
                    if (linkNow) hipLaunchKernelGGL(k_bwt_f_link_payoff, dim3(1), dim3(256), 0, s, v, w.linkedTile, w.survTile, nTiles, linkStepUsed, w.counters + 14, linkTr);
                    if (tune.link) prims::launch_scan<prims::SCAN_SUM_EXCL>(s, w.survTile, w.survTile, nTiles, nullptr, w.scanTmp, w.counters + 13); }
        if (nMed) {
            // two workgroup shapes over the same list, each takes the groups of its size class: 256 threads x 8 elements (21 KB of LDS,
            // groups up to 2048) and 512 threads x 16 elements (76 KB: two groups per CU in flight)
            uint4* sup = tune.noSuper ? (uint4*)nullptr : w.superList;
            { KScope ks_("k_bwt_f_sort_medium");
              const dim3 gridM(std::min<u32>(nMed, 8192));
              hipLaunchKernelGGL((k_bwt_f_sort_medium<256, 8>), gridM, dim3(256), 0, s, v, w.med[cur], nMed, npass, SM_G, w.med[nxt], w.large[nxt], w.descInfo, sup);
              hipLaunchKernelGGL((k_bwt_f_sort_medium<512, 16>), gridM, dim3(512), 0, s, v, w.med[cur], nMed, npass, 2048u, w.med[nxt], w.large[nxt], w.descInfo, sup); }
            // groups whose majority looks at the group itself (sort_medium has listed them; the kernel reads the count itself)
            if (sup) { KScope ks_("k_bwt_f_super"); hipLaunchKernelGGL(k_bwt_f_super, dim3(std::min<u32>(nMed, 512)), dim3(1024), 0, s, v, sup, h, npass, w.med[nxt], w.large[nxt]); }
        }
        if (nLarge) {
            u32* k32a = reinterpret_cast<u32*>(lkA); u32* k32b = reinterpret_cast<u32*>(lkB);
            int r;
            { KScope ks_("k_bwt_f_sort_large");
              hipLaunchKernelGGL(prims::k_rs_one_segment, dim3(1), dim3(64), 0, s, w.seg2, largeElems);
              prims::rs_launch_layout(s, rs1);
              // (count + scatter passes here: with the many passes of these keys, most of them on constant digits, counting all of
              // them ahead costs more than it saves -- period 3 / 5 / 7 / 768 at 8 MiB: 11.1-17.6 ms against 11.8-18.2)
              // (knob bwt_large_os: from that many members on the passes are the one-sweep ones; real files' 27 M members in the first round: 4.11 -> 3.94 ms,
              // and counted for every sort they cost the many small ones of periodic data more than they save: 4.11 -> 4.56)
              const bool os = tune.largeOs != 0 && largeElems >= (u32)tune.largeOs;
              r = small32 ? prims::rs_sort<u32, true>(s, rs1, k32a, k32b, w.valsA, w.valsB, (size_t)largeElems, 0, kbits + lbits, os)
                          : prims::rs_sort<u64, true>(s, rs1, lkA, lkB, w.valsA, w.valsB, (size_t)largeElems, 0, kbits + lbits, os); }
            const u32* sk32 = r ? k32b : k32a; const u64* sk64 = r ? lkB : lkA; const u32* sv = r ? w.valsB : w.valsA;
            { KScope ks_("k_bwt_f_large_flags");
              if (small32) hipLaunchKernelGGL(k_bwt_f_large_flags<u32>, GRID1(largeElems), sk32, largeElems, w.t0, w.t2);
              else hipLaunchKernelGGL(k_bwt_f_large_flags<u64>, GRID1(largeElems), sk64, largeElems, w.t0, w.t2); }
            { KScope ks_("k_bwt_f_scan_max"); prims::launch_scan<prims::SCAN_MAX_INCL>(s, w.t0, w.t1, largeElems, nullptr, w.scanTmp); }
            { KScope ks_("k_bwt_f_scan_min"); prims::launch_scan<prims::SCAN_MIN_INCL>(s, w.t2, w.t3, largeElems, nullptr, w.scanTmp); }
            { KScope ks_("k_bwt_f_large_place");
              if (small32) hipLaunchKernelGGL(k_bwt_f_large_place<u32>, GRID1(largeElems), v, w.large[cur], w.loff, largeElems, kbits, sk32, sv, w.t1, w.t3, w.med[nxt], w.large[nxt], (const u32*)nullptr, (u32*)nullptr, (u32*)nullptr, w.lbase);
              else hipLaunchKernelGGL(k_bwt_f_large_place<u64>, GRID1(largeElems), v, w.large[cur], w.loff, largeElems, kbits, sk64, sv, w.t1, w.t3, w.med[nxt], w.large[nxt], (const u32*)nullptr, (u32*)nullptr, (u32*)nullptr, w.lbase); }
        }
        { KScope ks_("k_bwt_f_merge_bits"); hipLaunchKernelGGL(k_bwt_f_merge_bits, GRID1(total / 32 + 2), w.gbits, w.gnew, total / 32 + 2); }
        compactMedium(w.med[nxt]);
        if (hipMemcpyAsync(h_pinned, w.counters, 64, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
        if (hipStreamSynchronize(s) != hipSuccess) return -1;
        surv = h_pinned[0]; nMed = h_pinned[1]; nLarge = h_pinned[2]; largeElems = h_pinned[3];
        survMembers = h_pinned[13];
        if (tune.stats) {
            // windows that still hold tied small groups (from the scanned per-window counts; developer statistics only)
            std::vector<u32> hs(nTiles);
            u32 activeTiles = 0;
            if (tune.link && hipMemcpyAsync(hs.data(), w.survTile, 4 * (size_t)nTiles, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess) {
                for (u32 t = 0; t + 1 < nTiles; t++) activeTiles += hs[t + 1] != hs[t] ? 1u : 0u;
                fprintf(stderr, "  windows with tied small groups after the round: %u of %u\n", activeTiles, nTiles);
            }
            const std::chrono::steady_clock::time_point now = std::chrono::steady_clock::now();
            fprintf(stderr, "round h=%u (%.3f ms): small members worked on %u in %u groups, medium members %u; after it: small left %u, medium groups %u, large %u (%u members); %u groups took the chain round; %u small members still tied\n",
                    h, std::chrono::duration<double, std::milli>(now - statT).count(), h_pinned[10], h_pinned[11], h_pinned[12], surv, nMed, nLarge, largeElems, h_pinned[7], h_pinned[13]);
            statT = now;
        }
        cur = nxt;
        h <<= 1;
    }
    const dim3 gridB((unsigned)std::min<size_t>(((size_t)bv.VS + 255) / 256, 4096), st.nBlocks);
    { KScope ks_("k_bwt_f_emit"); hipLaunchKernelGGL(k_bwt_f_emit, gridB, dim3(256), 0, s, bv, w.base, st.ok, w.SA, v, st.newLen); }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace knz
