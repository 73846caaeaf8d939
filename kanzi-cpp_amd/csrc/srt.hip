// SRT (sorted rank transform) on gfx950: forward one wave per 4 KiB tile; inverse parallel up to records, then one chain per block.
//
// Reference being replaced: transform/SRT.cpp:22-109 (forward), :111-204 (inverse), :206-244 (preprocess:
// symbols by descending frequency, ties by ascending value), :246-308 (header: 256 var-ints).
//
// What the transform is: a move-to-front over the RUNS of the input (initial list = symbols in order of first
// appearance), whose ranks are not written in place but appended to one bucket per symbol (buckets laid out in
// "preprocess" order); the bytes of a run after its head are zeros in the same bucket.
//
//   forward  tile parallel (round 3; one wave per block before): the ranks are move-to-front ranks from an initial list in
//            order of first appearance, so the block is cut into 4 KiB tiles whose start lists and bucket offsets come from
//            scans over per-tile tables (see "forward, tile parallel" below); inside a tile the 256-entry list is held in
//            registers (4 entries per lane, SWAR zero-byte test + ballot to find a symbol, one DPP shift to rotate).
//   inverse  inherently serial over runs (which bucket is read next depends on the list head). Everything that is not the chain is
//            taken off it: the body is cut into per-run records in parallel first, and the chain wave only steps the list (31
//            instructions per run) while a second wave writes the runs and a third keeps the per-symbol record queues filled
//            (see "inverse" below). Round 2: 6.0 s per 32 MiB block of BWT(text); now 1.4-1.5 s.
#include "common.hpp"
#include "stages.hpp"
#include "prims.hpp"

namespace knz {

// rotate list positions [0, rank] right by one and put `front` at position 0 (rank = 4*lane0 + byteIdx)
__device__ __forceinline__ u32 srt_to_front(u32 w, int lane, int lane0, int byteIdx, u32 front)
{
    const u32 prev = (u32)__builtin_amdgcn_update_dpp(0, (int)w, 0x138, 0xF, 0xF, false);   // wave_shr:1 (lane i <- lane i-1)
    const u32 carry = (lane == 0) ? front : (prev >> 24);
    const u32 shifted = (w << 8) | carry;
    if (lane < lane0) return shifted;
    if (lane == lane0) {
        const u32 mask = (byteIdx == 3) ? 0xFFFFFFFFu : ((1u << (8 * (byteIdx + 1))) - 1u);
        return (shifted & mask) | (w & ~mask);
    }
    return w;
}

// positions [1, rank] move down by one; position rank receives `ins` (pick = 5) or keeps its value (pick = rank & 3)
// (memmove(&r2s[0], &r2s[1], rank) [+ r2s[rank] = ins], SRT.cpp:189-199). One byte permute per lane over (this lane's four entries,
// the next lane's first, ins): lanes below the rank's lane shift by one, lanes above keep, the rank's lane gets a selector put
// together in scalar registers. rank, pick, ins are wave uniform.
__device__ __forceinline__ u32 srt_drop_front(u32 w, int lane, u32 rank, u32 pick, u32 ins)
{
    const u32 lane0 = rank >> 2, sh = (rank & 3u) << 3;
    const u32 next = (u32)__builtin_amdgcn_update_dpp(0, (int)w, 0x130, 0xF, 0xF, true);    // wave_shl:1 (lane i <- lane i+1, 0 into lane 63)
    const u32 hi = (next & 0xFFu) | (ins << 8);                  // permute bytes 4 (the entry that moves in from the next lane) and 5 (ins)
    const u32 low = (1u << sh) - 1u, at = 0xFFu << sh;
    const u32 selAt = (0x04030201u & low) | (0x03020100u & ~(low | at)) | (pick << sh);
    const u32 sel = ((u32)lane < lane0) ? 0x04030201u : (((u32)lane == lane0) ? selAt : 0x03020100u);
    return __builtin_amdgcn_perm(hi, w, sel);
}

// order index of every present symbol: number of present symbols with a larger frequency, or an equal one and
// a smaller value (SRT.cpp:206-244 is a shell sort under exactly this total order).  Returns nbSymbols.
__device__ u32 srt_order(int lane, const u32* freqs, u8* symbols)
{
    u32 present = 0;
    for (int k = 0; k < 4; k++) {
        const int c = 4 * lane + k;
        const u32 fc = freqs[c];
        if (fc == 0) continue;
        present++;
        u32 r = 0;
        for (int q = 0; q < 256; q++) {
            const u32 fq = freqs[q];
            r += (fq > fc || (fq == fc && q < c && fq != 0)) ? 1u : 0u;
        }
        symbols[r] = (u8)c;
    }
    return wave_sum(present);
}

__global__ __launch_bounds__(256) void k_srt_zero(XfStage st)
{
    const int b = blockIdx.y;
    const u32 length = st.len[b];
    if (length == 0 || st.cap[b] < length + 1024) return;
    const u32 total = length + 1024;
    uint4* d = reinterpret_cast<uint4*>(st.dst[b]);
    const bool al = (reinterpret_cast<uintptr_t>(st.dst[b]) & 15) == 0;
    const u32 n16 = (total + 15) >> 4;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) {
        if (al && 16 * i + 16 <= st.cap[b]) d[i] = z;
        else for (u32 k = 16 * i; k < 16 * i + 16 && k < st.cap[b] && k < total; k++) st.dst[b][k] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// inverse
// ------------------------------------------------------------------------------------------------
// Serial over runs by construction: after a run of symbol c, c's bucket says at which rank c will be found next (or that it is
// exhausted), c is put there, and the list head is the next symbol. What is NOT serial is reading the buckets: a bucket is a sequence
// of records (zeros that follow a head = rest of the run, rank of the next head or END), and every record can be cut out of the body in
// parallel. So the body is pre-parsed into records first (head flags as a bit map, a scan of the word counts, head positions, then
// record k from heads k and k + 1), and the chain steps once per RUN on ready records: per step one LDS read (the symbol's next
// record, from a queue of 32 records per symbol), the list update, the run handed to the writer wave ("the chain kernels" below).
// Round 2 walked the raw bucket bytes through 64-byte LDS windows: 6.0 s per 32 MiB block.
constexpr u32 SRT_QD = 32;          // records per symbol held in LDS

struct SrtInv {
    u32* info;         // per block [8]: 0 ok so far, 1 header bytes, 2 body length, 3 symbols present, 6 the general chain has to run
    u32* bstart;       // [nBlocks][256] bucket start in the body (first entry), 0xFFFFFFFF: symbol absent
    u32* bend;         // [nBlocks][256]
    u8* order;         // [nBlocks][256] symbols in bucket order
    u32* bits;         // head flags, one bit per body position, blocks back to back (wordsPer words each)
    u32* wcount; u32* wprefix;
    u32* headPos;      // [heads]
    u64* rec;          // [heads] low word: zeros behind the head, high word: next rank (0 = END)
    u32 wordsPer;
};

__global__ __launch_bounds__(64) void k_srt_i_header(XfStage st, SrtInv w)
{
    const int b = blockIdx.x;
    const int lane = lane_id();
    const int total = (int)st.len[b];
    __shared__ u32 freqs[256];
    __shared__ u8 symbols[256];
    __shared__ u8 hdrBytes[1280];
    __shared__ int shHdr;
    u32* info = w.info + 8 * b;
    if (lane == 0) { st.ok[b] = 0; st.newLen[b] = 0; }
    if (lane < 8) info[lane] = 0;
    for (int i = lane; i < 256; i += 64) { w.bstart[b * 256 + i] = 0xFFFFFFFFu; w.bend[b * 256 + i] = 0; symbols[i] = 0; }
    if (total == 0) { if (lane == 0) st.ok[b] = 1; return; }
    if (total < 256) return;                                   // SRT.cpp:122
    const u8* in = st.src[b];
    for (int i = lane; i < 1280; i += 64) hdrBytes[i] = (i < total) ? in[i] : 0;
    __syncthreads();
    if (lane == 0) {
        // decodeHeader, SRT.cpp:279-308
        int srcIdx = 0;
        bool okh = true;
        for (int i = 0; i < 256 && okh; i++) {
            u32 res = 0;
            int shift = 0;
            for (int j = 0; j < 5; j++) {
                if (srcIdx >= total) { okh = false; break; }
                const u32 val = hdrBytes[srcIdx++];
                res |= ((val & 0x7F) << shift);
                if ((val & 0x80) == 0) break;
                if (j == 4) { okh = false; break; }
                shift += 7;
            }
            freqs[i] = res;
        }
        shHdr = okh ? srcIdx : -1;
    }
    __syncthreads();
    const int hdr = shHdr;
    if (hdr < 0) return;
    const int length = total - hdr;
    if (length < 0 || (u32)length > st.cap[b]) return;
    const int nbSymbols = (int)srt_order(lane, freqs, symbols);
    __syncthreads();
    if (lane == 0) {
        int okb = 1;
        long long bucketPos = 0;
        for (int i = 0; i < nbSymbols; i++) {
            const u8 c = symbols[i];
            if (bucketPos >= length) { okb = 0; break; }          // SRT.cpp:152-153
            w.bstart[b * 256 + c] = (u32)bucketPos;
            bucketPos += (long long)freqs[c];
            if (bucketPos > 0xFFFFFFF0ll) { okb = 0; break; }
            w.bend[b * 256 + c] = (u32)bucketPos;
        }
        info[1] = (u32)hdr; info[2] = (u32)length; info[3] = (u32)nbSymbols;
        info[0] = (u32)okb;
    }
    for (int i = lane; i < 256; i += 64) w.order[b * 256 + i] = symbols[i];
}

// head flags of 32 body positions: the first entry of a bucket, and every non-zero entry
__global__ __launch_bounds__(256) void k_srt_i_flags(XfStage st, SrtInv w)
{
    const int b = blockIdx.y;
    const u32* info = w.info + 8 * b;
    const u32 wi = blockIdx.x * 256 + threadIdx.x;
    if (wi >= w.wordsPer) return;
    u32 word = 0;
    if (info[0]) {
        const u32 length = info[2], nb = info[3];
        const u8* src = st.src[b] + info[1];
        const u32 i0 = 32u * wi;
        if (i0 < length) {
            if (i0 + 32 <= length && ((reinterpret_cast<uintptr_t>(src + i0) & 3) == 0)) {
                const u32* p = reinterpret_cast<const u32*>(src + i0);
#pragma unroll
                for (int q = 0; q < 8; q++) { const u32 x = p[q]; for (int k = 0; k < 4; k++) if ((x >> (8 * k)) & 0xFF) word |= 1u << (4 * q + k); }
            } else {
                for (u32 k = 0; k < 32 && i0 + k < length; k++) if (src[i0 + k]) word |= 1u << k;
            }
            // bucket starts inside this word: the buckets are laid out in `order`, so their starts ascend
            u32 lo = 0, hi = nb;
            while (lo < hi) { const u32 mid = (lo + hi) >> 1; if (w.bstart[b * 256 + w.order[b * 256 + mid]] < i0) lo = mid + 1; else hi = mid; }
            for (u32 k = lo; k < nb; k++) { const u32 s0 = w.bstart[b * 256 + w.order[b * 256 + k]]; if (s0 >= i0 + 32) break; word |= 1u << (s0 - i0); }
        }
    }
    const size_t at = (size_t)b * w.wordsPer + wi;
    w.bits[at] = word;
    w.wcount[at] = (u32)__popc(word);
}

__global__ __launch_bounds__(256) void k_srt_i_heads(SrtInv w, u32 nWords)
{
    const u32 wi = blockIdx.x * 256 + threadIdx.x;
    if (wi >= nWords) return;
    u32 word = w.bits[wi];
    u32 at = w.wprefix[wi];
    const u32 i0 = (wi % w.wordsPer) * 32u;
    while (word) {
        const u32 k = (u32)__ffs((int)word) - 1;
        w.headPos[at++] = i0 + k;
        word &= word - 1;
    }
}

// record of head k: zeros behind it up to the next head (or the bucket's end), and the next head's rank unless that head starts another bucket
__global__ __launch_bounds__(256) void k_srt_i_records(XfStage st, SrtInv w, int nBlocks)
{
    const int b = blockIdx.y;
    const u32* info = w.info + 8 * b;
    if (!info[0]) return;
    const u32 first = w.wprefix[(size_t)b * w.wordsPer];
    const u32 end = (b + 1 < nBlocks) ? w.wprefix[(size_t)(b + 1) * w.wordsPer] : w.wprefix[(size_t)nBlocks * w.wordsPer];
    const u8* src = st.src[b] + info[1];
    for (u32 k = first + blockIdx.x * 256 + threadIdx.x; k < end; k += gridDim.x * 256) {
        const u32 pos = w.headPos[k];
        // the bucket that holds this head: the last one that starts at or before it (the buckets are laid out in `order`)
        u32 lo = 0, hi = info[3];
        while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (w.bstart[b * 256 + w.order[b * 256 + mid]] <= pos) lo = mid; else hi = mid; }
        const u32 bendc = w.bend[b * 256 + w.order[b * 256 + lo]];
        u32 nextPos = bendc, r = 0;
        if (pos >= bendc) nextPos = pos + 1;                     // (a stray byte behind the last bucket of a malformed block: nobody reads it)
        else if (k + 1 < end && w.headPos[k + 1] < bendc) { nextPos = w.headPos[k + 1]; r = src[nextPos]; }
        w.rec[k] = ((u64)r << 32) | (u64)(nextPos - pos - 1);
    }
}

// first record of every bucket (per block): the record index of its start position = word prefix + bits below
__device__ __forceinline__ u32 srt_rec_index(const SrtInv& w, int b, u32 pos)
{
    const size_t at = (size_t)b * w.wordsPer + (pos >> 5);
    return w.wprefix[at] + (u32)__popc(w.bits[at] & ((1u << (pos & 31)) - 1u));
}

// The chain kernels: three waves per block.
//   wave 0  the chain. Per run: the front symbol's next record (one LDS read at an address that comes out of registers), the list
//           update, and the run (symbol, length) dropped into a pair of VGPRs that go to the run ring every 64 runs. The record of
//           the NEXT run's symbol (list[1], whatever this run's record says) is read a run ahead: the rest of the step sits between
//           the read and its use (44 cycles of LDS latency against ≈ 250 for a step, tools/issuebench.hip).
//   wave 1  the writer: takes 64 runs at a time from the ring, scans their lengths, stores the bytes.
//   wave 2  the refiller: a symbol's queue is two halves of 16 records; when the chain finishes a half it posts the symbol, the refiller
//           loads the next 16 records (four requests side by side) while the chain works through the other half.
// The waves talk through LDS only (ring heads and tails, `filled`): one wave's LDS operations are carried out in program order, so data
// written before a counter is visible to whoever reads the counter first.
//
// Two chains. k_srt_inverse is the one a well-formed block runs on: one list entry per lane (position 64 j + l in lane l of register
// j), an entry = (symbol << 8) | byte offset of the symbol's next record inside its queue = the LDS address of that record, so a step
// reads list[1] with one v_readlane and has its address; the offset travels with the symbol through the list. Moving the front to
// rank r < 64 is a DPP shift, a select under a scalar lane mask and a v_writelane. That only equals the reference while every symbol
// is in the list once, which a malformed block can break (first bytes of the buckets that are no permutation, a rank beyond the
// live part of the list that pulls stale copies in): the chain then gives up (info[6]) and k_srt_inverse_general redoes the block
// with the reference's semantics to the letter: list as bytes (four per lane), per-symbol read counters in a VGPR.
constexpr u32 SRT_QH = SRT_QD / 2;
constexpr u32 SRT_RING = 1024;                                  // runs between chain and writer
constexpr u32 SRT_REQ = 512;                                    // (a symbol has at most two requests open: never full)
constexpr u64 SRT_END = 0ull;                                   // what a symbol without records left reads: no zeros, END (SRT.cpp:190-198: one more byte, then it leaves the list)

#ifdef KNZ_EMU
#define KNZ_LDS_FENCE() ((void)__ballot(1))
#define KNZ_SPIN() hipemu::wave_spin()
#define KNZ_LDS_LOAD(p) (*reinterpret_cast<const volatile u32*>(p))
#define KNZ_LDS_STORE(p, v) (*reinterpret_cast<volatile u32*>(p) = (v))
#else
#define KNZ_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define KNZ_SPIN() __builtin_amdgcn_s_sleep(1)
#define KNZ_LDS_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define KNZ_LDS_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#endif
__device__ __forceinline__ u32 lds_peek(const u32* p) { return (u32)__builtin_amdgcn_readfirstlane((int)KNZ_LDS_LOAD(p)); }
__device__ __forceinline__ void lds_poke(u32* p, u32 v) { KNZ_LDS_STORE(p, v); }

// Lane-level register edits with wave uniform operands. (This compiler has no writelane builtin; the lane select goes through m0,
// which nothing else in these kernels uses.)
#ifdef KNZ_EMU
__device__ __forceinline__ u32 srt_writelane(u32 val, u32 sel, u32 old) { return (u32)lane_id() == sel ? val : old; }
__device__ __forceinline__ void srt_writelane2(u32& a, u32& b, u32 va, u32 vb, u32 sel) { if ((u32)lane_id() == sel) { a = va; b = vb; } }
// lanes below n take `shifted`, lane n takes ins when insert is set
__device__ __forceinline__ u32 srt_lane_merge(u32 w, u32 shifted, u32 n, bool insert, u32 ins)
{
    const u32 l = (u32)lane_id();
    return l < n ? shifted : ((l == n && insert) ? ins : w);
}
#else
// (an "s" operand the compiler believes divergent would come out as a VGPR: readfirstlane, which folds away for scalar values)
__device__ __forceinline__ u32 srt_uni(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u32 srt_writelane(u32 val, u32 sel, u32 old)
{
    val = srt_uni(val); sel = srt_uni(sel);
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(val), "s"(sel) : "m0");
    return old;
}
__device__ __forceinline__ void srt_writelane2(u32& a, u32& b, u32 va, u32 vb, u32 sel)
{
    va = srt_uni(va); vb = srt_uni(vb); sel = srt_uni(sel);
    asm("s_mov_b32 m0, %4\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0" : "+v"(a), "+v"(b) : "s"(va), "s"(vb), "s"(sel) : "m0");
}
__device__ __forceinline__ u32 srt_lane_merge(u32 w, u32 shifted, u32 n, bool insert, u32 ins)
{
    n = srt_uni(n); ins = srt_uni(ins);
    const u64 below = (1ull << n) - 1ull;                        // (n < 64)
    if (insert) asm("s_mov_b32 m0, %4\n\tv_cndmask_b32 %0, %0, %1, %2\n\tv_writelane_b32 %0, %3, m0" : "+v"(w) : "v"(shifted), "s"(below), "s"(ins), "s"(n) : "m0");
    else asm("v_cndmask_b32 %0, %0, %1, %2" : "+v"(w) : "v"(shifted), "s"(below));
    return w;
}
#endif

struct SrtLds {
    u64* q;            // [256][SRT_QD] the next records of every symbol (ring of two halves)
    u32* recFirst;     // [256] the symbol's records in w.rec: first, one past the last
    u32* recEnd;
    u32* filled;       // [256] halves (16 records) of the symbol loaded so far (the refiller's)
    u32* doneHalves;   // [256] halves the chain has finished (the chain's own)
    u32* listw;        // [64] the initial list, a byte per rank
    u64* ring;         // [SRT_RING] runs: (symbol << 32) | length
    u32* req;          // [SRT_REQ]
    u32* ctl;          // run ring head, run ring tail, request head, chain done (1) / gave up (2)
};

// what the three waves share; returns whether the initial list is a permutation of the present symbols (wave uniform)
__device__ __forceinline__ bool srt_inv_setup(const SrtLds& L, const SrtInv& w, int b, const u8* src, u32 length, u32 nbSymbols, int tid)
{
    __shared__ u32 listOk;
    if (tid < 64) L.listw[tid] = 0;
    if (tid < 4) L.ctl[tid] = 0;
    for (int c = tid; c < 256; c += 192) {
        const u32 s0 = w.bstart[b * 256 + c];
        u32 first = 0, last = 0;
        if (s0 != 0xFFFFFFFFu) {
            first = srt_rec_index(w, b, s0);
            // records of the bucket: heads in [start, end)
            const u32 e0 = w.bend[b * 256 + c];
            last = (e0 < length) ? srt_rec_index(w, b, e0) : ((b + 1 < (int)gridDim.x) ? w.wprefix[(size_t)(b + 1) * w.wordsPer] : w.wprefix[(size_t)gridDim.x * w.wordsPer]);
        }
        L.recFirst[c] = first; L.recEnd[c] = last; L.filled[c] = 2; L.doneHalves[c] = 0;
    }
    __syncthreads();
    for (u32 e = (u32)tid; e < 256 * SRT_QD; e += 192) {
        const u32 sym = e / SRT_QD, idx = L.recFirst[sym] + (e % SRT_QD);
        L.q[e] = (idx < L.recEnd[sym]) ? w.rec[idx] : SRT_END;
    }
    if (tid == 0) {
        u8* listb = reinterpret_cast<u8*>(L.listw);
        u32 seen[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        u32 good = 1;
        for (u32 i = 0; i < nbSymbols; i++) {                   // SRT.cpp:150-157
            const u8 c = w.order[b * 256 + i];
            const u32 r = src[w.bstart[b * 256 + c]];
            if (r >= nbSymbols || ((seen[r >> 5] >> (r & 31)) & 1u)) good = 0;
            seen[r >> 5] |= 1u << (r & 31);
            listb[r] = c;
        }
        listOk = good;
    }
    __syncthreads();
    return listOk != 0;
}

__device__ __forceinline__ void srt_inv_writer(const SrtLds& L, u8* dst, int lane)
{
    u32 tail = 0, off = 0;
    for (;;) {
        const u32 done = lds_peek(&L.ctl[3]);
        const u32 h = lds_peek(&L.ctl[0]);                      // (read after the flag: a set flag means this head is final)
        if (h == tail) { if (done) break; KNZ_SPIN(); continue; }
        KNZ_LDS_FENCE();
        const u32 n = (h - tail < 64u) ? h - tail : 64u;
        const u64 e = ((u32)lane < n) ? L.ring[(tail + (u32)lane) & (SRT_RING - 1)] : 0ull;
        const u32 len = (u32)e, sym = (u32)(e >> 32);
        const u32 incl = wave_incl_scan(len);
        const u32 at = off + incl - len;
        const u32 shortLen = len < 8u ? len : 8u;
        for (u32 t = 0; t < shortLen; t++) dst[at + t] = (u8)sym;
        u64 m = __ballot(len > 8u);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            m &= m - 1;
            const u32 la = (u32)__shfl((int)at, l, 64), ll = (u32)__shfl((int)len, l, 64), ls = (u32)__shfl((int)sym, l, 64);
            for (u32 t = 8u + (u32)lane; t < ll; t += 64) dst[la + t] = (u8)ls;
        }
        off += (u32)__shfl((int)incl, 63, 64);
        tail += n;
        KNZ_LDS_FENCE();                                        // (the ring has been read before its slots are handed back)
        if (lane == 0) lds_poke(&L.ctl[1], tail);
    }
}

__device__ __forceinline__ void srt_inv_refiller(const SrtLds& L, const SrtInv& w, int lane)
{
    u32 tail = 0;
    for (;;) {
        const u32 done = lds_peek(&L.ctl[3]);
        const u32 h = lds_peek(&L.ctl[2]);
        if (h == tail) { if (done) break; KNZ_SPIN(); continue; }
        KNZ_LDS_FENCE();
        u32 n = (h - tail < 4u) ? h - tail : 4u;
        // a symbol that asks twice inside one batch: its second request waits for the next batch (it reads `filled`)
        const u32 s0 = lds_peek(&L.req[tail & (SRT_REQ - 1)]), s1 = lds_peek(&L.req[(tail + 1) & (SRT_REQ - 1)]);
        const u32 s2 = lds_peek(&L.req[(tail + 2) & (SRT_REQ - 1)]), s3 = lds_peek(&L.req[(tail + 3) & (SRT_REQ - 1)]);
        if (n > 1 && s1 == s0) n = 1;
        if (n > 2 && (s2 == s0 || s2 == s1)) n = 2;
        if (n > 3 && (s3 == s0 || s3 == s1 || s3 == s2)) n = 3;
        const u32 g = (u32)lane >> 4, i = (u32)lane & 15u;
        const u32 sym = g == 0 ? s0 : g == 1 ? s1 : g == 2 ? s2 : s3;
        u32 f = 0;
        if (g < n) {
            f = L.filled[sym];
            const u32 idx = L.recFirst[sym] + f * SRT_QH + i;
            L.q[sym * SRT_QD + (f & 1u) * SRT_QH + i] = (idx < L.recEnd[sym]) ? w.rec[idx] : SRT_END;
        }
        KNZ_LDS_FENCE();
        if (g < n && i == 0) lds_poke(&L.filled[sym], f + 1);
        tail += n;
    }
}

// the chain's side of the two rings
struct SrtChainIo {
    u32 head, reqHead, outLen, outSym;
};

__device__ __forceinline__ void srt_chain_flush(const SrtLds& L, SrtChainIo& io, u32 n, int lane)
{
    while (io.head - lds_peek(&L.ctl[1]) > SRT_RING) KNZ_SPIN();
    KNZ_LDS_FENCE();
    if ((u32)lane < n) L.ring[(io.head - n + (u32)lane) & (SRT_RING - 1)] = ((u64)io.outSym << 32) | io.outLen;
    KNZ_LDS_FENCE();
    if (lane == 0) lds_poke(&L.ctl[0], io.head);
}

__device__ __forceinline__ void srt_chain_push(const SrtLds& L, SrtChainIo& io, u32 sym, u32 len, int lane)
{
    srt_writelane2(io.outLen, io.outSym, len, sym, io.head & 63u);
    io.head++;
    if (__builtin_expect((io.head & 63u) == 0, 0)) srt_chain_flush(L, io, 64u, lane);
}

// a half of the symbol's queue is finished: the refiller may have it; the half the chain enters now must have arrived
__device__ __forceinline__ void srt_chain_half(const SrtLds& L, SrtChainIo& io, u32 sym, int lane)
{
    const u32 dh = lds_peek(&L.doneHalves[sym]) + 1;
    if (lane == 0) { L.doneHalves[sym] = dh; L.req[io.reqHead & (SRT_REQ - 1)] = sym; }
    io.reqHead++;
    KNZ_LDS_FENCE();
    if (lane == 0) lds_poke(&L.ctl[2], io.reqHead);
    while (lds_peek(&L.filled[sym]) < dh + 1) KNZ_SPIN();
    KNZ_LDS_FENCE();
}

__device__ __forceinline__ void srt_chain_close(const SrtLds& L, SrtChainIo& io, u32 how, int lane)
{
    if (io.head & 63u) srt_chain_flush(L, io, io.head & 63u, lane);
    KNZ_LDS_FENCE();
    if (lane == 0) lds_poke(&L.ctl[3], how);
}

#define KNZ_SRT_INV_SHARED \
    __shared__ u64 q[256 * SRT_QD]; \
    __shared__ u32 recFirst[256], recEnd[256], filled[256], doneHalves[256], listw[64]; \
    __shared__ u64 ring[SRT_RING]; \
    __shared__ u32 req[SRT_REQ]; \
    __shared__ u32 ctl[4]; \
    SrtLds L; \
    L.q = q; L.recFirst = recFirst; L.recEnd = recEnd; L.filled = filled; L.doneHalves = doneHalves; L.listw = listw; L.ring = ring; L.req = req; L.ctl = ctl;

// position 64 j + l of the list in lane l of a[j]; positions [0, rank) take their successor, position rank takes ins (insert) or
// keeps its value
__device__ __forceinline__ void srt_list_drop(u32 (&a)[4], u32 rank, bool insert, u32 ins)
{
    const u32 j = rank >> 6, n = rank & 63u;
    const u32 s0 = (u32)__builtin_amdgcn_update_dpp(0, (int)a[0], 0x130, 0xF, 0xF, true);   // wave_shl:1 (lane i <- lane i+1)
    if (__builtin_expect(j == 0, 1)) { a[0] = srt_lane_merge(a[0], s0, n, insert, ins); return; }
    // registers below the rank's move as a whole and take their last entry from the next one's first
    a[0] = srt_writelane((u32)__builtin_amdgcn_readlane((int)a[1], 0), 63u, s0);
    const u32 s1 = (u32)__builtin_amdgcn_update_dpp(0, (int)a[1], 0x130, 0xF, 0xF, true);
    if (j == 1) { a[1] = srt_lane_merge(a[1], s1, n, insert, ins); return; }
    a[1] = srt_writelane((u32)__builtin_amdgcn_readlane((int)a[2], 0), 63u, s1);
    const u32 s2 = (u32)__builtin_amdgcn_update_dpp(0, (int)a[2], 0x130, 0xF, 0xF, true);
    if (j == 2) { a[2] = srt_lane_merge(a[2], s2, n, insert, ins); return; }
    a[2] = srt_writelane((u32)__builtin_amdgcn_readlane((int)a[3], 0), 63u, s2);
    const u32 s3 = (u32)__builtin_amdgcn_update_dpp(0, (int)a[3], 0x130, 0xF, 0xF, true);
    a[3] = srt_lane_merge(a[3], s3, n, insert, ins);
}

#ifndef KNZ_EMU
// The runs of the fast chain that need nothing special, in the instruction order they are meant to have (the compiler wraps the
// loop's rare exits in flag registers and copies that double its length). One trip = one run whose front goes back to a rank inside
// the first register (1 <= term <= lim1) and that ends before the block does:
//   the record of list[1] is read (LDS address = the entry + the queues' base), its entry with the record taken goes back to lane 1;
//   the run (symbol, zeros + 1) goes to lane head & 63 of the two staging registers;
//   lanes below the rank take their successor (DPP shift under a scalar lane mask), the rank's lane the front's entry;
//   list[1]'s entry and record become the front's.
// Leaves with 1 BEFORE touching anything when the next run is not of that kind, with 2 AFTER a run when a queue half is finished
// (the front's entry then still has the offset's carry in it) or the staging registers are full. Wait states: the SGPRs the VALU
// writes (v_readlane, v_readfirstlane) are read by SALU instructions or at least two instructions later; the register the DPP move
// reads was last written five instructions earlier.
__device__ __forceinline__ u32 srt_hot_runs(u32& a0, u32& outLen, u32& outSym, u32& ec, u32& z, u32& term, u32& remaining, u32& head, u32 lim1, const u64* q)
{
    const u32 vbase = (u32)reinterpret_cast<uintptr_t>(q);      // (the low half of a generic LDS address is the LDS offset)
    u32 why, e1, en, t, u, sym, tmp, vaddr;
    u64 mask;
    asm volatile(
        "1:\n\t"
        "s_add_u32 %[t], %[z], 1\n\t"
        "s_cmp_ge_u32 %[t], %[rem]\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_add_u32 %[u], %[term], -1\n\t"
        "s_cmp_ge_u32 %[u], %[lim1]\n\t"
        "s_cbranch_scc1 2f\n\t"
        "v_readlane_b32 %[e1], %[a0], 1\n\t"
        "s_sub_u32 %[rem], %[rem], %[t]\n\t"
        "s_lshr_b32 %[sym], %[ec], 8\n\t"
        "s_and_b32 %[u], %[head], 63\n\t"
        "v_add_u32 %[vaddr], %[e1], %[vbase]\n\t"
        "ds_read_b64 v[100:101], %[vaddr]\n\t"
        "s_add_u32 %[en], %[e1], 8\n\t"
        "v_writelane_b32 %[a0], %[en], 1\n\t"
        "s_mov_b32 m0, %[u]\n\t"
        "v_writelane_b32 %[outLen], %[t], m0\n\t"
        "v_writelane_b32 %[outSym], %[sym], m0\n\t"
        "s_add_u32 %[head], %[head], 1\n\t"
        "s_bfm_b64 %[mask], %[term], 0\n\t"
        "v_mov_b32_dpp %[tmp], %[a0] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_mov_b32 m0, %[term]\n\t"
        "v_cndmask_b32 %[a0], %[a0], %[tmp], %[mask]\n\t"
        "v_writelane_b32 %[a0], %[ec], m0\n\t"
        "s_mov_b32 %[ec], %[en]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_readfirstlane_b32 %[z], v100\n\t"
        "v_readfirstlane_b32 %[term], v101\n\t"
        "s_and_b32 %[u], %[en], 0x7f\n\t"
        "s_cbranch_scc0 3f\n\t"
        "s_and_b32 %[u], %[head], 63\n\t"
        "s_cbranch_scc1 1b\n\t"
        "3:\n\t"
        "s_mov_b32 %[why], 2\n\t"
        "s_branch 4f\n\t"
        "2:\n\t"
        "s_mov_b32 %[why], 1\n\t"
        "4:"
        : [a0] "+v"(a0), [outLen] "+v"(outLen), [outSym] "+v"(outSym), [ec] "+s"(ec), [z] "+s"(z), [term] "+s"(term), [rem] "+s"(remaining), [head] "+s"(head),
          [why] "=&s"(why), [e1] "=&s"(e1), [en] "=&s"(en), [t] "=&s"(t), [u] "=&s"(u), [sym] "=&s"(sym), [tmp] "=&v"(tmp), [vaddr] "=&v"(vaddr), [mask] "=&s"(mask)
        : [lim1] "s"(lim1), [vbase] "v"(vbase)
        : "m0", "scc", "v100", "v101", "memory");
    return why;
}
#else
// the same loop for the CPU emulation of tests/emu (it has no assembler): statement for statement what the instructions above do
__device__ __forceinline__ u32 srt_hot_runs(u32& a0, u32& outLen, u32& outSym, u32& ec, u32& z, u32& term, u32& remaining, u32& head, u32 lim1, const u64* q)
{
    for (;;) {
        const u32 full = z + 1;
        if (full >= remaining) return 1;
        if (term - 1u >= lim1) return 1;
        const u32 e1 = (u32)__builtin_amdgcn_readlane((int)a0, 1);
        remaining -= full;
        const u64 rcn = *reinterpret_cast<const u64*>(reinterpret_cast<const char*>(q) + e1);
        const u32 en = e1 + 8u;
        a0 = srt_writelane(en, 1u, a0);
        srt_writelane2(outLen, outSym, full, ec >> 8, head & 63u);
        head++;
        const u32 s0 = (u32)__builtin_amdgcn_update_dpp(0, (int)a0, 0x130, 0xF, 0xF, true);
        a0 = srt_lane_merge(a0, s0, term, true, ec);
        ec = en;
        z = (u32)__builtin_amdgcn_readfirstlane((int)(u32)rcn); term = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(rcn >> 32));
        if ((en & 0x7Fu) == 0 || (head & 63u) == 0) return 2;
    }
}
#endif

__global__ __launch_bounds__(192) void k_srt_inverse(XfStage st, SrtInv w)
{
    const int b = blockIdx.x;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32* info = w.info + 8 * b;
    if (!info[0]) return;                                      // (ok / newLen were set by the header kernel: refused, or an empty block)
    const u32 length = info[2];
    u32 nbSymbols = info[3];
    const u8* src = st.src[b] + info[1];
    KNZ_SRT_INV_SHARED
    if (!srt_inv_setup(L, w, b, src, length, nbSymbols, tid)) { if (tid == 0) info[6] = 1; return; }
    if (wave == 1) {
        srt_inv_writer(L, st.dst[b], lane);
        const u32 how = lds_peek(&L.ctl[3]);
        if (lane == 0) {
            if (how == 1u) { st.ok[b] = 1; st.newLen[b] = length; }
            else info[6] = 1;                                   // the chain gave up: k_srt_inverse_general takes the block
        }
        return;
    }
    if (wave == 2) { srt_inv_refiller(L, w, lane); return; }

    // ---- chain
    const u8* listb = reinterpret_cast<const u8*>(L.listw);
    u32 a[4];
    for (int j = 0; j < 4; j++) a[j] = (u32)listb[64 * j + lane] << 8;
    SrtChainIo io = { 0, 0, 0, 0 };
    u32 remaining = length;
    // the record at list entry e (symbol << 8 | offset in its queue) and the entry with the record taken
    auto fetch = [&](u32 e, u32& en) -> u64 {
        const u64 r = *reinterpret_cast<const u64*>(reinterpret_cast<const char*>(L.q) + e);
        en = e + 8u;
        if (__builtin_expect((en & (8u * SRT_QH - 1u)) == 0, 0)) {
            if ((en & 0xFFu) == 0) en -= 0x100u;                // (the offset wraps inside its byte)
            srt_chain_half(L, io, e >> 8, lane);
        }
        return r;
    };
    u32 ec;                                                     // the front's entry (its record already taken)
    const u64 rc0 = fetch((u32)__builtin_amdgcn_readlane((int)a[0], 0), ec);
    u32 z = (u32)__builtin_amdgcn_readfirstlane((int)(u32)rc0), term = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(rc0 >> 32));
    u32 how = 1;
    bool open = true;                                           // (false: the block is complete, or given up)
    u32 lim1 = (nbSymbols < 64u ? nbSymbols : 64u) - 1u;        // the ranks the short way handles: 1 .. lim1
    if (nbSymbols >= 2) for (;;) {
        if (srt_hot_runs(a[0], io.outLen, io.outSym, ec, z, term, remaining, io.head, lim1, L.q) == 2u) {
            if ((ec & (8u * SRT_QH - 1u)) == 0) {
                if ((ec & 0xFFu) == 0) ec -= 0x100u;
                srt_chain_half(L, io, ec >> 8, lane);
            }
            if ((io.head & 63u) == 0) srt_chain_flush(L, io, 64u, lane);
            continue;
        }
        // (the long way, any run.) Whatever this run's record says (the front goes back to a rank >= 1, or leaves the list), the symbol
        // of the NEXT run is the one behind the front now: its record is read first and picked up at the bottom, behind the work on this run.
        u32 en;
        const u64 rcn = fetch((u32)__builtin_amdgcn_readlane((int)a[0], 1), en);
        a[0] = srt_writelane(en, 1u, a[0]);
        const u32 full = z + 1;
        const u32 emit = full < remaining ? full : remaining;
        srt_chain_push(L, io, ec >> 8, emit, lane);
        if (full >= remaining) { open = false; break; }
        remaining -= emit;
        if (term == 0) {                                        // the front is exhausted: it leaves the list
            nbSymbols--;
            lim1 = (nbSymbols < 64u ? nbSymbols : 64u) - 1u;
            srt_list_drop(a, nbSymbols, false, 0);
        } else {
            if (term >= nbSymbols) { open = false; how = 2; break; }
            srt_list_drop(a, term, true, ec);
        }
        ec = en;
        z = (u32)__builtin_amdgcn_readfirstlane((int)(u32)rcn); term = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(rcn >> 32));
        if (nbSymbols < 2) break;
    }
    if (open) {
        // one symbol left: a well-formed block ends inside this run
        const u32 full = z + 1;
        const u32 emit = full < remaining ? full : remaining;
        srt_chain_push(L, io, ec >> 8, emit, lane);
        if (full < remaining) {
            if (term == 0) srt_chain_push(L, io, ec >> 8, remaining - emit, lane);      // SRT.cpp:194-195: the last symbol fills the rest
            else how = 2;
        }
    }
    srt_chain_close(L, io, how, lane);
}

// the reference's semantics whatever the block holds (see above); blocks the fast chain finished are skipped
__global__ __launch_bounds__(192) void k_srt_inverse_general(XfStage st, SrtInv w)
{
    const int b = blockIdx.x;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32* info = w.info + 8 * b;
    if (!info[0] || !info[6]) return;
    const u32 length = info[2];
    u32 nbSymbols = info[3];
    const u8* src = st.src[b] + info[1];
    KNZ_SRT_INV_SHARED
    (void)srt_inv_setup(L, w, b, src, length, nbSymbols, tid);
    if (wave == 1) {
        srt_inv_writer(L, st.dst[b], lane);
        if (lane == 0) { st.ok[b] = 1; st.newLen[b] = length; }
        return;
    }
    if (wave == 2) { srt_inv_refiller(L, w, lane); return; }

    // ---- chain
    u32 lw = L.listw[lane];
    u32 qv = 0;                                                 // byte j: where symbol 64 j + lane reads next in its queue (byte offset, mod 256)
    SrtChainIo io = { 0, 0, 0, 0 };
    u32 remaining = length;
    // the symbol's next record; it counts as taken
    auto fetch = [&](u32 sym) -> u64 {
        const u32 sl = sym & 63u, sh = (sym >> 3) & 0x18u;
        const u32 x = (u32)__builtin_amdgcn_readlane((int)qv, (int)sl);
        const u32 k8 = (x >> sh) & 0xFFu;
        const u64 r = *reinterpret_cast<const u64*>(reinterpret_cast<const char*>(L.q) + ((sym << 8) | k8));
        u32 nx = x + (8u << sh);
        if (__builtin_expect(((k8 + 8u) & (8u * SRT_QH - 1u)) == 0, 0)) {
            if (k8 == 0xF8u && sh < 24u) nx -= 1u << (sh + 8);  // (the byte wrapped: take the carry back out of its neighbour)
            srt_chain_half(L, io, sym, lane);
        }
        qv = srt_writelane(nx, sl, qv);
        return r;
    };
    u32 c = (u32)__builtin_amdgcn_readfirstlane((int)lw) & 0xFFu;
    const u64 rc0 = fetch(c);
    u32 z = (u32)__builtin_amdgcn_readfirstlane((int)(u32)rc0), term = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(rc0 >> 32));
    for (;;) {
        // Whatever this run's record says (c goes back to a rank >= 1, or leaves the list), the symbol of the NEXT run is the one
        // behind the front now -- a stale entry when the list is down to one symbol, read all the same (SRT.cpp:189-199 goes on
        // with r2s[1] whatever it holds): its record is read first and picked up at the bottom, behind the work on this run.
        const u32 cn = ((u32)__builtin_amdgcn_readfirstlane((int)lw) >> 8) & 0xFFu;
        const u64 rcn = fetch(cn);
        const u32 full = z + 1;
        const u32 emit = full < remaining ? full : remaining;
        srt_chain_push(L, io, c, emit, lane);
        if (full >= remaining) break;
        remaining -= emit;
        if (term == 0) {                                        // c is exhausted
            if (nbSymbols == 1) { srt_chain_push(L, io, c, remaining, lane); break; }   // SRT.cpp:194-195: the last symbol fills the rest
            nbSymbols--;
            lw = srt_drop_front(lw, lane, nbSymbols, nbSymbols & 3u, 0);
        } else {
            lw = srt_drop_front(lw, lane, term, 5u, c);
        }
        c = cn;
        z = (u32)__builtin_amdgcn_readfirstlane((int)(u32)rcn); term = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(rcn >> 32));
    }
    srt_chain_close(L, io, 1u, lane);
}

// ------------------------------------------------------------------------------------------------
// forward, tile parallel
// ------------------------------------------------------------------------------------------------
// The ranks SRT writes are plain move-to-front ranks of the byte sequence (a byte inside a run finds its symbol at the front:
// rank 0, list unchanged) from an initial list in order of first appearance; what is special is where they go: the k-th
// occurrence of symbol c lands at bucket(c) + k. So the block is cut into tiles of 4 KiB like MTFT: the list at a tile start
// is the symbols seen so far by their last occurrence (exclusive prefix maximum over per-tile tables), then the ones still to
// come by their first occurrence in the block; the output offset of a tile inside every bucket is an exclusive prefix sum over
// per-tile histograms. Both scans run in two levels (groups of ~sqrt(tiles) tiles, then over the groups). The body of the output
// is zero-filled first, only non-zero ranks are stored.
constexpr u32 SRT_T = 4096;

__host__ __device__ inline u32 srt_seg_tiles(u32 perTiles) { u32 g = 1; while (g * g < perTiles) g <<= 1; return g; }

struct SrtWs {
    u32* tileCnt;      // [nBlocks][perTiles][256] counts, then exclusive inside the tile's segment
    u32* tileLast;     // [nBlocks][perTiles][256] last occurrence (position + 1), then exclusive maximum inside the segment
    u32* segSum;       // [nBlocks][nSeg][256]
    u32* segMax;       // [nBlocks][nSeg][256]
    u32* firstOcc;     // [nBlocks][256] first occurrence of the symbol in the block (0xFFFFFFFF: absent)
    u32* bstart;       // [nBlocks][256] start of the symbol's bucket in the body
    u32* hdrLen;       // [nBlocks] (0: the transform does not apply)
    int perTiles; u32 segT, nSeg;
};

__global__ __launch_bounds__(64) void k_srt_f_tiles(XfStage st, SrtWs w)
{
    const int b = blockIdx.y;
    const u32 n = st.len[b];
    const u32 tbase = blockIdx.x * SRT_T;
    if (tbase >= n || st.cap[b] < n + 1024) return;
    const u8* s = st.src[b];
    __shared__ u32 cnt[256], last[256], first[256];
    const int lane = lane_id();
    for (int i = lane; i < 256; i += 64) { cnt[i] = 0; last[i] = 0; first[i] = 0xFFFFFFFFu; }
    __syncthreads();
    const u32 end = (tbase + SRT_T < n) ? tbase + SRT_T : n;
    for (u32 i = tbase + lane; i < end; i += 64) { const u32 c = s[i]; atomicAdd(&cnt[c], 1u); atomicMax(&last[c], i + 1); atomicMin(&first[c], i); }
    __syncthreads();
    const size_t o = ((size_t)b * w.perTiles + blockIdx.x) * 256;
    for (int i = lane; i < 256; i += 64) {
        w.tileCnt[o + i] = cnt[i];
        w.tileLast[o + i] = last[i];
        if (first[i] != 0xFFFFFFFFu) atomicMin(&w.firstOcc[b * 256 + i], first[i]);
    }
}

__global__ __launch_bounds__(256) void k_srt_f_scan1(XfStage st, SrtWs w)
{
    const int b = blockIdx.y;
    const u32 seg = blockIdx.x;
    const u32 n = st.len[b];
    const u32 nT = (st.cap[b] < n + 1024) ? 0u : (n + SRT_T - 1) / SRT_T;
    const u32 t0 = seg * w.segT, t1 = (t0 + w.segT < nT) ? t0 + w.segT : nT;
    u32* pc = w.tileCnt + (size_t)b * w.perTiles * 256 + threadIdx.x;
    u32* pl = w.tileLast + (size_t)b * w.perTiles * 256 + threadIdx.x;
    u32 sum = 0, mx = 0;
    for (u32 t = t0; t < t1; t++) {
        const u32 x = pc[(size_t)t * 256]; pc[(size_t)t * 256] = sum; sum += x;
        const u32 y = pl[(size_t)t * 256]; pl[(size_t)t * 256] = mx; mx = y > mx ? y : mx;
    }
    w.segSum[((size_t)b * w.nSeg + seg) * 256 + threadIdx.x] = sum;
    w.segMax[((size_t)b * w.nSeg + seg) * 256 + threadIdx.x] = mx;
}

// per block: scans over the segments, frequencies, bucket order (SRT.cpp:206-244), bucket starts, header (:246-277)
__global__ __launch_bounds__(256) void k_srt_f_scan2(XfStage st, SrtWs w)
{
    const int b = blockIdx.x;
    const int tid = (int)threadIdx.x;
    const u32 n = st.len[b];
    __shared__ u32 freqs[256];
    __shared__ u32 order[256];
    if (tid == 0) { st.ok[b] = 0; st.newLen[b] = 0; w.hdrLen[b] = 0; }
    if (n == 0) { if (tid == 0) st.ok[b] = 1; return; }
    if (st.cap[b] < n + 1024) return;                          // SRT.hpp:38
    u32* ps = w.segSum + (size_t)b * w.nSeg * 256 + tid;
    u32* pm = w.segMax + (size_t)b * w.nSeg * 256 + tid;
    u32 sum = 0, mx = 0;
    for (u32 g = 0; g < w.nSeg; g++) {
        const u32 x = ps[(size_t)g * 256]; ps[(size_t)g * 256] = sum; sum += x;
        const u32 y = pm[(size_t)g * 256]; pm[(size_t)g * 256] = mx; mx = y > mx ? y : mx;
    }
    freqs[tid] = sum;
    __syncthreads();
    // order index: number of present symbols with a larger frequency, or an equal one and a smaller value
    u32 r = 0;
    const u32 fc = freqs[tid];
    for (int q = 0; q < 256; q++) { const u32 fq = freqs[q]; r += (fq != 0 && (fq > fc || (fq == fc && q < tid))) ? 1u : 0u; }
    order[tid] = fc ? r : 0xFFFFFFFFu;
    __syncthreads();
    u32 start = 0;
    if (fc) for (int q = 0; q < 256; q++) start += (order[q] < r) ? freqs[q] : 0u;
    w.bstart[b * 256 + tid] = start;
    if (tid == 0) {
        u8* out = st.dst[b];
        u32 hdr = 0;
        for (int i = 0; i < 256; i++) {
            u32 f = freqs[i];
            for (int k = 0; k < 4 && f >= 128; k++) { out[hdr++] = (u8)(0x80 | f); f >>= 7; }
            out[hdr++] = (u8)f;
        }
        w.hdrLen[b] = hdr;
        st.ok[b] = 1;
        st.newLen[b] = hdr + n;
    }
}

__global__ __launch_bounds__(64) void k_srt_f_rank(XfStage st, SrtWs w)
{
    const int b = blockIdx.y;
    const u32 n = st.len[b];
    const u32 tbase = blockIdx.x * SRT_T;
    if (tbase >= n) return;
    const u32 hdr = w.hdrLen[b];
    if (hdr == 0) return;
    const int lane = lane_id();
    __shared__ u32 keys[256];
    __shared__ u32 listw[64];
    __shared__ u32 base[256];
    __shared__ u32 cntL[256];
    __shared__ u32 tile[SRT_T / 4];
    const u8* src = st.src[b] + tbase;
    const u32 cnt = (n - tbase < SRT_T) ? n - tbase : SRT_T;
    const size_t ot = ((size_t)b * w.perTiles + blockIdx.x) * 256, og = ((size_t)b * w.nSeg + blockIdx.x / w.segT) * 256;
    for (int c = lane; c < 256; c += 64) {
        const u32 lastBefore = max(w.tileLast[ot + c], w.segMax[og + c]);
        const u32 fo = w.firstOcc[b * 256 + c];
        // seen symbols by last occurrence (most recent first), then the ones still to come by first occurrence, absent ones last
        keys[c] = lastBefore ? (0x80000000u | lastBefore) : (fo == 0xFFFFFFFFu ? 0u : 0x7FFFFFFFu - fo);
        base[c] = hdr + w.bstart[b * 256 + c] + w.segSum[og + c] + w.tileCnt[ot + c];
        cntL[c] = 0;
    }
    {
        u8* tb = reinterpret_cast<u8*>(tile);
        for (u32 i = (u32)lane; i < cnt; i += 64) tb[i] = src[i];
    }
    __syncthreads();
    u8* listb = reinterpret_cast<u8*>(listw);
    for (int c = lane; c < 256; c += 64) {
        const u32 kc = keys[c];
        u32 r = 0;
        for (int q = 0; q < 256; q++) { const u32 kq = keys[q]; r += (kq > kc || (kq == kc && q < c)) ? 1u : 0u; }
        listb[r] = (u8)c;
    }
    __syncthreads();
    u32 lw = listw[lane];
    u32 front = (u32)__builtin_amdgcn_readfirstlane((int)lw) & 0xFF;
    u8* dst = st.dst[b];
    const u8* tb = reinterpret_cast<const u8*>(tile);
    u32 cur = 256, curCnt = 0;
    for (u32 k = 0; k < cnt; k++) {
        const u32 c = tb[k];                                   // (uniform)
        if (c != cur) {                                        // the bucket offset of the symbol at hand lives in a register
            if (cur < 256 && lane == 0) cntL[cur] = curCnt;
            KNZ_WAVE_ORDER();
            cur = c;
            curCnt = cntL[c];
        }
        if (c != front) {
            front = c;
            const u32 x = lw ^ (c * 0x01010101u);
            const u32 hz = (x - 0x01010101u) & ~x & 0x80808080u;
            const u64 m = __ballot(hz != 0);
            const int lane0 = __ffsll((long long)m) - 1;
            const u32 hz0 = (u32)__builtin_amdgcn_readlane((int)hz, lane0);
            const int byteIdx = (__ffs((int)hz0) - 1) >> 3;
            if (lane == 0) dst[base[c] + curCnt] = (u8)(4 * lane0 + byteIdx);
            lw = srt_to_front(lw, lane, lane0, byteIdx, c);
        }
        curCnt++;
    }
}

size_t srt_scratch_u32(int nBlocks, u32 maxLen)
{
    const size_t perTiles = ((size_t)maxLen + SRT_T - 1) / SRT_T;
    const u32 segT = srt_seg_tiles((u32)perTiles);
    const size_t nSeg = (perTiles + segT - 1) / segT;
    return (size_t)nBlocks * (2 * perTiles + 2 * nSeg + 2) * 256 + (size_t)nBlocks + 64;
}

void launch_srt_forward(hipStream_t s, const XfStage& st)
{
    SrtWs w;
    w.perTiles = (int)(((size_t)st.maxLen + SRT_T - 1) / SRT_T);
    if (w.perTiles < 1) w.perTiles = 1;
    w.segT = srt_seg_tiles((u32)w.perTiles);
    w.nSeg = ((u32)w.perTiles + w.segT - 1) / w.segT;
    u32* p = st.scratchU32;
    w.tileCnt = p; p += (size_t)st.nBlocks * w.perTiles * 256;
    w.tileLast = p; p += (size_t)st.nBlocks * w.perTiles * 256;
    w.segSum = p; p += (size_t)st.nBlocks * w.nSeg * 256;
    w.segMax = p; p += (size_t)st.nBlocks * w.nSeg * 256;
    w.firstOcc = p; p += (size_t)st.nBlocks * 256;
    w.bstart = p; p += (size_t)st.nBlocks * 256;
    w.hdrLen = p;
    const u32 per = (st.maxLen + 1024 + 4095) / 4096;
    { KScope ks_("k_srt_zero"); hipLaunchKernelGGL(k_srt_zero, dim3(per < 1024 ? per : 1024, st.nBlocks), dim3(256), 0, s, st); }
    hipMemsetAsync(w.firstOcc, 0xFF, sizeof(u32) * 256 * (size_t)st.nBlocks, s);
    const dim3 gridT((unsigned)w.perTiles, st.nBlocks);
    { KScope ks_("k_srt_f_tiles"); hipLaunchKernelGGL(k_srt_f_tiles, gridT, dim3(64), 0, s, st, w); }
    { KScope ks_("k_srt_f_scan"); hipLaunchKernelGGL(k_srt_f_scan1, dim3(w.nSeg, st.nBlocks), dim3(256), 0, s, st, w);
      hipLaunchKernelGGL(k_srt_f_scan2, dim3(st.nBlocks), dim3(256), 0, s, st, w); }
    { KScope ks_("k_srt_f_rank"); hipLaunchKernelGGL(k_srt_f_rank, gridT, dim3(64), 0, s, st, w); }
}

size_t srt_inverse_scratch_u32(int nBlocks, u32 maxLen)
{
    const size_t wordsPer = (size_t)maxLen / 32 + 2, nWords = wordsPer * (size_t)nBlocks + 1;
    const size_t heads = (size_t)nBlocks * ((size_t)maxLen + 256) + 64;
    return 8 * (size_t)nBlocks + 2 * 256 * (size_t)nBlocks + 64 * (size_t)nBlocks + 3 * nWords + heads + 2 * heads + prims::scan_tmp_bytes(nWords + 16) / 4 + 256;
}

void launch_srt_inverse(hipStream_t s, const XfStage& st)
{
    SrtInv w;
    w.wordsPer = (u32)((size_t)st.maxLen / 32 + 2);
    const size_t nWords = (size_t)w.wordsPer * st.nBlocks + 1;
    const size_t heads = (size_t)st.nBlocks * ((size_t)st.maxLen + 256) + 64;
    u32* p = st.scratchU32;
    w.info = p; p += 8 * (size_t)st.nBlocks;
    w.bstart = p; p += 256 * (size_t)st.nBlocks;
    w.bend = p; p += 256 * (size_t)st.nBlocks;
    w.order = reinterpret_cast<u8*>(p); p += 64 * (size_t)st.nBlocks;
    w.bits = p; p += nWords; w.wcount = p; p += nWords; w.wprefix = p; p += nWords;
    w.headPos = p; p += heads;
    p += ((reinterpret_cast<uintptr_t>(p) & 7) ? 1 : 0);
    w.rec = reinterpret_cast<u64*>(p); p += 2 * heads;
    void* scanTmp = p;
    { KScope ks_("k_srt_i_header"); hipLaunchKernelGGL(k_srt_i_header, dim3(st.nBlocks), dim3(64), 0, s, st, w); }
    { KScope ks_("k_srt_i_flags"); hipLaunchKernelGGL(k_srt_i_flags, dim3((w.wordsPer + 255) / 256, st.nBlocks), dim3(256), 0, s, st, w); }
    hipMemsetAsync(w.wcount + (nWords - 1), 0, 4, s);          // (the scan runs one word past the last block: the total lands there)
    { KScope ks_("k_srt_i_scan"); prims::launch_scan<prims::SCAN_SUM_EXCL>(s, w.wcount, w.wprefix, nWords, nullptr, scanTmp); }
    { KScope ks_("k_srt_i_heads"); hipLaunchKernelGGL(k_srt_i_heads, dim3((unsigned)((nWords + 255) / 256)), dim3(256), 0, s, w, (u32)(nWords - 1)); }
    { KScope ks_("k_srt_i_records"); hipLaunchKernelGGL(k_srt_i_records, dim3(1024, st.nBlocks), dim3(256), 0, s, st, w, st.nBlocks); }
    { KScope ks_("k_srt_inverse"); hipLaunchKernelGGL(k_srt_inverse, dim3(st.nBlocks), dim3(192), 0, s, st, w); }
    { KScope ks_("k_srt_inverse_general"); hipLaunchKernelGGL(k_srt_inverse_general, dim3(st.nBlocks), dim3(192), 0, s, st, w); }
}

}  // namespace knz
