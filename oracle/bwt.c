/* TEST INFRASTRUCTURE ONLY (see knz_oracle.h).
 * BWT as defined by the reference (transform/BWT.cpp:92-134 -> DivSufSort::computeBWT,
 * transform/DivSufSort.cpp:171-295; definition in transform/BWT.hpp:40-61):
 *   SA = suffix array with "proper prefix sorts first";
 *   out[0] = in[n-1]; then in[SA[r]-1] for every rank r with SA[r] != 0, in rank order;
 *   primaryIndex(0) = rank(suffix 0) + 1; with 8 chunks (n >= 256), step = ceil(n/8) and
 *   primaryIndex(k) = rank(suffix k*step) + 1.
 * The suffix array itself is unique, so any correct construction matches divsufsort; this
 * restatement uses SA-IS (Nong, Zhang, Chan 2009) in its textbook form.
 * Inverse (transform/BWT.cpp:136-166, :169-292 mergeTPSI, :295-492 biPSIv2): both reference
 * algorithms walk the same psi permutation from the 8 primary indexes; restated once.
 */
#include "knz_oracle.h"
#include <stdlib.h>
#include <string.h>

int knzo_bwt_chunks(int n) { return (n < 256) ? 1 : 8; }   /* BWT.hpp:130-133 */

/* ---- SA-IS over an int string whose last symbol is a unique smallest sentinel ---- */
#define TGET(i) ((t[(i) >> 3] >> ((i) & 7)) & 1)
#define TSET(i, b) do { if (b) t[(i) >> 3] |= (uint8_t)(1u << ((i) & 7)); else t[(i) >> 3] &= (uint8_t)~(1u << ((i) & 7)); } while (0)
#define ISLMS(i) ((i) > 0 && TGET(i) && !TGET((i) - 1))

static void get_buckets(const int* s, int* bkt, int n, int K, int end)
{
    int sum = 0;
    for (int i = 0; i <= K; i++) bkt[i] = 0;
    for (int i = 0; i < n; i++) bkt[s[i]]++;
    for (int i = 0; i <= K; i++) { sum += bkt[i]; bkt[i] = end ? sum : sum - bkt[i]; }
}

static void induce_l(const uint8_t* t, int* SA, const int* s, int* bkt, int n, int K)
{
    get_buckets(s, bkt, n, K, 0);
    for (int i = 0; i < n; i++) {
        const int j = SA[i] - 1;
        if (j >= 0 && !TGET(j)) SA[bkt[s[j]]++] = j;
    }
}

static void induce_s(const uint8_t* t, int* SA, const int* s, int* bkt, int n, int K)
{
    get_buckets(s, bkt, n, K, 1);
    for (int i = n - 1; i >= 0; i--) {
        const int j = SA[i] - 1;
        if (j >= 0 && TGET(j)) SA[--bkt[s[j]]] = j;
    }
}

static void sa_is(const int* s, int* SA, int n, int K)
{
    uint8_t* t = (uint8_t*)calloc((size_t)n / 8 + 1, 1);
    int* bkt = (int*)malloc(sizeof(int) * ((size_t)K + 1));
    int i, j;
    TSET(n - 2, 0); TSET(n - 1, 1);
    for (i = n - 3; i >= 0; i--)
        TSET(i, (s[i] < s[i + 1] || (s[i] == s[i + 1] && TGET(i + 1) == 1)) ? 1 : 0);

    get_buckets(s, bkt, n, K, 1);
    for (i = 0; i < n; i++) SA[i] = -1;
    for (i = 1; i < n; i++) if (ISLMS(i)) SA[--bkt[s[i]]] = i;
    induce_l(t, SA, s, bkt, n, K);
    induce_s(t, SA, s, bkt, n, K);

    int n1 = 0;
    for (i = 0; i < n; i++) if (ISLMS(SA[i])) SA[n1++] = SA[i];
    for (i = n1; i < n; i++) SA[i] = -1;
    int name = 0, prev = -1;
    for (i = 0; i < n1; i++) {
        int pos = SA[i], diff = 0;
        for (int d = 0; d < n; d++) {
            if (prev == -1 || s[pos + d] != s[prev + d] || TGET(pos + d) != TGET(prev + d)) { diff = 1; break; }
            else if (d > 0 && (ISLMS(pos + d) || ISLMS(prev + d))) break;
        }
        if (diff) { name++; prev = pos; }
        pos = pos / 2;
        SA[n1 + pos] = name - 1;
    }
    for (i = n - 1, j = n - 1; i >= n1; i--) if (SA[i] >= 0) SA[j--] = SA[i];

    int* SA1 = SA;
    int* s1 = SA + n - n1;
    if (name < n1) sa_is(s1, SA1, n1, name - 1);
    else for (i = 0; i < n1; i++) SA1[s1[i]] = i;

    get_buckets(s, bkt, n, K, 1);
    for (i = 1, j = 0; i < n; i++) if (ISLMS(i)) s1[j++] = i;
    for (i = 0; i < n1; i++) SA1[i] = s1[SA1[i]];
    for (i = n1; i < n; i++) SA[i] = -1;
    for (i = n1 - 1; i >= 0; i--) { j = SA[i]; SA[i] = -1; SA[--bkt[s[j]]] = j; }
    induce_l(t, SA, s, bkt, n, K);
    induce_s(t, SA, s, bkt, n, K);
    free(bkt); free(t);
}

int knzo_bwt_forward_raw(const uint8_t* src, int n, uint8_t* dst, int* primary)
{
    for (int i = 0; i < 8; i++) primary[i] = 0;
    if (n <= 0) return n == 0;
    if (n == 1) { dst[0] = src[0]; return 1; }            /* BWT.cpp:109-115 */
    int* s = (int*)malloc(sizeof(int) * ((size_t)n + 1));
    int* SA = (int*)malloc(sizeof(int) * ((size_t)n + 1));
    for (int i = 0; i < n; i++) s[i] = (int)src[i] + 1;
    s[n] = 0;
    sa_is(s, SA, n + 1, 256);
    /* SA[0] == n (sentinel); ranks of real suffixes are SA[1..n] */
    const int chunks = knzo_bwt_chunks(n);
    const int st = n / chunks;
    const int step = (chunks * st == n) ? st : st + 1;
    int o = 0;
    dst[o++] = src[n - 1];
    for (int r = 0; r < n; r++) {
        const int p = SA[r + 1];
        if (p != 0) dst[o++] = src[p - 1];
        if ((p % step) == 0 && (p / step) < chunks) primary[p / step] = r + 1;
    }
    free(s); free(SA);
    return 1;
}

int knzo_bwt_inverse_raw(const uint8_t* src, int count, uint8_t* dst, const int* primary)
{
    if (count <= 0) return count == 0;
    if (count == 1) { dst[0] = src[0]; return 1; }
    const int pIdx = primary[0];
    if (pIdx <= 0 || pIdx > count) return 0;
    const int chunks = knzo_bwt_chunks(count);
    if (count > 2 * 1024 * 1024) {                         /* biPSIv2 validates all 8 up front, :313-318 */
        for (int i = 1; i < 8; i++)
            if (primary[i] <= 0 || primary[i] > count) return 0;
    }
    /* psi: sorted position -> (next position, byte); BWT.cpp:188-216 */
    uint32_t* nxt = (uint32_t*)calloc((size_t)count, sizeof(uint32_t));
    uint8_t* val = (uint8_t*)malloc((size_t)count);
    uint32_t buckets[256];
    memset(buckets, 0, sizeof(buckets));
    for (int i = 0; i < count; i++) buckets[src[i]]++;
    for (int i = 0, sum = 0; i < 256; i++) { const int tmp = (int)buckets[i]; buckets[i] = (uint32_t)sum; sum += tmp; }
    nxt[buckets[src[0]]] = 0; val[buckets[src[0]]] = src[0]; buckets[src[0]]++;
    for (int i = 1; i < pIdx; i++) {
        const uint8_t v = src[i];
        nxt[buckets[v]] = (uint32_t)(i - 1); val[buckets[v]] = v; buckets[v]++;
    }
    for (int i = pIdx; i < count; i++) {
        const uint8_t v = src[i];
        nxt[buckets[v]] = (uint32_t)i; val[buckets[v]] = v; buckets[v]++;
    }
    int ok = 1;
    if (chunks != 8) {
        uint32_t t = (uint32_t)(pIdx - 1);
        for (int n = 0; n < count; n++) { dst[n] = val[t]; t = nxt[t]; }
    } else {
        const int ckSize = ((count & 7) == 0) ? count >> 3 : (count >> 3) + 1;
        for (int k = 0; k < 8 && ok; k++) {
            int t = primary[k] - 1;
            if (t < 0 || t >= count) { ok = 0; break; }
            const int start = k * ckSize;
            const int len = (k < 7) ? ckSize : count - 7 * ckSize;
            uint32_t tt = (uint32_t)t;
            for (int n = 0; n < len; n++) { dst[start + n] = val[tt]; tt = nxt[tt]; }
        }
    }
    free(nxt); free(val);
    return ok;
}
