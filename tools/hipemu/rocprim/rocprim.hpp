// TEST INFRASTRUCTURE ONLY -- CPU stand-ins for the few rocPRIM device primitives the kernels' host code calls
// (same signatures: a null temporary-storage pointer asks for the size).
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>
#include "../hip/hip_runtime.h"

namespace rocprim {
template <class T> struct maximum { T operator()(const T& a, const T& b) const { return a < b ? b : a; } };
template <class T> struct minimum { T operator()(const T& a, const T& b) const { return b < a ? b : a; } };
template <class T> struct plus { T operator()(const T& a, const T& b) const { return a + b; } };

template <class K, class V>
hipError_t radix_sort_pairs(void* tmp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, size_t n, unsigned beginBit, unsigned endBit, hipStream_t)
{
    if (!tmp) { bytes = 16; return hipSuccess; }
    std::vector<size_t> idx(n);
    std::iota(idx.begin(), idx.end(), (size_t)0);
    const unsigned nb = endBit - beginBit;
    const K mask = (nb >= 8 * sizeof(K)) ? (K)~(K)0 : (K)((((K)1) << nb) - 1);
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ((kin[a] >> beginBit) & mask) < ((kin[b] >> beginBit) & mask); });
    for (size_t i = 0; i < n; i++) { kout[i] = kin[idx[i]]; vout[i] = vin[idx[i]]; }
    return hipSuccess;
}
template <class K>
hipError_t radix_sort_keys(void* tmp, size_t& bytes, const K* kin, K* kout, size_t n, unsigned beginBit, unsigned endBit, hipStream_t)
{
    if (!tmp) { bytes = 16; return hipSuccess; }
    std::vector<K> v(kin, kin + n);
    const unsigned nb = endBit - beginBit;
    const K mask = (nb >= 8 * sizeof(K)) ? (K)~(K)0 : (K)((((K)1) << nb) - 1);
    std::stable_sort(v.begin(), v.end(), [&](const K& a, const K& b) { return ((a >> beginBit) & mask) < ((b >> beginBit) & mask); });
    for (size_t i = 0; i < n; i++) kout[i] = v[i];
    return hipSuccess;
}
template <class T, class Op>
hipError_t inclusive_scan(void* tmp, size_t& bytes, const T* in, T* out, size_t n, Op op, hipStream_t)
{
    if (!tmp) { bytes = 16; return hipSuccess; }
    T run = T();
    for (size_t i = 0; i < n; i++) { run = i ? op(run, in[i]) : in[i]; out[i] = run; }
    return hipSuccess;
}
template <class T, class Op>
hipError_t exclusive_scan(void* tmp, size_t& bytes, const T* in, T* out, T init, size_t n, Op op, hipStream_t)
{
    if (!tmp) { bytes = 16; return hipSuccess; }
    T run = init;
    for (size_t i = 0; i < n; i++) { const T x = in[i]; out[i] = run; run = op(run, x); }
    return hipSuccess;
}
}  // namespace rocprim
