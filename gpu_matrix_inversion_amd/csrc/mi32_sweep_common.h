// mi32_sweep_common.h -- element-type helpers shared by the unblocked sweep (mi32_sweep.hip) and the fp64 blocked
// path (mi32_blocked64.hip): 4-element row vectors, arg-max records, fused multiply-add by type.
#pragma once
#include "mi32_internal.h"

namespace mi32 {

// four consecutive elements of a row: one 16-byte (float) or two 16-byte (double) accesses
template <typename T>
struct __attribute__((aligned(4 * sizeof(T)))) Vec4 {
    T x, y, z, w;
};

// arg-max record of one candidate: {bits(|a|), ~row} under an unsigned lexicographic order (largest |a|, lowest
// row among equal maxima, NaN never wins; see pivot_key in mi32_internal.h).  float: one 64-bit word.
// double: |a| alone has 63 significant bits, so the record is two words.
template <typename T>
struct PivotRec;
template <>
struct PivotRec<float> {
    unsigned long long k;
    __device__ static PivotRec none() { return PivotRec{0ull}; }
    __device__ static PivotRec make(float a, int row) { return PivotRec{pivot_key(a, row)}; }
    __device__ bool beats(const PivotRec &o) const { return k > o.k; }
    // the better of two records
    __device__ static PivotRec best_of(const PivotRec &a, const PivotRec &b) { return PivotRec{a.k > b.k ? a.k : b.k}; }
    __device__ int row(int fallback) const { return pivot_key_row(k, fallback); }
    __device__ PivotRec shfl_xor(int off) const { return PivotRec{(unsigned long long)__shfl_xor(k, off, 64)}; }
};
template <>
struct PivotRec<double> {
    unsigned long long v, nrow;  // bits(|a|), ~row; {0, 0} = no candidate
    __device__ static PivotRec none() { return PivotRec{0ull, 0ull}; }
    __device__ static PivotRec make(double a, int row)
    {
        const double m = __builtin_fabs(a);
        if (!(m == m)) return none();
        return PivotRec{(unsigned long long)__double_as_longlong(m), (unsigned long long)(0xFFFFFFFFu - (unsigned)row)};
    }
    __device__ bool beats(const PivotRec &o) const { return v > o.v || (v == o.v && nrow > o.nrow); }
    // the better of two records, word by word: `cond ? a : b` on the two-word struct goes through scratch memory
    // with a run-time index in hipcc's code (measured: +4 us per pivot step of the fp64 kernels)
    __device__ static PivotRec best_of(const PivotRec &a, const PivotRec &b)
    {
        const bool t = a.beats(b);
        return PivotRec{t ? a.v : b.v, t ? a.nrow : b.nrow};
    }
    __device__ int row(int fallback) const
    {
        return (v == 0ull && nrow == 0ull) ? fallback : (int)(0xFFFFFFFFu - (unsigned)nrow);
    }
    __device__ PivotRec shfl_xor(int off) const
    {
        return PivotRec{(unsigned long long)__shfl_xor(v, off, 64), (unsigned long long)__shfl_xor(nrow, off, 64)};
    }
};
template <typename T>
__device__ __forceinline__ PivotRec<T> wave_max_rec(PivotRec<T> k)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const PivotRec<T> o = k.shfl_xor(off);
        k = PivotRec<T>::best_of(o, k);
    }
    return k;
}
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }


}  // namespace mi32
