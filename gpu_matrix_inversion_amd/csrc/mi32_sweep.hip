// mi32_sweep.hip -- the unblocked Gauss-Jordan sweep for gfx950.
//
// One fused launch per pivot step r replaces the reference's five
// (maxPivotKernel, finalMaxPivotKernel, pivotElementsKernel, fixRowKernel,
// fixColumnKernel; /root/reference/Matlab/mat_inv_32/mat_inv_32/mat_inv_32.cpp
// :61-106, :112-132, :154-173, :138-150, :13-57, host loop :317-362).
//
// Data layout (differs from the reference on purpose): instead of the N x 2N
// [A|I] panel the working matrix is N x N -- from step r on, column r holds
// the one right-half column of [A|I] that went dense at step r, every other
// right-half column is still an exact unit vector and every other finished
// left-half column an exact unit vector, so nothing else needs storing.  The
// stored values are bit-identical to the augmented form (oracle/gj_oracle.c
// proves it on the CPU); traffic is 8 N^2 bytes per step instead of 16 N^2.
// Like the reference the step is out-of-place (two working copies, ping-pong
// on r % 2, mat_inv_32.cpp:318,353-360): that is what lets a single launch
// read rows r and p everywhere while their slots are being rewritten.
//
// Per step, every workgroup:
//   1. reduces the per-row-tile arg-max records the PREVIOUS launch left for
//      column r (wave64 __shfl_xor max on a 64-bit {|a|, ~row} key, then LDS
//      across the 4 waves)                          -> pivot row p, pivot a[p][r]
//   2. reads its 1024-column slice of the pivot row and normalises it with a
//      true IEEE division (fixRowKernel)            -> kept in registers
//   3. streams its TR rows: slot r <- normalised pivot row, slot p <- old row r
//      (the swap of pivotElementsKernel, done by redirecting the read), every
//      slot but r eliminated with one fma per element, skipping exact-zero
//      multipliers as fixColumnKernel does (mat_inv_32.cpp:28)
//   4. the one thread that owns column r+1 keeps the arg-max of the values it
//      just produced and leaves the record for the next launch.
//
// The whole path is a template on the element type: float is the library's matrix_inv_32, double the
// reference's second precision (matrix_inversion_FP64, matrix_inversion_FP64.cpp:13 -- the same five kernels
// in double, kernels :18-206): same launches, 16 N^2 bytes per step.
#include "mi32_internal.h"
#include "mi32_sweep_common.h"

namespace mi32 {

static constexpr int kSweepThreads = 256;
static constexpr int kColsPerTile = kSweepThreads * 4;

SweepPlan make_sweep_plan(int n)
{
    SweepPlan p;
    p.n = n;
    p.ld = (n + 3) & ~3;
    p.col_tiles = (p.ld + kColsPerTile - 1) / kColsPerTile;
    long want = ((long)n * p.col_tiles) / 1024;  // rows per workgroup that leaves >= ~1024 workgroups
    int tr = 4;
    while (tr < 32 && tr * 2 <= want) tr *= 2;
    p.tr = tr;
    p.row_tiles = (n + tr - 1) / tr;
    return p;
}

// workspace: [W0][W1][keys0][keys1][orig][invp], each region 256-B aligned
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
struct SweepWs {
    void *w0, *w1;   // the two working copies (element type T)
    void *k0, *k1;   // per-row-tile arg-max records (PivotRec<T>)
    int *orig, *invp;
    size_t wstride;  // elements per matrix
};
static size_t sweep_carve(const SweepPlan &p, int batch, void *base, SweepWs *o, size_t elem_bytes)
{
    const size_t wbytes = align256((size_t)p.n * p.ld * elem_bytes);
    const size_t kbytes = align256((size_t)p.row_tiles * 2 * sizeof(unsigned long long) * batch);
    const size_t ibytes = align256((size_t)p.n * sizeof(int) * batch);
    char *c = (char *)base;
    size_t off = 0;
    if (o) { o->w0 = (void *)(c + off); o->wstride = wbytes / elem_bytes; }
    off += wbytes * batch;
    if (o) o->w1 = (void *)(c + off);
    off += wbytes * batch;
    if (o) o->k0 = (void *)(c + off);
    off += kbytes;
    if (o) o->k1 = (void *)(c + off);
    off += kbytes;
    if (o) o->orig = (int *)(c + off);
    off += ibytes;
    if (o) o->invp = (int *)(c + off);
    off += ibytes;
    return off;
}
size_t sweep_workspace_bytes(const SweepPlan &p, int batch, size_t elem_bytes)
{
    return sweep_carve(p, batch, nullptr, nullptr, elem_bytes);
}

// ---- makeAugmentedMatrix counterpart (mat_inv_32.cpp:177-192) ---------------
// Copies A into the first working copy (the identity half is implicit) and
// leaves the arg-max records of column 0 for step 0.
template <typename T, int TR>
__global__ __launch_bounds__(kSweepThreads) void sweep_init_kernel(const T *__restrict__ in, int n, int ld,
                                                                    size_t wstride, T *__restrict__ w0,
                                                                    PivotRec<T> *__restrict__ keys, int npart,
                                                                    int *__restrict__ orig, int *__restrict__ status)
{
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int j4 = (blockIdx.x * kSweepThreads + tid) * 4;
    const int row0 = blockIdx.y * TR;
    const T *a = in + (size_t)b * n * n;
    T *w = w0 + (size_t)b * wstride;
    PivotRec<T> best = PivotRec<T>::none();
    bool nonfinite = false;  // boundary rule: a NaN / inf anywhere in the input is an invalid matrix
    if (j4 < ld) {
#pragma unroll 4
        for (int u = 0; u < TR; ++u) {
            const int i = row0 + u;
            if (i >= n) break;
            Vec4<T> v;
            v.x = (j4 + 0 < n) ? a[(size_t)i * n + j4 + 0] : T(0);
            v.y = (j4 + 1 < n) ? a[(size_t)i * n + j4 + 1] : T(0);
            v.z = (j4 + 2 < n) ? a[(size_t)i * n + j4 + 2] : T(0);
            v.w = (j4 + 3 < n) ? a[(size_t)i * n + j4 + 3] : T(0);
            nonfinite = nonfinite || (v.x - v.x != T(0)) || (v.y - v.y != T(0)) || (v.z - v.z != T(0)) || (v.w - v.w != T(0));
            *reinterpret_cast<Vec4<T> *>(w + (size_t)i * ld + j4) = v;
            if (j4 == 0) {
                const PivotRec<T> k = PivotRec<T>::make(v.x, i);
                best = decltype(best)::best_of(k, best);
            }
        }
    }
    if (blockIdx.x == 0) {
        if (tid == 0) keys[(size_t)b * npart + blockIdx.y] = best;
        for (int u = tid; u < TR; u += kSweepThreads)
            if (row0 + u < n) orig[(size_t)b * n + row0 + u] = row0 + u;
    }
    // status[b] was zeroed (MI32_OK) by the host before this launch; every writer stores the same value
    if (nonfinite && status) status[b] = MI32_SINGULAR;
}

// ---- one pivot step ---------------------------------------------------------
template <typename T>
__device__ __forceinline__ Vec4<T> fma4_neg(T f, Vec4<T> a, Vec4<T> c)
{
    Vec4<T> o;
    o.x = fma_t(-f, a.x, c.x);
    o.y = fma_t(-f, a.y, c.y);
    o.z = fma_t(-f, a.z, c.z);
    o.w = fma_t(-f, a.w, c.w);
    return o;
}
template <typename T>
__device__ __forceinline__ T comp(const Vec4<T> &v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
template <typename T>
__device__ __forceinline__ void set_comp(Vec4<T> &v, int c, T x)
{
    if (c == 0) v.x = x;
    else if (c == 1) v.y = x;
    else if (c == 2) v.z = x;
    else v.w = x;
}

// PIVOT = false: the reference's no-pivot variant (matrix_inversion_no_pivots.cpp:10): the pivot of step r is the
// diagonal entry (findCrr, :41-45) -- no arg-max records are read or written, no row is swapped.
template <typename T, int TR, bool PIVOT>
__global__ __launch_bounds__(kSweepThreads) void gj_sweep_step_kernel(const T *__restrict__ src_all,
                                                                       T *__restrict__ dst_all, int n, int ld,
                                                                       size_t wstride, int r,
                                                                       const PivotRec<T> *__restrict__ keys_in,
                                                                       PivotRec<T> *__restrict__ keys_out,
                                                                       int npart, int *__restrict__ orig,
                                                                       int *__restrict__ status)
{
    __shared__ PivotRec<T> s_key[kSweepThreads / 64];
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const T *src = src_all + (size_t)b * wstride;
    T *dst = dst_all + (size_t)b * wstride;

    // 1. finalMaxPivot: reduce the per-row-tile records of column r
    int p = r;
    if constexpr (PIVOT) {
        PivotRec<T> k = PivotRec<T>::none();
        for (int t = tid; t < npart; t += kSweepThreads) {
            const PivotRec<T> o = keys_in[(size_t)b * npart + t];
            k = decltype(k)::best_of(o, k);
        }
        k = wave_max_rec<T>(k);
        if ((tid & 63) == 0) s_key[tid >> 6] = k;
        __syncthreads();
        {
            const PivotRec<T> a = PivotRec<T>::best_of(s_key[0], s_key[1]);
            const PivotRec<T> c = PivotRec<T>::best_of(s_key[2], s_key[3]);
            k = decltype(k)::best_of(a, c);
        }
        p = k.row(r);
    }
    const T piv = src[(size_t)p * ld + r];  // read before the swap, as mat_inv_32.cpp:70,129-130

    const int j4 = (blockIdx.x * kSweepThreads + tid) * 4;
    const bool active = j4 < ld;
    const int rc = r - j4;                   // component of column r inside this thread's group, if 0..3
    const bool has_r = (rc >= 0 && rc < 4);
    const int nc = r + 1 - j4;               // component of column r+1
    const bool has_next = PIVOT && (nc >= 0 && nc < 4) && (r + 1 < n);
    const int row0 = blockIdx.y * TR;

    // 2. fixRow: the normalised pivot row slice (IEEE division), identity entry -> 1/piv
    const Vec4<T> zero4 = {T(0), T(0), T(0), T(0)};
    Vec4<T> prn = zero4;
    if (active) {
        const Vec4<T> pr = *reinterpret_cast<const Vec4<T> *>(src + (size_t)p * ld + j4);
        prn.x = pr.x / piv;
        prn.y = pr.y / piv;
        prn.z = pr.z / piv;
        prn.w = pr.w / piv;
        if (has_r) set_comp<T>(prn, rc, T(1) / piv);
    }

    // 3. pivotElements + fixColumn over this workgroup's rows
    PivotRec<T> best = PivotRec<T>::none();
#pragma unroll
    for (int u0 = 0; u0 < TR; u0 += 4) {
        T f[4];
        Vec4<T> v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = row0 + u0 + u;
            const int s = (i == p) ? r : i;  // slot p receives the old row r
            const bool ok = (i < n);
            f[u] = ok ? src[(size_t)s * ld + r] : T(0);
            v[u] = (ok && active) ? *reinterpret_cast<const Vec4<T> *>(src + (size_t)s * ld + j4) : zero4;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = row0 + u0 + u;
            if (i >= n || !active) continue;
            Vec4<T> o;
            if (i == r) {
                o = prn;
            } else {
                o = v[u];
                if (has_r) set_comp<T>(o, rc, T(0));  // the implicit identity column's entry in this row
                if (f[u] != T(0)) o = fma4_neg<T>(f[u], prn, o);
            }
            *reinterpret_cast<Vec4<T> *>(dst + (size_t)i * ld + j4) = o;
            if (has_next && i > r) {
                const PivotRec<T> kk = PivotRec<T>::make(comp<T>(o, nc), i);
                best = decltype(best)::best_of(kk, best);
            }
        }
    }
    // 4. maxPivot record of column r+1 for the next launch
    if (has_next && active) keys_out[(size_t)b * npart + blockIdx.y] = best;

    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        if (p != r) {
            int *og = orig + (size_t)b * n;
            const int t = og[r];
            og[r] = og[p];
            og[p] = t;
        }
        if (status && (piv == T(0) || piv - piv != T(0))) status[b] = MI32_SINGULAR;  // zero, NaN or infinite pivot
    }
}

// ---- getInvertedMatrix counterpart (mat_inv_32.cpp:195-203) ------------------
// Working column c holds inverse column orig[c]; gather through the inverse map
// so that the stores are coalesced.
__global__ void invert_perm_kernel(const int *__restrict__ orig, int *__restrict__ invp, int n)
{
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) invp[(size_t)b * n + orig[(size_t)b * n + c]] = c;
}

template <typename T>
__global__ __launch_bounds__(256) void unpermute_columns_kernel(const T *__restrict__ w_all, int ld, size_t wstride,
                                                                 const int *__restrict__ invp, int n,
                                                                 T *__restrict__ out)
{
    const int b = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const T *w = w_all + (size_t)b * wstride;
    T *o = out + (size_t)b * n * n;
    const int c = invp[(size_t)b * n + j];
    const int i0 = blockIdx.y * 16;
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
        const int i = i0 + u;
        if (i < n) o[(size_t)i * n + j] = w[(size_t)i * ld + c];
    }
}

template <typename T, int TR>
static hipError_t sweep_run(const SweepPlan &p, const T *d_a, T *d_inv, int batch, int *d_status, const SweepWs &ws,
                            hipStream_t stream, Profiler *prof, bool pivoting)
{
    const dim3 grid(p.col_tiles, p.row_tiles, batch);
    const dim3 block(kSweepThreads);
    T *w0 = (T *)ws.w0, *w1 = (T *)ws.w1;
    PivotRec<T> *k0 = (PivotRec<T> *)ws.k0, *k1 = (PivotRec<T> *)ws.k1;
    if (d_status) {
        hipError_t e = hipMemsetAsync(d_status, 0, sizeof(int) * (size_t)batch, stream);  // MI32_OK
        if (e != hipSuccess) return e;
    }
    {
        ProfScope ps(prof, KC_INIT, stream);
        hipLaunchKernelGGL((sweep_init_kernel<T, TR>), grid, block, 0, stream, d_a, p.n, p.ld, ws.wstride, w0, k0,
                           p.row_tiles, ws.orig, d_status);
    }
    for (int r = 0; r < p.n; ++r) {
        const bool even = (r % 2) == 0;
        ProfScope ps(prof, KC_SWEEP_STEP, stream);
        if (pivoting)
            hipLaunchKernelGGL((gj_sweep_step_kernel<T, TR, true>), grid, block, 0, stream, even ? w0 : w1, even ? w1 : w0,
                               p.n, p.ld, ws.wstride, r, even ? k0 : k1, even ? k1 : k0, p.row_tiles, ws.orig, d_status);
        else
            hipLaunchKernelGGL((gj_sweep_step_kernel<T, TR, false>), grid, block, 0, stream, even ? w0 : w1, even ? w1 : w0,
                               p.n, p.ld, ws.wstride, r, even ? k0 : k1, even ? k1 : k0, p.row_tiles, ws.orig, d_status);
    }
    const T *fin = (p.n % 2 == 0) ? w0 : w1;  // the last-written copy (mat_inv_32.cpp:369-372)
    ProfScope ps(prof, KC_FINISH, stream);
    hipLaunchKernelGGL(invert_perm_kernel, dim3((p.n + 255) / 256, batch), dim3(256), 0, stream, ws.orig, ws.invp,
                       p.n);
    hipLaunchKernelGGL((unpermute_columns_kernel<T>), dim3((p.n + 255) / 256, (p.n + 15) / 16, batch), dim3(256), 0,
                       stream, fin, p.ld, ws.wstride, ws.invp, p.n, d_inv);
    return hipGetLastError();
}

template <typename T>
static hipError_t sweep_invert_t(const SweepPlan &p, const T *d_a, T *d_inv, int batch, int *d_status, void *wsp,
                                 hipStream_t stream, Profiler *prof, bool pivoting)
{
    SweepWs ws;
    sweep_carve(p, batch, wsp, &ws, sizeof(T));
    switch (p.tr) {
        case 4: return sweep_run<T, 4>(p, d_a, d_inv, batch, d_status, ws, stream, prof, pivoting);
        case 8: return sweep_run<T, 8>(p, d_a, d_inv, batch, d_status, ws, stream, prof, pivoting);
        case 16: return sweep_run<T, 16>(p, d_a, d_inv, batch, d_status, ws, stream, prof, pivoting);
        default: return sweep_run<T, 32>(p, d_a, d_inv, batch, d_status, ws, stream, prof, pivoting);
    }
}

hipError_t sweep_invert(const SweepPlan &p, const float *d_a, float *d_inv, int batch, int *d_status, void *wsp,
                        hipStream_t stream, Profiler *prof, bool pivoting)
{
    return sweep_invert_t<float>(p, d_a, d_inv, batch, d_status, wsp, stream, prof, pivoting);
}
hipError_t sweep_invert_f64(const SweepPlan &p, const double *d_a, double *d_inv, int batch, int *d_status, void *wsp,
                            hipStream_t stream, Profiler *prof, bool pivoting)
{
    return sweep_invert_t<double>(p, d_a, d_inv, batch, d_status, wsp, stream, prof, pivoting);
}

}  // namespace mi32
