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
#include "mi32_internal.h"

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
    float *w0, *w1;
    unsigned long long *k0, *k1;
    int *orig, *invp;
    size_t wstride;  // floats per matrix
};
static size_t sweep_carve(const SweepPlan &p, int batch, void *base, SweepWs *o)
{
    const size_t wbytes = align256((size_t)p.n * p.ld * sizeof(float));
    const size_t kbytes = align256((size_t)p.row_tiles * sizeof(unsigned long long) * batch);
    const size_t ibytes = align256((size_t)p.n * sizeof(int) * batch);
    char *c = (char *)base;
    size_t off = 0;
    if (o) { o->w0 = (float *)(c + off); o->wstride = wbytes / sizeof(float); }
    off += wbytes * batch;
    if (o) o->w1 = (float *)(c + off);
    off += wbytes * batch;
    if (o) o->k0 = (unsigned long long *)(c + off);
    off += kbytes;
    if (o) o->k1 = (unsigned long long *)(c + off);
    off += kbytes;
    if (o) o->orig = (int *)(c + off);
    off += ibytes;
    if (o) o->invp = (int *)(c + off);
    off += ibytes;
    return off;
}
size_t sweep_workspace_bytes(const SweepPlan &p, int batch) { return sweep_carve(p, batch, nullptr, nullptr); }

// ---- makeAugmentedMatrix counterpart (mat_inv_32.cpp:177-192) ---------------
// Copies A into the first working copy (the identity half is implicit) and
// leaves the arg-max records of column 0 for step 0.
template <int TR>
__global__ __launch_bounds__(kSweepThreads) void sweep_init_kernel(const float *__restrict__ in, int n, int ld,
                                                                    size_t wstride, float *__restrict__ w0,
                                                                    unsigned long long *__restrict__ keys, int npart,
                                                                    int *__restrict__ orig, int *__restrict__ status)
{
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int j4 = (blockIdx.x * kSweepThreads + tid) * 4;
    const int row0 = blockIdx.y * TR;
    const float *a = in + (size_t)b * n * n;
    float *w = w0 + (size_t)b * wstride;
    unsigned long long best = 0ull;
    if (j4 < ld) {
#pragma unroll 4
        for (int u = 0; u < TR; ++u) {
            const int i = row0 + u;
            if (i >= n) break;
            float4 v;
            v.x = (j4 + 0 < n) ? a[(size_t)i * n + j4 + 0] : 0.0f;
            v.y = (j4 + 1 < n) ? a[(size_t)i * n + j4 + 1] : 0.0f;
            v.z = (j4 + 2 < n) ? a[(size_t)i * n + j4 + 2] : 0.0f;
            v.w = (j4 + 3 < n) ? a[(size_t)i * n + j4 + 3] : 0.0f;
            *reinterpret_cast<float4 *>(w + (size_t)i * ld + j4) = v;
            if (j4 == 0) {
                const unsigned long long k = pivot_key(v.x, i);
                best = k > best ? k : best;
            }
        }
    }
    if (blockIdx.x == 0) {
        if (tid == 0) keys[(size_t)b * npart + blockIdx.y] = best;
        for (int u = tid; u < TR; u += kSweepThreads)
            if (row0 + u < n) orig[(size_t)b * n + row0 + u] = row0 + u;
        if (blockIdx.y == 0 && tid == 0 && status) status[b] = MI32_OK;
    }
}

// ---- one pivot step ---------------------------------------------------------
__device__ __forceinline__ float4 fma4_neg(float f, float4 a, float4 c)
{
    float4 o;
    o.x = __builtin_fmaf(-f, a.x, c.x);
    o.y = __builtin_fmaf(-f, a.y, c.y);
    o.z = __builtin_fmaf(-f, a.z, c.z);
    o.w = __builtin_fmaf(-f, a.w, c.w);
    return o;
}
__device__ __forceinline__ float comp(const float4 &v, int c) { return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
__device__ __forceinline__ void set_comp(float4 &v, int c, float x)
{
    if (c == 0) v.x = x;
    else if (c == 1) v.y = x;
    else if (c == 2) v.z = x;
    else v.w = x;
}

template <int TR>
__global__ __launch_bounds__(kSweepThreads) void gj_sweep_step_kernel(const float *__restrict__ src_all,
                                                                       float *__restrict__ dst_all, int n, int ld,
                                                                       size_t wstride, int r,
                                                                       const unsigned long long *__restrict__ keys_in,
                                                                       unsigned long long *__restrict__ keys_out,
                                                                       int npart, int *__restrict__ orig,
                                                                       int *__restrict__ status)
{
    __shared__ unsigned long long s_key[kSweepThreads / 64];
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const float *src = src_all + (size_t)b * wstride;
    float *dst = dst_all + (size_t)b * wstride;

    // 1. finalMaxPivot: reduce the per-row-tile records of column r
    unsigned long long k = 0ull;
    for (int t = tid; t < npart; t += kSweepThreads) {
        const unsigned long long o = keys_in[(size_t)b * npart + t];
        k = o > k ? o : k;
    }
    k = wave_max_u64(k);
    if ((tid & 63) == 0) s_key[tid >> 6] = k;
    __syncthreads();
    {
        const unsigned long long a = s_key[0] > s_key[1] ? s_key[0] : s_key[1];
        const unsigned long long c = s_key[2] > s_key[3] ? s_key[2] : s_key[3];
        k = a > c ? a : c;
    }
    const int p = pivot_key_row(k, r);
    const float piv = src[(size_t)p * ld + r];  // read before the swap, as mat_inv_32.cpp:70,129-130

    const int j4 = (blockIdx.x * kSweepThreads + tid) * 4;
    const bool active = j4 < ld;
    const int rc = r - j4;                   // component of column r inside this thread's group, if 0..3
    const bool has_r = (rc >= 0 && rc < 4);
    const int nc = r + 1 - j4;               // component of column r+1
    const bool has_next = (nc >= 0 && nc < 4) && (r + 1 < n);
    const int row0 = blockIdx.y * TR;

    // 2. fixRow: the normalised pivot row slice (IEEE division), identity entry -> 1/piv
    float4 prn = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        const float4 pr = *reinterpret_cast<const float4 *>(src + (size_t)p * ld + j4);
        prn.x = pr.x / piv;
        prn.y = pr.y / piv;
        prn.z = pr.z / piv;
        prn.w = pr.w / piv;
        if (has_r) set_comp(prn, rc, 1.0f / piv);
    }

    // 3. pivotElements + fixColumn over this workgroup's rows
    unsigned long long best = 0ull;
#pragma unroll
    for (int u0 = 0; u0 < TR; u0 += 4) {
        float f[4];
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = row0 + u0 + u;
            const int s = (i == p) ? r : i;  // slot p receives the old row r
            const bool ok = (i < n);
            f[u] = ok ? src[(size_t)s * ld + r] : 0.0f;
            v[u] = (ok && active) ? *reinterpret_cast<const float4 *>(src + (size_t)s * ld + j4)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = row0 + u0 + u;
            if (i >= n || !active) continue;
            float4 o;
            if (i == r) {
                o = prn;
            } else {
                o = v[u];
                if (has_r) set_comp(o, rc, 0.0f);  // the implicit identity column's entry in this row
                if (f[u] != 0.0f) o = fma4_neg(f[u], prn, o);
            }
            *reinterpret_cast<float4 *>(dst + (size_t)i * ld + j4) = o;
            if (has_next && i > r) {
                const unsigned long long kk = pivot_key(comp(o, nc), i);
                best = kk > best ? kk : best;
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
        if (status && (piv == 0.0f || piv != piv)) status[b] = MI32_SINGULAR;
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

__global__ __launch_bounds__(256) void unpermute_columns_kernel(const float *__restrict__ w_all, int ld, size_t wstride,
                                                                 const int *__restrict__ invp, int n,
                                                                 float *__restrict__ out)
{
    const int b = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float *w = w_all + (size_t)b * wstride;
    float *o = out + (size_t)b * n * n;
    const int c = invp[(size_t)b * n + j];
    const int i0 = blockIdx.y * 16;
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
        const int i = i0 + u;
        if (i < n) o[(size_t)i * n + j] = w[(size_t)i * ld + c];
    }
}

template <int TR>
static hipError_t sweep_run(const SweepPlan &p, const float *d_a, float *d_inv, int batch, int *d_status,
                            const SweepWs &ws, hipStream_t stream, Profiler *prof)
{
    const dim3 grid(p.col_tiles, p.row_tiles, batch);
    const dim3 block(kSweepThreads);
    {
        ProfScope ps(prof, KC_INIT, stream);
        hipLaunchKernelGGL(sweep_init_kernel<TR>, grid, block, 0, stream, d_a, p.n, p.ld, ws.wstride, ws.w0, ws.k0,
                           p.row_tiles, ws.orig, d_status);
    }
    for (int r = 0; r < p.n; ++r) {
        const bool even = (r % 2) == 0;
        ProfScope ps(prof, KC_SWEEP_STEP, stream);
        hipLaunchKernelGGL(gj_sweep_step_kernel<TR>, grid, block, 0, stream, even ? ws.w0 : ws.w1,
                           even ? ws.w1 : ws.w0, p.n, p.ld, ws.wstride, r, even ? ws.k0 : ws.k1,
                           even ? ws.k1 : ws.k0, p.row_tiles, ws.orig, d_status);
    }
    const float *fin = (p.n % 2 == 0) ? ws.w0 : ws.w1;  // the last-written copy (mat_inv_32.cpp:369-372)
    ProfScope ps(prof, KC_FINISH, stream);
    hipLaunchKernelGGL(invert_perm_kernel, dim3((p.n + 255) / 256, batch), dim3(256), 0, stream, ws.orig, ws.invp,
                       p.n);
    hipLaunchKernelGGL(unpermute_columns_kernel, dim3((p.n + 255) / 256, (p.n + 15) / 16, batch), dim3(256), 0,
                       stream, fin, p.ld, ws.wstride, ws.invp, p.n, d_inv);
    return hipGetLastError();
}

hipError_t sweep_invert(const SweepPlan &p, const float *d_a, float *d_inv, int batch, int *d_status, void *wsp,
                        hipStream_t stream, Profiler *prof)
{
    SweepWs ws;
    sweep_carve(p, batch, wsp, &ws);
    switch (p.tr) {
        case 4: return sweep_run<4>(p, d_a, d_inv, batch, d_status, ws, stream, prof);
        case 8: return sweep_run<8>(p, d_a, d_inv, batch, d_status, ws, stream, prof);
        case 16: return sweep_run<16>(p, d_a, d_inv, batch, d_status, ws, stream, prof);
        default: return sweep_run<32>(p, d_a, d_inv, batch, d_status, ws, stream, prof);
    }
}

}  // namespace mi32
