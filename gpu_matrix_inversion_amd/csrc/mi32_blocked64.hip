// mi32_blocked64.hip -- blocked Gauss-Jordan in double with delayed rank-bw updates on the fp64 matrix cores
// (gfx950: v_mfma_f64_16x16x4_f64), behind the reference's fp64 entry point matrix_inversion_FP64
// (/root/reference/matrix_inv_solution/matrix_inversion_solution/matrix_inversion/matrix_inversion_FP64.cpp:13,
// headers.h:9: the five-kernel step of the fp32 library in double, kernels :18-206, loop like mat_inv_32.cpp:317-362).
//
// The unblocked fp64 sweep (mi32_sweep.hip on double) moves 16 N^2 bytes per pivot step: 1.1 TB at N = 4096, 299 ms.
// Here the same steps run on a WINDOW of bw columns at a time:
//
//   for each block K = [c0, c0 + bw) of pivot columns:
//     bw launches of the fused step kernel (maxPivot + finalMaxPivot + pivotElements + fixRow + fixColumn, exactly
//       the arithmetic of the sweep: IEEE division, one fma per element, zero multipliers skipped) restricted to
//       the block's columns: 16 N bw bytes per step instead of 16 N^2; the block's columns ping-pong between the
//       two working copies like the reference's buffers (mat_inv_32.cpp:318,353-360);
//     the row swaps of the block are recorded in a row map and NOT applied to the other columns;
//     one rank-bw update of every other column on the fp64 matrix cores,
//         y[i][j] = (i in K ? 0 : x[map[i]][j]) + sum_k G[i][k] * x[map[c0 + k]][j],
//       G = the block's transformed columns, one k-ascending chain of v_mfma_f64_16x16x4_f64 per element whose
//       C operand is the old value (oracle/gj_oracle.c: gjo_matrix_inv_64_blocked is the CPU mirror, bit for bit);
//     the arg-max records of the next block's first column.
//
// Working matrix: the N x N in-place form padded with an identity block to a multiple of 64 (no bounds checks in the
// tiles: inv(diag(A, I)) = diag(inv(A), I); a real column never takes its pivot from the padding, whose entries
// in real columns are exact zeros).
#include <cstdlib>

#include "mi32_internal.h"
#include "mi32_sweep_common.h"

namespace mi32 {

typedef double b64_d4v __attribute__((ext_vector_type(4)));

static constexpr int kB64Threads = 256;
// rows per workgroup of the step kernel: 8 up to N = 4096 (more, shorter workgroups: the step is a chain of
// dependent round trips), 32 above (every workgroup reduces one arg-max record per row tile: N / TR of them);
// measured N = 4096: 25.1 / 26.3 / 28.3 ms for 8 / 16 / 32, N = 8192: 83.7 / 77.8 / 76.4 ms

Blocked64Plan make_blocked64_plan(int n, int bw)
{
    Blocked64Plan p;
    p.n = n;
    // block width 64, 128 (default) or 256; the padded order is a multiple of it, so that every block is full
    // (a padded pivot step divides an identity row by 1 and eliminates nothing)
    if (bw <= 0) bw = 128;
    bw = (bw >= 256) ? 256 : (bw >= 128) ? 128 : 64;
    while (bw > 64 && bw > ((n + 63) & ~63)) bw >>= 1;
    p.bw = bw;
    p.np = ((n + bw - 1) / bw) * bw;
    p.ld = p.np;
    p.tr = (p.np <= 4096) ? 8 : 32;
    if (const char *e = std::getenv("MI32_B64_TR")) {  // experiments
        const int v = std::atoi(e);
        if (v == 8 || v == 16 || v == 32) p.tr = v;
    }
    p.row_tiles = (p.np + p.tr - 1) / p.tr;
    return p;
}

static inline size_t b64_align256(size_t x) { return (x + 255) & ~(size_t)255; }
struct B64Ws {
    double *w0, *w1;
    PivotRec<double> *k0, *k1;
    int *orig, *invp, *rowmap;
    size_t wstride;
};
static size_t b64_carve(const Blocked64Plan &p, int batch, void *base, B64Ws *o)
{
    const size_t wbytes = b64_align256((size_t)p.np * p.ld * sizeof(double));
    const size_t kbytes = b64_align256((size_t)p.row_tiles * sizeof(PivotRec<double>) * batch);
    const size_t ibytes = b64_align256((size_t)p.np * sizeof(int) * batch);
    char *c = (char *)base;
    size_t off = 0;
    if (o) { o->w0 = (double *)(c + off); o->wstride = wbytes / sizeof(double); }
    off += wbytes * batch;
    if (o) o->w1 = (double *)(c + off);
    off += wbytes * batch;
    if (o) o->k0 = (PivotRec<double> *)(c + off);
    off += kbytes;
    if (o) o->k1 = (PivotRec<double> *)(c + off);
    off += kbytes;
    if (o) o->orig = (int *)(c + off);
    off += ibytes;
    if (o) o->invp = (int *)(c + off);
    off += ibytes;
    if (o) o->rowmap = (int *)(c + off);
    off += ibytes;
    return off;
}
size_t blocked64_workspace_bytes(const Blocked64Plan &p, int batch) { return b64_carve(p, batch, nullptr, nullptr); }

// ---- makeAugmented counterpart: A -> diag(A, I) in the first working copy --------------------
__global__ __launch_bounds__(256) void b64_init_kernel(const double *__restrict__ in, int n, int np, int ld, size_t wstride,
                                                        double *__restrict__ w0, int *__restrict__ orig,
                                                        int *__restrict__ status)
{
    const int b = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i0 = blockIdx.y * 16;
    const double *a = in + (size_t)b * n * n;
    double *w = w0 + (size_t)b * wstride;
    bool nonfinite = false;  // boundary rule: a NaN / inf anywhere in the input is an invalid matrix
    if (j < np) {
#pragma unroll 4
        for (int u = 0; u < 16; ++u) {
            const int i = i0 + u;
            if (i >= np) break;
            double v;
            if (i < n && j < n) v = a[(size_t)i * n + j];
            else v = (i == j) ? 1.0 : 0.0;
            nonfinite = nonfinite || (v - v != 0.0);
            w[(size_t)i * ld + j] = v;
        }
    }
    if (blockIdx.y == 0 && j < np) orig[(size_t)b * np + j] = j;
    if (nonfinite && status) status[b] = MI32_SINGULAR;  // status[b] was zeroed by the host; same value from every writer
}

// ---- arg-max records of column c over the rows >= c (the first column of a block), row map reset ----
__global__ __launch_bounds__(64) void b64_block_prep_kernel(const double *__restrict__ w_all, int np, int ld, size_t wstride,
                                                             int c, PivotRec<double> *__restrict__ keys, int npart,
                                                             int *__restrict__ rowmap, int tr)
{
    const int b = blockIdx.y;
    const int lane = threadIdx.x;
    const double *w = w_all + (size_t)b * wstride;
    const int row0 = blockIdx.x * tr;
    PivotRec<double> best = PivotRec<double>::none();
    if (lane < tr) {
        const int i = row0 + lane;
        if (i < np) {
            rowmap[(size_t)b * np + i] = i;
            if (i >= c) best = PivotRec<double>::make(w[(size_t)i * ld + c], i);
        }
    }
    best = wave_max_rec<double>(best);
    if (lane == 0) keys[(size_t)b * npart + blockIdx.x] = best;
}

// ---- one pivot step on the block's columns [col_lo, col_lo + 4 * TX) ---------------------------
// The fused step of mi32_sweep.hip (same arithmetic, see there) on a window: TX column threads (4 columns each) x
// 256 / TX rows per pass, TR rows per workgroup.
template <int TX, int TR>
__global__ __launch_bounds__(kB64Threads) void b64_panel_step_kernel(
    const double *__restrict__ src_all, double *__restrict__ dst_all, int np, int ld, size_t wstride, int r, int col_lo,
    const PivotRec<double> *__restrict__ keys_in, PivotRec<double> *__restrict__ keys_out, int npart,
    int *__restrict__ orig, int *__restrict__ rowmap, int *__restrict__ status)
{
    constexpr int TY = kB64Threads / TX;  // rows per pass
    __shared__ PivotRec<double> s_key[kB64Threads / 64];
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const double *src = src_all + (size_t)b * wstride;
    double *dst = dst_all + (size_t)b * wstride;

    const int tx = tid % TX, ty = tid / TX;
    const int j4 = col_lo + tx * 4;
    const int rc = r - j4;  // component of column r inside this thread's group, if 0..3
    const bool has_r = (rc >= 0 && rc < 4);
    const int nc = r + 1 - j4;
    const bool has_next = (nc >= 0 && nc < 4);  // column r + 1 belongs to this window and this thread
    const int row0 = blockIdx.y * TR;

    // The step is a chain of dependent global-memory round trips (records -> pivot row -> rows): this workgroup's
    // rows do not depend on the pivot search (but for the one row that receives the old row r), so they are
    // requested FIRST and arrive while the records are reduced and the pivot row is fetched.
    constexpr int NP = (TR + TY - 1) / TY;  // passes over the rows
    double fv[NP];
    Vec4<double> vv[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int i = row0 + ty + q * TY;
        const bool ok = (ty + q * TY < TR) && (i < np);
        fv[q] = ok ? src[(size_t)i * ld + r] : 0.0;
        vv[q] = ok ? *reinterpret_cast<const Vec4<double> *>(src + (size_t)i * ld + j4) : Vec4<double>{0.0, 0.0, 0.0, 0.0};
    }

    // finalMaxPivot: reduce the per-row-tile records of column r
    PivotRec<double> k = PivotRec<double>::none();
    for (int t = tid; t < npart; t += kB64Threads) {
        const PivotRec<double> o = keys_in[(size_t)b * npart + t];
        k = decltype(k)::best_of(o, k);
    }
    k = wave_max_rec<double>(k);
    if ((tid & 63) == 0) s_key[tid >> 6] = k;
    __syncthreads();
    {
        const PivotRec<double> a = PivotRec<double>::best_of(s_key[0], s_key[1]);
        const PivotRec<double> c = PivotRec<double>::best_of(s_key[2], s_key[3]);
        k = decltype(k)::best_of(a, c);
    }
    const int p = k.row(r);
    const double piv = src[(size_t)p * ld + r];  // read before the swap, as mat_inv_32.cpp:70,129-130

    // fixRow: the normalised pivot row slice (IEEE division), identity entry -> 1/piv
    Vec4<double> prn;
    {
        const Vec4<double> pr = *reinterpret_cast<const Vec4<double> *>(src + (size_t)p * ld + j4);
        prn.x = pr.x / piv;
        prn.y = pr.y / piv;
        prn.z = pr.z / piv;
        prn.w = pr.w / piv;
        if (has_r) {
            const double one = 1.0 / piv;
            if (rc == 0) prn.x = one; else if (rc == 1) prn.y = one; else if (rc == 2) prn.z = one; else prn.w = one;
        }
    }
    // pivotElements + fixColumn over this workgroup's rows
    PivotRec<double> best = PivotRec<double>::none();
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int i = row0 + ty + q * TY;
        if (!((ty + q * TY < TR) && (i < np))) continue;
        double f = fv[q];
        Vec4<double> o = vv[q];
        if (i == p && p != r) {  // slot p receives the old row r (the swap of pivotElements, done by redirecting the read)
            f = src[(size_t)r * ld + r];
            o = *reinterpret_cast<const Vec4<double> *>(src + (size_t)r * ld + j4);
        }
        if (i == r) {
            o = prn;
        } else {
            if (has_r) {  // the implicit identity column's entry in this row
                if (rc == 0) o.x = 0.0; else if (rc == 1) o.y = 0.0; else if (rc == 2) o.z = 0.0; else o.w = 0.0;
            }
            if (f != 0.0) {
                o.x = __builtin_fma(-f, prn.x, o.x);
                o.y = __builtin_fma(-f, prn.y, o.y);
                o.z = __builtin_fma(-f, prn.z, o.z);
                o.w = __builtin_fma(-f, prn.w, o.w);
            }
        }
        *reinterpret_cast<Vec4<double> *>(dst + (size_t)i * ld + j4) = o;
        if (has_next && i > r) {
            const double v = (nc == 0) ? o.x : (nc == 1) ? o.y : (nc == 2) ? o.z : o.w;
            const PivotRec<double> kk = PivotRec<double>::make(v, i);
            best = decltype(best)::best_of(kk, best);
        }
    }
    // maxPivot record of column r + 1 for the next launch: the TY threads (one tx) that own that column hold
    // partial results; thread (tx, 0) folds them.  Absent at the last step of a window (column r + 1 then belongs
    // to the next block: b64_block_prep_kernel writes its records after the rank-bw update).
    __shared__ PivotRec<double> s_part[TY];
    if (has_next) s_part[ty] = best;
    __syncthreads();
    if (has_next && ty == 0) {
        PivotRec<double> m = s_part[0];
#pragma unroll
        for (int q = 1; q < TY; ++q) m = PivotRec<double>::best_of(s_part[q], m);
        keys_out[(size_t)b * npart + blockIdx.y] = m;
    }
    if (blockIdx.y == 0 && tid == 0) {
        if (p != r) {
            int *og = orig + (size_t)b * np;
            const int t = og[r]; og[r] = og[p]; og[p] = t;
            int *rm = rowmap + (size_t)b * np;
            const int t2 = rm[r]; rm[r] = rm[p]; rm[p] = t2;
        }
        if (status && (piv == 0.0 || piv - piv != 0.0)) status[b] = MI32_SINGULAR;  // zero, NaN or infinite pivot
    }
}

// ---- the rank-bw update on the fp64 matrix cores --------------------------------------------------
//   dst[i][j] = (i in [c0, c0 + kdim) ? 0 : src[map[i]][j]) + sum_k G[i][k] * src[map[c0 + k]][j]
// for the 64 x 64 tiles of columns outside the block; the tiles inside copy G.  256 threads = 4 waves (2 x 2),
// each 32 x 32 = 2 x 2 tiles of v_mfma_f64_16x16x4_f64.  Operand lane maps (cdna_hip_programming.md, f64 MFMA):
// A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15], C/D[row = (lane >> 4) + 4 * reg][col = lane & 15].
// The accumulators START from the (row-mapped) old values and run through k = 0 .. kdim-1 in order.
__global__ __launch_bounds__(256) void b64_rank_update_kernel(const double *__restrict__ src_all, double *__restrict__ dst_all,
                                                               const double *__restrict__ g_all, int np, int ld,
                                                               size_t wstride, int c0, int kdim,
                                                               const int *__restrict__ map_all)
{
    constexpr int BK = 16, LDT = 64 + 2;
    __shared__ double s_a[BK * LDT];  // G of the tile's rows, [k][row]
    __shared__ double s_b[BK * LDT];  // pivot rows (through the row map) x 64 columns, [k][col]
    __shared__ int s_map[64];
    __shared__ int s_bmap[256];
    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    const double *src = src_all + (size_t)b * wstride;
    double *dst = dst_all + (size_t)b * wstride;
    const double *g = g_all + (size_t)b * wstride;
    const int *map = map_all + (size_t)b * np;

    if (col0 >= c0 && col0 < c0 + kdim) {  // inside the block: those columns are G itself (no swap pending there)
        for (int idx = tid; idx < 64 * 16; idx += 256) {
            const int rr = idx / 16, c4 = (idx % 16) * 4;
            *reinterpret_cast<Vec4<double> *>(dst + (size_t)(row0 + rr) * ld + col0 + c4) =
                *reinterpret_cast<const Vec4<double> *>(g + (size_t)(row0 + rr) * ld + col0 + c4);
        }
        return;
    }
    if (tid < 64) s_map[tid] = map[row0 + tid];
    for (int i = tid; i < kdim; i += 256) s_bmap[i] = map[c0 + i];
    __syncthreads();

    const int l15 = lane & 15, l4 = lane >> 4;
    b64_d4v acc[2][2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int lr = wr * 32 + tm * 16 + l4 + 4 * reg;
                const int grow = row0 + lr;
                const int col = col0 + wc * 32 + tn * 16 + l15;
                acc[tm][tn][reg] = (grow >= c0 && grow < c0 + kdim) ? 0.0 : src[(size_t)s_map[lr] * ld + col];
            }
    for (int kt = 0; kt < kdim; kt += BK) {
        {   // stage A: 64 rows x 16 k of the row-major panel, transposed; 4 threads per row, 4 k each
            const int rr = tid >> 2, k4 = (tid & 3) * 4;
            const Vec4<double> v = *reinterpret_cast<const Vec4<double> *>(g + (size_t)(row0 + rr) * ld + c0 + kt + k4);
            s_a[(k4 + 0) * LDT + rr] = v.x;
            s_a[(k4 + 1) * LDT + rr] = v.y;
            s_a[(k4 + 2) * LDT + rr] = v.z;
            s_a[(k4 + 3) * LDT + rr] = v.w;
        }
        {   // stage B: 16 pivot rows (through the row map) x 64 columns; 16 threads per row, 4 columns each
            const int kk = tid >> 4, c4 = (tid & 15) * 4;
            const Vec4<double> v = *reinterpret_cast<const Vec4<double> *>(src + (size_t)s_bmap[kt + kk] * ld + col0 + c4);
            s_b[kk * LDT + c4 + 0] = v.x;
            s_b[kk * LDT + c4 + 1] = v.y;
            s_b[kk * LDT + c4 + 2] = v.z;
            s_b[kk * LDT + c4 + 3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            double af[2], bf[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                af[q] = s_a[(kk + l4) * LDT + wr * 32 + q * 16 + l15];
                bf[q] = s_b[(kk + l4) * LDT + wc * 32 + q * 16 + l15];
            }
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[tm], bf[tn], acc[tm][tn], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int grow = row0 + wr * 32 + tm * 16 + l4 + 4 * reg;
                const int col = col0 + wc * 32 + tn * 16 + l15;
                dst[(size_t)grow * ld + col] = acc[tm][tn][reg];
            }
}

// ---- getInvertedMatrix counterpart --------------------------------------------------------------
__global__ void b64_invert_perm_kernel(const int *__restrict__ orig, int *__restrict__ invp, int np)
{
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < np) invp[(size_t)b * np + orig[(size_t)b * np + c]] = c;
}
__global__ __launch_bounds__(256) void b64_unpermute_kernel(const double *__restrict__ w_all, int ld, int np, size_t wstride,
                                                             const int *__restrict__ invp, int n, double *__restrict__ out)
{
    const int b = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const double *w = w_all + (size_t)b * wstride;
    double *o = out + (size_t)b * n * n;
    const int c = invp[(size_t)b * np + j];
    const int i0 = blockIdx.y * 16;
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
        const int i = i0 + u;
        if (i < n) o[(size_t)i * n + j] = w[(size_t)i * ld + c];
    }
}

template <int TX>
static void b64_launch_step(const dim3 &grid, hipStream_t stream, const double *x, double *y, const Blocked64Plan &p,
                            size_t wstride, int r, int col_lo, const PivotRec<double> *kin, PivotRec<double> *kout, int *orig,
                            int *rowmap, int *status)
{
    if (p.tr == 16)
        hipLaunchKernelGGL((b64_panel_step_kernel<TX, 16>), grid, dim3(kB64Threads), 0, stream, x, y, p.np, p.ld, wstride, r,
                           col_lo, kin, kout, p.row_tiles, orig, rowmap, status);
    else if (p.tr == 8)
        hipLaunchKernelGGL((b64_panel_step_kernel<TX, 8>), grid, dim3(kB64Threads), 0, stream, x, y, p.np, p.ld, wstride, r,
                           col_lo, kin, kout, p.row_tiles, orig, rowmap, status);
    else
        hipLaunchKernelGGL((b64_panel_step_kernel<TX, 32>), grid, dim3(kB64Threads), 0, stream, x, y, p.np, p.ld, wstride, r,
                           col_lo, kin, kout, p.row_tiles, orig, rowmap, status);
}

hipError_t blocked64_invert(const Blocked64Plan &p, const double *d_a, double *d_inv, int batch, int *d_status, void *wsp,
                            hipStream_t stream, Profiler *prof)
{
    B64Ws ws;
    b64_carve(p, batch, wsp, &ws);
    const int np = p.np;
    hipError_t e;
    if (d_status) {
        if ((e = hipMemsetAsync(d_status, 0, sizeof(int) * (size_t)batch, stream)) != hipSuccess) return e;
    }
    {
        ProfScope ps(prof, KC_INIT, stream);
        hipLaunchKernelGGL(b64_init_kernel, dim3((np + 255) / 256, (np + 15) / 16, batch), dim3(256), 0, stream, d_a, p.n,
                           np, p.ld, ws.wstride, ws.w0, ws.orig, d_status);
    }
    double *cur = ws.w0, *oth = ws.w1;
    PivotRec<double> *kin = ws.k0, *kout = ws.k1;
    const dim3 step_grid(1, p.row_tiles, batch);
    for (int c0 = 0; c0 < np; c0 += p.bw) {
        const int kb = p.bw;  // np is a multiple of bw
        {   // arg-max records of the block's first column (its values are final only now), row map = identity
            ProfScope ps(prof, KC_PANEL, stream);
            hipLaunchKernelGGL(b64_block_prep_kernel, dim3(p.row_tiles, batch), dim3(64), 0, stream, cur, np, p.ld,
                               ws.wstride, c0, kin, p.row_tiles, ws.rowmap, p.tr);
        }
        double *x = cur, *y = oth;  // the block's columns alternate between the two copies
        for (int s = 0; s < kb; ++s) {
            ProfScope ps(prof, KC_PANEL, stream);
            const int r = c0 + s;
            switch (kb) {
                case 64: b64_launch_step<16>(step_grid, stream, x, y, p, ws.wstride, r, c0, kin, kout, ws.orig, ws.rowmap, d_status); break;
                case 128: b64_launch_step<32>(step_grid, stream, x, y, p, ws.wstride, r, c0, kin, kout, ws.orig, ws.rowmap, d_status); break;
                default: b64_launch_step<64>(step_grid, stream, x, y, p, ws.wstride, r, c0, kin, kout, ws.orig, ws.rowmap, d_status); break;
            }
            double *t = x; x = y; y = t;
            PivotRec<double> *tk = kin; kin = kout; kout = tk;
        }
        // kb is even: the block's columns are back in `cur` (x == cur); every other column is valid in `cur` too,
        // in the row order of the block's start
        if (kb < np) {
            ProfScope ps(prof, KC_UPDATE_OUT, stream);
            hipLaunchKernelGGL(b64_rank_update_kernel, dim3(np / 64, np / 64, batch), dim3(256), 0, stream, cur, oth, x, np,
                               p.ld, ws.wstride, c0, kb, ws.rowmap);
            double *t = cur; cur = oth; oth = t;
        }
    }
    ProfScope ps(prof, KC_FINISH, stream);
    hipLaunchKernelGGL(b64_invert_perm_kernel, dim3((np + 255) / 256, batch), dim3(256), 0, stream, ws.orig, ws.invp, np);
    hipLaunchKernelGGL(b64_unpermute_kernel, dim3((p.n + 255) / 256, (p.n + 15) / 16, batch), dim3(256), 0, stream, cur,
                       p.ld, np, ws.wstride, ws.invp, p.n, d_inv);
    return hipGetLastError();
}

}  // namespace mi32
