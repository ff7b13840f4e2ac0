// mi32_blocked.hip -- blocked Gauss-Jordan with delayed rank-k updates on the
// fp32 matrix cores (gfx950: v_mfma_f32_32x32x2_f32).
//
// Same elimination as mi32_sweep.hip (and as the reference's step loop,
// /root/reference/Matlab/mat_inv_32/mat_inv_32/mat_inv_32.cpp:317-362), with the
// column updates of a block of pivots delayed:
//
//   for each outer block K = [C0, C0+kb) of pivot columns:
//     for each sub-panel Ks = [c0, c0+W) of K:
//       gj_panel_kernel     -- ONE workgroup per matrix holds all rows of the W
//                              sub-panel columns in registers and runs the W
//                              pivot steps on them: column arg-max (wave64 shuffle
//                              + LDS), row swap, IEEE-division normalise,
//                              eliminate.  Result: G_s = the W transformed
//                              columns (the inverse columns of these pivots).
//       gj_rank_update_kernel (K = W)  -- every other column j of the block:
//                              M[i][j] = (i in Ks ? 0 : M[src(i)][j])
//                                        + sum_k G_s[i][k] * M[src(c0+k)][j]
//     gj_rank_update_kernel (K = kb)   -- every column outside the block, same
//                              formula with the block's composite G and row map.
//
// Row swaps are never applied as data movement of their own: the updates read
// their C rows and their B (pivot-row) operand THROUGH a row map and write
// out-of-place into the second working copy, so swap + snapshot + update are
// one launch and no launch has a read-after-write hazard between workgroups.
// The two working copies alternate roles exactly like the reference's
// ping-pong buffers (mat_inv_32.cpp:318,353-360).
//
// The working matrix is the N x N in-place form (see mi32_sweep.hip), padded
// with an identity block to a multiple of 128 so that no tile needs bounds
// checks: inv(diag(A, I)) = diag(inv(A), I), and the padding rows are exact
// zeros in every real column, so they can never win a pivot search.
#include "mi32_internal.h"

namespace mi32 {

typedef float float16v __attribute__((ext_vector_type(16)));

static constexpr int kPanelThreads = 512;

BlockedPlan make_blocked_plan(int n, int w, int bw)
{
    BlockedPlan p;
    p.n = n;
    p.np = (n + 127) & ~127;
    // Row stride: np + 64 floats.  A power-of-two stride would put the same column chunk of every
    // row on one L2 channel (the panel kernel reads 64 B of each of np rows); 256 B of padding
    // rotates consecutive rows over the channels and keeps rows 256-B aligned.
    p.ld = p.np + 64;
    p.nthreads_panel = kPanelThreads;
    int rpt = (p.np + kPanelThreads - 1) / kPanelThreads;
    int r2 = 1;
    while (r2 < rpt) r2 *= 2;
    p.rpt = r2;
    // registers: rpt * w floats per thread must stay <= 128
    int wmax = 128 / r2;
    if (wmax > 16) wmax = 16;
    if (wmax < 4) wmax = 4;
    if (w <= 0) w = 16;
    if (w > wmax) w = wmax;
    if (w != 4 && w != 8 && w != 16) w = (w > 8) ? 16 : (w > 4 ? 8 : 4);
    p.w = w;
    if (bw <= 0) bw = 256;
    bw = (bw + 127) & ~127;
    if (bw > 512) bw = 512;
    if (bw > p.np) bw = p.np;
    p.bw = bw;
    return p;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
struct BlockedWs {
    float *m0, *m1;
    int *submap, *rowsrc, *orig, *invp;
    size_t mstride;
};
static size_t blocked_carve(const BlockedPlan &p, int batch, void *base, BlockedWs *o)
{
    const size_t mbytes = align256((size_t)p.np * p.ld * sizeof(float));
    const size_t ibytes = align256((size_t)p.np * sizeof(int) * batch);
    char *c = (char *)base;
    size_t off = 0;
    if (o) { o->m0 = (float *)(c + off); o->mstride = mbytes / sizeof(float); }
    off += mbytes * batch;
    if (o) o->m1 = (float *)(c + off);
    off += mbytes * batch;
    if (o) o->submap = (int *)(c + off);
    off += ibytes;
    if (o) o->rowsrc = (int *)(c + off);
    off += ibytes;
    if (o) o->orig = (int *)(c + off);
    off += ibytes;
    if (o) o->invp = (int *)(c + off);
    off += ibytes;
    return off;
}
size_t blocked_workspace_bytes(const BlockedPlan &p, int batch) { return blocked_carve(p, batch, nullptr, nullptr); }

// ---- init: A -> diag(A, I) in the first working copy --------------------------
__global__ __launch_bounds__(256) void blocked_init_kernel(const float *__restrict__ in, int n, int np, int ld,
                                                            size_t mstride,
                                                            float *__restrict__ m0, int *__restrict__ orig,
                                                            int *__restrict__ status)
{
    const int b = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i0 = blockIdx.y * 16;
    const float *a = in + (size_t)b * n * n;
    float *m = m0 + (size_t)b * mstride;
    if (j < np) {
#pragma unroll 4
        for (int u = 0; u < 16; ++u) {
            const int i = i0 + u;
            if (i >= np) break;
            float v;
            if (i < n && j < n) v = a[(size_t)i * n + j];
            else v = (i == j) ? 1.0f : 0.0f;
            m[(size_t)i * ld + j] = v;
        }
    }
    if (blockIdx.y == 0 && j < np) orig[(size_t)b * np + j] = j;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && status) status[b] = MI32_OK;
}

// ---- wave-level arg-max helpers (DPP, no LDS traffic) ----------------------------
// Canonical gfx9 wave64 reduction: quad_perm x2, row_half_mirror, row_mirror, then
// row_bcast15 / row_bcast31 fold the four rows; lane 63 ends up with the total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    unsigned t;
    t = dpp_u32<0xB1, 0xF>(v); v = t > v ? t : v;   // quad_perm [1,0,3,2]
    t = dpp_u32<0x4E, 0xF>(v); v = t > v ? t : v;   // quad_perm [2,3,0,1]
    t = dpp_u32<0x141, 0xF>(v); v = t > v ? t : v;  // row_half_mirror
    t = dpp_u32<0x140, 0xF>(v); v = t > v ? t : v;  // row_mirror
    t = dpp_u32<0x142, 0xA>(v); v = t > v ? t : v;  // row_bcast15 -> rows 1,3
    t = dpp_u32<0x143, 0xC>(v); v = t > v ? t : v;  // row_bcast31 -> rows 2,3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    unsigned t;
    t = dpp_u32<0xB1, 0xF>(v); v = t < v ? t : v;
    t = dpp_u32<0x4E, 0xF>(v); v = t < v ? t : v;
    t = dpp_u32<0x141, 0xF>(v); v = t < v ? t : v;
    t = dpp_u32<0x140, 0xF>(v); v = t < v ? t : v;
    t = dpp_u32<0x142, 0xA>(v); v = t < v ? t : v;
    t = dpp_u32<0x143, 0xC>(v); v = t < v ? t : v;
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float lane_bcast(float v, int srclane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srclane));
}

// ---- the panel: W pivot steps on an (np x W) register-resident slab -----------
// Thread t keeps rows t, t+512, ... (RPT of them) of the source copy X in registers
// for the whole kernel: row CONTENTS never move between threads.  What a row swap
// changes is only an integer label pos[k] = the position (row index of the working
// matrix) that the content of register row k currently occupies:
//   pivotElements (mat_inv_32.cpp:154-173)  ==  exchange of two labels.
// The permutation becomes real when the slab is stored: register row k goes to row
// pos[k] of the destination copy Y, and submap[pos[k]] = its row in X tells the
// rank-k updates where every other column's data for that position still lives.
template <int W>
struct __attribute__((aligned(16))) PanelShared {
    float prow[W];              // the pivot row as found (un-normalised)
    unsigned long long key[W];  // one cross-wave arg-max word per step, zeroed at kernel start
};

// One pivot step; R is a template parameter so that every index into the register
// slab is a compile-time constant (a runtime index would send the slab to scratch).
template <int RPT, int W, int R>
__device__ __forceinline__ void panel_step(float (&a)[RPT][W], int (&pos)[RPT], PanelShared<W> &sh, int tid, int nrows,
                                           int n, int c0, bool &singular)
{
    const int lane = tid & 63;
    const int slot = c0 + R;
    // a real column may only take its pivot from the real rows: the identity padding must never be
    // swapped into the matrix (it would be, on an all-zero/NaN column, where every candidate ties at 0)
    const unsigned span = (unsigned)((slot < n ? n : nrows) - slot);

    // -- maxPivot: positions >= slot, largest |a|, lowest position among equals, NaN never wins.
    //    |a| >= 0, so its bit pattern orders like the value: integer max/min on the bits.
    unsigned bm = 0u, bi = 0x7fffffffu;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const float v = __builtin_fabsf(a[k][R]);
        const unsigned m = __float_as_uint(v);
        const bool ok = ((unsigned)(pos[k] - slot) < span) && (v == v);
        const bool take = ok && ((bi == 0x7fffffffu) || (m > bm) || (m == bm && (unsigned)pos[k] < bi));
        bm = take ? m : bm;
        bi = take ? (unsigned)pos[k] : bi;
    }
    const unsigned wm = wave_max_u32(bm);
    const unsigned wi = wave_min_u32((bm == wm) ? bi : 0x7fffffffu);
    if (lane == 0) atomicMax(&sh.key[R], ((unsigned long long)wm << 32) | (unsigned long long)(0xFFFFFFFFu - wi));
    __syncthreads();
    const unsigned long long key = sh.key[R];
    const unsigned pidx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
    const int p = (pidx == 0x7fffffffu) ? slot : (int)pidx;  // no candidate at all: keep the slot's own row

    // -- the holder of position p publishes its row (only that wave enters the block)
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const bool mine = (pos[k] == p);
        if (__any(mine)) {
            if (mine) {
#pragma unroll
                for (int c = 0; c < W; c += 4)
                    *reinterpret_cast<float4 *>(&sh.prow[c]) = make_float4(a[k][c], a[k][c + 1], a[k][c + 2], a[k][c + 3]);
            }
        }
    }
    __syncthreads();

    // -- fixRow: lanes 0..W-1 of every wave divide one element each (IEEE), the
    //    identity entry becomes 1/piv; broadcast through SGPRs
    const float piv = sh.prow[R];
    const float num = (lane < W) ? ((lane == R) ? 1.0f : sh.prow[lane]) : 0.0f;
    const float qv = num / piv;
    float prn[W];
#pragma unroll
    for (int c = 0; c < W; ++c) prn[c] = lane_bcast(qv, c);
    if (piv == 0.0f || piv != piv) singular = true;

    // -- fixColumn on the slab, branch-free (the pivot row itself is overwritten right after)
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const float f = a[k][R];
        a[k][R] = 0.0f;
#pragma unroll
        for (int c = 0; c < W; ++c) a[k][c] = __builtin_fmaf(-f, prn[c], a[k][c]);
    }
    // -- pivot row := normalised pivot row; pivotElements == exchange of the two position labels
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const bool mine = (pos[k] == p);
        if (__any(mine)) {
            if (mine) {
#pragma unroll
                for (int c = 0; c < W; ++c) a[k][c] = prn[c];
            }
        }
        pos[k] = mine ? slot : ((pos[k] == slot) ? p : pos[k]);
    }
}

template <int RPT, int W, int R>
struct PanelSteps {
    static __device__ __forceinline__ void run(float (&a)[RPT][W], int (&pos)[RPT], PanelShared<W> &sh, int tid,
                                               int nrows, int n, int c0, bool &singular)
    {
        panel_step<RPT, W, R>(a, pos, sh, tid, nrows, n, c0, singular);
        PanelSteps<RPT, W, R + 1>::run(a, pos, sh, tid, nrows, n, c0, singular);
    }
};
template <int RPT, int W>
struct PanelSteps<RPT, W, W> {
    static __device__ __forceinline__ void run(float (&)[RPT][W], int (&)[RPT], PanelShared<W> &, int, int, int, int,
                                               bool &)
    {
    }
};

template <int RPT, int W>
__global__ __launch_bounds__(kPanelThreads) void gj_panel_kernel(const float *__restrict__ x_all,
                                                                  float *__restrict__ y_all, int np, int ld, int n,
                                                                  size_t mstride, int c0,
                                                                  int *__restrict__ submap_all,
                                                                  int *__restrict__ rowsrc_all,
                                                                  int *__restrict__ orig_all, int first_in_block,
                                                                  int *__restrict__ status)
{
    __shared__ PanelShared<W> sh;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const float *x = x_all + (size_t)b * mstride;
    float *y = y_all + (size_t)b * mstride;
    if (tid < W) sh.key[tid] = 0ull;

    float a[RPT][W];
    int pos[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int row = tid + k * kPanelThreads;
        // rows beyond the matrix (only when np is not a multiple of 512) get a label no step can match
        pos[k] = row < np ? row : 0x40000000 + row;
        if (row < np) {
#pragma unroll
            for (int c = 0; c < W; c += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(x + (size_t)row * ld + c0 + c);
                a[k][c] = v.x; a[k][c + 1] = v.y; a[k][c + 2] = v.z; a[k][c + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int c = 0; c < W; ++c) a[k][c] = 0.0f;
        }
    }
    bool singular = false;
    __syncthreads();  // sh.key[] zeroed before any wave's first atomicMax
    PanelSteps<RPT, W, 0>::run(a, pos, sh, tid, np, n, c0, singular);

    // -- store G_s through the permutation and publish the row maps
    int *submap = submap_all + (size_t)b * np;
    int *rowsrc = rowsrc_all + (size_t)b * np;
    int *orig = orig_all + (size_t)b * np;
    int nrs[RPT], nor[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int row = tid + k * kPanelThreads;
        if (row < np) {
#pragma unroll
            for (int c = 0; c < W; c += 4)
                *reinterpret_cast<float4 *>(y + (size_t)pos[k] * ld + c0 + c) =
                    make_float4(a[k][c], a[k][c + 1], a[k][c + 2], a[k][c + 3]);
            submap[pos[k]] = row;                            // position pos[k] now holds X's row `row`
            nrs[k] = first_in_block ? row : rowsrc[row];     // composite map of the block so far
            nor[k] = orig[row];
        }
    }
    __syncthreads();  // every read of rowsrc/orig above precedes every write below
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int row = tid + k * kPanelThreads;
        if (row < np) {
            rowsrc[pos[k]] = nrs[k];
            orig[pos[k]] = nor[k];
        }
    }
    if (singular && tid == 0 && status) status[b] = MI32_SINGULAR;
}

// ---- rank-k update on the fp32 matrix cores ------------------------------------
//   dst[i][j] = (i in [c0,c0+kdim) ? 0 : src[map[i]][j]) + sum_k G[i][c0+k] * src[map[c0+k]][j]
// for the columns j of this tile that are not panel columns.  256 threads = 4
// waves in a 2x2 arrangement; each wave owns (BM/2)x(BN/2) as 32x32 MFMA tiles.
// A (= G, row-major, k contiguous) is transposed into LDS as [k][row] so that the
// 32 lanes of a half-wave read 32 consecutive floats; B (= pivot rows, row-major)
// is staged as it lies.  One accumulation chain per output element, k ascending:
// bit-for-bit the fmaf chain of oracle/gj_oracle.c's blocked restatement.
template <int BM, int BN, int BK>
__global__ __launch_bounds__(256) void gj_rank_update_kernel(const float *__restrict__ src_all,
                                                              float *__restrict__ dst_all,
                                                              const float *__restrict__ g_all, int np, int ld,
                                                              size_t mstride, int c0, int kdim, int col_lo,
                                                              const int *__restrict__ map_all, int copy_panel)
{
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int PADA = (32 / BK) > 0 ? (32 / BK) : 1;
    constexpr int LDA = BM + PADA;
    constexpr int LDB = BN + 4;
    __shared__ float s_a[BK * LDA];
    __shared__ __attribute__((aligned(16))) float s_b[BK * LDB];
    __shared__ int s_map[BM];

    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int row0 = blockIdx.y * BM;
    const int col0 = col_lo + blockIdx.x * BN;
    const float *src = src_all + (size_t)b * mstride;
    float *dst = dst_all + (size_t)b * mstride;
    const float *g = g_all + (size_t)b * mstride;
    const int *map = map_all + (size_t)b * np;

    if (col0 >= c0 && col0 + BN <= c0 + kdim) {
        // tile lies inside the panel: those columns are G itself
        if (copy_panel) {
            for (int idx = tid; idx < BM * (BN / 4); idx += 256) {
                const int rr = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
                *reinterpret_cast<float4 *>(dst + (size_t)(row0 + rr) * ld + col0 + c4) =
                    *reinterpret_cast<const float4 *>(g + (size_t)(row0 + rr) * ld + col0 + c4);
            }
        }
        return;
    }

    for (int i = tid; i < BM; i += 256) s_map[i] = map[row0 + i];
    __syncthreads();

    // accumulators start from the (row-mapped) old values; rows of the block start from 0
    float16v acc[TM][TN];
    const int lcol = lane & 31;
    const int lhalf = lane >> 5;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = col0 + wc * WN + tn * 32 + lcol;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int lr = wr * WM + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhalf;
                const int grow = row0 + lr;
                const bool in_block = (grow >= c0 && grow < c0 + kdim);
                acc[tm][tn][reg] = in_block ? 0.0f : src[(size_t)s_map[lr] * ld + col];
            }
        }

    for (int kt = 0; kt < kdim; kt += BK) {
        // stage A: BM x BK of G, transposed
#pragma unroll
        for (int q = 0; q < (BM * BK / 4 + 255) / 256; ++q) {
            const int idx = tid + q * 256;
            if (idx < BM * BK / 4) {
                const int rr = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
                const float4 v = *reinterpret_cast<const float4 *>(g + (size_t)(row0 + rr) * ld + c0 + kt + k4);
                s_a[(k4 + 0) * LDA + rr] = v.x;
                s_a[(k4 + 1) * LDA + rr] = v.y;
                s_a[(k4 + 2) * LDA + rr] = v.z;
                s_a[(k4 + 3) * LDA + rr] = v.w;
            }
        }
        // stage B: BK pivot rows (through the row map) x BN columns
#pragma unroll
        for (int q = 0; q < (BK * BN / 4 + 255) / 256; ++q) {
            const int idx = tid + q * 256;
            if (idx < BK * BN / 4) {
                const int kk = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
                const int brow = map[c0 + kt + kk];
                const float4 v = *reinterpret_cast<const float4 *>(src + (size_t)brow * ld + col0 + c4);
                *reinterpret_cast<float4 *>(&s_b[kk * LDB + c4]) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float af[TM], bf[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) af[tm] = s_a[(kk + lhalf) * LDA + wr * WM + tm * 32 + lcol];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bf[tn] = s_b[(kk + lhalf) * LDB + wc * WN + tn * 32 + lcol];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm], bf[tn], acc[tm][tn], 0, 0, 0);
        }
        __syncthreads();
    }

#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = col0 + wc * WN + tn * 32 + lcol;
            if (col >= c0 && col < c0 + kdim) continue;  // panel column: already holds G
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int grow = row0 + wr * WM + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhalf;
                dst[(size_t)grow * ld + col] = acc[tm][tn][reg];
            }
        }
}

// ---- getInvertedMatrix counterpart: undo the column permutation ----------------
__global__ void invert_perm_ld_kernel(const int *__restrict__ orig, int *__restrict__ invp, int n, int istride)
{
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) invp[(size_t)b * istride + orig[(size_t)b * istride + c]] = c;
}
__global__ __launch_bounds__(256) void unpermute_columns_ld_kernel(const float *__restrict__ w_all, int ld,
                                                                    size_t wstride, const int *__restrict__ invp,
                                                                    int istride, int n, float *__restrict__ out)
{
    const int b = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float *w = w_all + (size_t)b * wstride;
    float *o = out + (size_t)b * n * n;
    const int c = invp[(size_t)b * istride + j];
    const int i0 = blockIdx.y * 16;
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
        const int i = i0 + u;
        if (i < n) o[(size_t)i * n + j] = w[(size_t)i * ld + c];
    }
}

template <int RPT, int W>
static void launch_panel(const BlockedWs &ws, const float *x, float *y, int np, int ld, int n, int c0, int first,
                         int batch, int *d_status, hipStream_t stream)
{
    hipLaunchKernelGGL((gj_panel_kernel<RPT, W>), dim3(batch), dim3(kPanelThreads), 0, stream, x, y, np, ld, n,
                       ws.mstride, c0, ws.submap, ws.rowsrc, ws.orig, first, d_status);
}

static void dispatch_panel(const BlockedPlan &p, const BlockedWs &ws, const float *x, float *y, int c0, int first,
                           int batch, int *d_status, hipStream_t stream)
{
#define MI32_PANEL_CASE(R, WW)                                                  \
    if (p.rpt == R && p.w == WW) {                                              \
        launch_panel<R, WW>(ws, x, y, p.np, p.ld, p.n, c0, first, batch, d_status, stream); \
        return;                                                                 \
    }
    MI32_PANEL_CASE(1, 16) MI32_PANEL_CASE(2, 16) MI32_PANEL_CASE(4, 16) MI32_PANEL_CASE(8, 16)
    MI32_PANEL_CASE(1, 8) MI32_PANEL_CASE(2, 8) MI32_PANEL_CASE(4, 8) MI32_PANEL_CASE(8, 8) MI32_PANEL_CASE(16, 8)
    MI32_PANEL_CASE(1, 4) MI32_PANEL_CASE(2, 4) MI32_PANEL_CASE(4, 4) MI32_PANEL_CASE(8, 4) MI32_PANEL_CASE(16, 4)
    MI32_PANEL_CASE(32, 4)
#undef MI32_PANEL_CASE
}

static void launch_inner_update(const BlockedPlan &p, const BlockedWs &ws, const float *x, float *y, int c0, int C0,
                                int kb, int batch, hipStream_t stream)
{
    // columns [C0, C0+kb) of the block, K = w
    const dim3 grid(kb / 64, p.np / 64, batch);
    if (p.w == 16)
        hipLaunchKernelGGL((gj_rank_update_kernel<64, 64, 16>), grid, dim3(256), 0, stream, x, y, y, p.np, p.ld,
                           ws.mstride,
                           c0, 16, C0, ws.submap, 0);
    else if (p.w == 8)
        hipLaunchKernelGGL((gj_rank_update_kernel<64, 64, 8>), grid, dim3(256), 0, stream, x, y, y, p.np, p.ld,
                           ws.mstride,
                           c0, 8, C0, ws.submap, 0);
    else
        hipLaunchKernelGGL((gj_rank_update_kernel<64, 64, 4>), grid, dim3(256), 0, stream, x, y, y, p.np, p.ld,
                           ws.mstride,
                           c0, 4, C0, ws.submap, 0);
}

hipError_t blocked_invert(const BlockedPlan &p, const float *d_a, float *d_inv, int batch, int *d_status, void *wsp,
                          hipStream_t stream, Profiler *prof)
{
    BlockedWs ws;
    blocked_carve(p, batch, wsp, &ws);
    const int np = p.np;
    {
        ProfScope ps(prof, KC_INIT, stream);
        hipLaunchKernelGGL(blocked_init_kernel, dim3((np + 255) / 256, (np + 15) / 16, batch), dim3(256), 0, stream,
                           d_a, p.n, np, p.ld, ws.mstride, ws.m0, ws.orig, d_status);
    }
    float *cur = ws.m0, *oth = ws.m1;
    for (int C0 = 0; C0 < np; C0 += p.bw) {
        const int kb = (C0 + p.bw <= np) ? p.bw : np - C0;
        float *x = cur, *y = oth;  // the block's panel columns alternate between the two copies
        for (int s = 0; s * p.w < kb; ++s) {
            const int c0 = C0 + s * p.w;
            {
                ProfScope ps(prof, KC_PANEL, stream);
                dispatch_panel(p, ws, x, y, c0, s == 0, batch, d_status, stream);
            }
            if (kb > p.w) {
                ProfScope ps(prof, KC_UPDATE_IN, stream);
                launch_inner_update(p, ws, x, y, c0, C0, kb, batch, stream);
            }
            float *t = x; x = y; y = t;
        }
        // x now holds the block's G; every other column is still valid in `cur` only
        if (kb < np) {
            const dim3 grid(np / 128, np / 128, batch);
            ProfScope ps(prof, KC_UPDATE_OUT, stream);
            hipLaunchKernelGGL((gj_rank_update_kernel<128, 128, 32>), grid, dim3(256), 0, stream, cur, oth, x, np,
                               p.ld, ws.mstride, C0, kb, 0, ws.rowsrc, (x != oth) ? 1 : 0);
            float *t = cur; cur = oth; oth = t;
        } else {
            cur = x;  // single block: the panel is the whole matrix
        }
    }
    ProfScope ps(prof, KC_FINISH, stream);
    // over ALL np entries: orig is a permutation of [0, np), so every invp[j] is defined and in range
    hipLaunchKernelGGL(invert_perm_ld_kernel, dim3((np + 255) / 256, batch), dim3(256), 0, stream, ws.orig, ws.invp,
                       np, np);
    hipLaunchKernelGGL(unpermute_columns_ld_kernel, dim3((p.n + 255) / 256, (p.n + 15) / 16, batch), dim3(256), 0,
                       stream, cur, p.ld, ws.mstride, ws.invp, np, p.n, d_inv);
    return hipGetLastError();
}

}  // namespace mi32
