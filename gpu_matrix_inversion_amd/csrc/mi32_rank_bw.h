// mi32_rank_bw.h -- the rank-bw update of the blocked path (gfx950 only).
//
//   dst[i][j] = C[i][j] - sum_m f_m[i] * u_m[j],   m = 0 .. kdim-1 ascending, one fmaf chain per element
//
// i.e. fixColumnKernel (/root/reference/Matlab/mat_inv_32/mat_inv_32/mat_inv_32.cpp:13-57) for all columns outside
// the current block of kdim pivots, with the block's kdim eliminations applied at once on the fp32 matrix cores, in
// the reference's own order: f_m[i] = the entry row i had in the pivot column when step m ran (the block's
// multipliers), u_m[j] = the pivot row of step m as fixColumn saw it (the block's strip, mi32_blocked.hip), C = the
// element's old value -- a v_mfma_f32_32x32x2_f32 chain with C as its accumulator is exactly that fmaf chain.
//
//  * both operand tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no
//    ds_write instructions, and the loads of k-tile t+1 are in flight during the MFMAs of k-tile t;
//  * for that the A operand (the negated multipliers, row-major by block-start row in mf) is first gathered through
//    the block's row map and transposed into a compact k-major array Gk[k][row] by gj_mult_transpose_kernel, so that
//    an A tile is [BK][128] contiguous rows exactly like a B tile -- the LDS image of an LDS-DMA is lane-linear, it
//    cannot transpose; the B operand ub[k][col] is compact and k-major as the strip leaves it.
#pragma once
#include "mi32_internal.h"
#include <type_traits>
#include <utility>

namespace mi32 {

// Where a wide kernel exports freshly computed columns for the panel workgroup: columns
// [col, col + w * count) go to `count` consecutive compact panels (w columns each) starting at `base`.
struct PanelExport {
    float *base;     // first compact panel, matrix 0
    size_t bstride;  // floats between consecutive compact panels
    int col, w, count;
};
__device__ __forceinline__ void panel_export_store(const PanelExport &e, size_t tstride, int b, int np, int col, int grow,
                                                   float v)
{
    const int idx = col - e.col;
    if ((unsigned)idx < (unsigned)(e.w * e.count))
        e.base[(size_t)(idx / e.w) * e.bstride + (size_t)b * tstride + (size_t)(idx % e.w) * np + grow] = v;
}

// four consecutive rows (grow4 a multiple of 4) of one exported column: one 16-byte store
__device__ __forceinline__ void panel_export_store4(const PanelExport &e, size_t tstride, int b, int np, int col,
                                                    int grow4, float v0, float v1, float v2, float v3)
{
    const int idx = col - e.col;
    if ((unsigned)idx < (unsigned)(e.w * e.count)) {
        typedef float pe_f4v __attribute__((ext_vector_type(4)));
        pe_f4v v;
        v[0] = v0; v[1] = v1; v[2] = v2; v[3] = v3;
        *reinterpret_cast<pe_f4v *>(e.base + (size_t)(idx / e.w) * e.bstride + (size_t)b * tstride +
                                    (size_t)(idx % e.w) * np + grow4) = v;
    }
}

typedef float rb_float16v __attribute__((ext_vector_type(16)));
typedef float rb_f4v __attribute__((ext_vector_type(4)));

// One LDS-DMA: 64 lanes x 16 B from per-lane global addresses to 1 KiB of LDS at a wave-uniform base.
__device__ __forceinline__ void rb_glds16(const float *gsrc, float *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// LDS footprint of one workgroup of gj_rank_bw2_kernel<BK>: two stages of (A tile + B tile), the C-row
// map of the tile and the pivot-row map of the block.
template <int BK, int BN = 128>
constexpr size_t rank_bw2_lds_bytes(int kdim)
{
    return (size_t)(2 * BK * (128 + BN)) * sizeof(float) + (size_t)(128 + kdim) * sizeof(int);
}

// XCD-aware tile order (see mi32_blocked.hip): workgroups that share an XCD (= an L2) cover a compact sub-grid of
// TR/2 x TC/4 tiles, and walk it in column strips of `sw` tiles, row by row inside a strip: the strip's B operand
// (sw x 128 columns x kdim steps) stays in the XCD's 4 MB L2 while the rows go by, and every A row tile is fetched once
// per strip.  Row-major over the whole sub-grid (rounds 1-2) re-fetched the sub-grid's B panel for every few tile rows
// once it outgrew the L2: 16384^2, kdim 256, 1.97x the algorithmic bytes from HBM.
__device__ __forceinline__ int rb_strip_width(int kdim, int bn)
{
    const int sw = (2 * 1024 * 1024) / (bn * kdim * (int)sizeof(float));  // half the L2 for the B strip
    return sw < 1 ? 1 : sw;
}
__device__ __forceinline__ void rb_tile_of(int id, int TR, int TC, int sw, int &rt, int &ct)
{
    if ((TR & 1) == 0 && (TC & 3) == 0) {
        const int xcd = id & 7, idx = id >> 3;
        const int tr = TR / 2, tc = TC / 4;
        int r, c;
        const int nfull = tc / sw;  // full strips
        if (idx < nfull * tr * sw) {
            const int strip = idx / (tr * sw), rem = idx - strip * (tr * sw);
            r = rem / sw;
            c = strip * sw + rem % sw;
        } else {  // the last, narrower strip
            const int w = tc - nfull * sw, rem = idx - nfull * tr * sw;
            r = rem / w;
            c = nfull * sw + rem % w;
        }
        rt = (xcd >> 2) * tr + r;
        ct = (xcd & 3) * tc + c;
    } else {
        rt = id / TC;
        ct = id % TC;
    }
}

// Diagnostic builds (-DMI32_RB_STAMPS) record s_memtime at the phase boundaries of
// every workgroup; in the product build the macro expands to nothing.
#ifdef MI32_RB_STAMPS
__device__ unsigned long long *g_rb_stamps;  // [workgroup][8]
#define MI32_RB_STAMP(slot_)                                                                          \
    do {                                                                                              \
        if (g_rb_stamps && threadIdx.x == 0)                                                          \
            g_rb_stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot_)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define MI32_RB_STAMP(slot_) do { } while (0)
#endif

// One 128 x BN output tile (rt, ct) of matrix b (BN = 128 or 64: 4 waves as 2 x 2, 64 x BN/2 each).
// rb_smem: rank_bw2_lds_bytes<BK, BN>(kdim) bytes of LDS.
// The accumulators START from the old values C (row-mapped), so that every element goes through one fmaf chain from
// its old value, k ascending -- the reference's own operation order (mat_inv_32.cpp:28-38).  The old values are
// requested before the first operand stage, so both travel together.  (PF is kept as a template parameter for the
// call sites; round 2 fetched the old values under the last k-tiles, which the order from the old value rules out.)
template <int BK, int BN = 128, bool PF = false>
__device__ __forceinline__ void rank_bw2_tile(
    const float *__restrict__ src_all, float *__restrict__ dst_all, const float *__restrict__ g_all, size_t gstride,
    const float *__restrict__ gk_all, size_t gkstride, const float *__restrict__ ub_all, const float *__restrict__ xs_all,
    int np, int ld, size_t mstride, int c0, int kdim, const int *__restrict__ map_all, int copy_panel,
    const PanelExport &ex, size_t tstride, int skip_lo, int skip_hi, int b, int rt, int ct, float *rb_smem)
{
    constexpr int BM = 128;
    constexpr int TN = BN / 64;          // 32-column MFMA tiles per wave
    constexpr int ND = BK / 8;           // LDS-DMA instructions per wave per stage for A (each moves 2 k-rows)
    constexpr int NDB = BK * BN / 1024;  // ... for B (each moves 256 / BN k-rows)
    constexpr int KPB = 256 / BN;        // k-rows of B per LDS-DMA instruction
    static_assert(NDB >= 1, "B tile smaller than one LDS-DMA per wave");
    float *s_a = rb_smem;                    // [2][BK][128]
    float *s_b = rb_smem + 2 * BK * 128;     // [2][BK][BN]
    int *s_map = reinterpret_cast<int *>(rb_smem + 2 * BK * (128 + BN));  // [128]  C rows of this tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int row0 = rt * BM, col0 = ct * BN;
    const float *src = src_all + (size_t)b * mstride;
    float *dst = dst_all + (size_t)b * mstride;
    const float *gk = gk_all + (size_t)b * gkstride;
    const float *ub = ub_all + (size_t)b * gkstride;
    const float *xs = xs_all + (size_t)b * gkstride;
    const int *map = map_all + (size_t)b * np;

    if (col0 >= skip_lo && col0 < skip_hi) return;
    if (col0 >= c0 && col0 + BN <= c0 + kdim) {  // tile inside the panel: those columns are up to date already
        if (copy_panel) {
            const float *g = g_all + (size_t)b * gstride;
            for (int idx = tid; idx < BM * (BN / 4); idx += 256) {
                const int rr = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
                *reinterpret_cast<rb_f4v *>(dst + (size_t)(row0 + rr) * ld + col0 + c4) =
                    *reinterpret_cast<const rb_f4v *>(g + (size_t)(row0 + rr) * ld + col0 + c4);
            }
        }
        return;
    }
    // Block bounds are multiples of 128, so a whole tile is either inside the block or outside it.  The block's own
    // pivot rows start from what the strip of their sub-panel left (xs[k][col], k-major like ub) and only take the
    // later sub-panels' steps: gj_mult_transpose_kernel has zeroed their other multipliers.
    const bool tile_in_block = (row0 >= c0 && row0 < c0 + kdim);

    MI32_RB_STAMP(0);
    if (tid < BM) s_map[tid] = map[row0 + tid];
    __syncthreads();

    // per-lane source addresses of this wave's DMA slices: k-rows 2*(wave*ND + i) + (lane >> 5) of a stage
    const int lk = lane >> 5, lc4 = (lane & 31) * 4;
    const int lkb = lane / (BN / 4), lcb4 = (lane % (BN / 4)) * 4;  // B: k-row and column inside one LDS-DMA
    const float *a_src = gk + (size_t)lk * np + row0 + lc4;   // + (kt + 2 * (wave * ND + i)) * np
    const float *b_src = ub + (size_t)lkb * np + col0 + lcb4; // + (kt + KPB * (wave * NDB + i)) * np

#define MI32_RB_ISSUE(STAGE, KT)                                                                       \
    _Pragma("unroll") for (int i = 0; i < ND; ++i) {                                                   \
        const int kr = 2 * (wave * ND + i);                                                            \
        rb_glds16(a_src + (size_t)((KT) + kr) * np, s_a + ((STAGE) * BK + kr) * 128);                  \
    }                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NDB; ++i) {                                                  \
        const int kr = KPB * (wave * NDB + i);                                                         \
        rb_glds16(b_src + (size_t)((KT) + kr) * np, s_b + ((STAGE) * BK + kr) * BN);                   \
    }

    const int lcol = lane & 31;
    const int lhalf = lane >> 5;
    const int nk = kdim / BK;
    rb_float16v acc[2][TN];
    if (tile_in_block) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int col = col0 + wc * (BN / 2) + tn * 32 + lcol;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int lr = wr * 64 + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhalf;
                    acc[tm][tn][reg] = xs[(size_t)(row0 + lr - c0) * np + col];
                }
            }
    } else {
        // Old values through the row map: read once -> streaming loads (they must not push the operand strips out of
        // the L2).  Straight-line code: the 16 map entries of a 32-row group in four 16-byte LDS reads, then per value
        // ONE 32-bit multiply-add for the byte offset from the matrix's scalar base (np <= 16384: the last byte of a
        // matrix lies below 2^31) -- a per-value choice between the two sources costs an LDS round trip and three
        // branches per value (measured: it did, for a round).
        const unsigned ld4 = (unsigned)ld * 4u;
        const char *srcb = reinterpret_cast<const char *>(src);
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            int4 m4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) m4[q] = *reinterpret_cast<const int4 *>(&s_map[wr * 64 + tm * 32 + 8 * q + 4 * lhalf]);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const unsigned col4 = (unsigned)(col0 + wc * (BN / 2) + tn * 32 + lcol) * 4u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int mm[4] = {m4[q].x, m4[q].y, m4[q].z, m4[q].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[tm][tn][4 * q + j] = __builtin_nontemporal_load(
                            reinterpret_cast<const float *>(srcb + ((unsigned)mm[j] * ld4 + col4)));
                }
            }
        }
    }
    MI32_RB_STAMP(1);
    MI32_RB_ISSUE(0, 0)
    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        // stage t has landed for this wave's DMAs (and, the first time, the old values); after the barrier for
        // everyone's, and every wave is done reading the other buffer (k-tile t-1), which the next DMA overwrites
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < nk) { MI32_RB_ISSUE(buf ^ 1, (t + 1) * BK) }
        const float *pa = s_a + buf * BK * 128 + lhalf * 128 + wr * 64 + lcol;
        const float *pb = s_b + buf * BK * BN + lhalf * BN + wc * (BN / 2) + lcol;
        float af[2], bf[TN];
        af[0] = pa[0]; af[1] = pa[32];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bf[tn] = pb[tn * 32];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float afn[2] = {0.f, 0.f}, bfn[TN] = {};
            if (kk + 2 < BK) {
                afn[0] = pa[(kk + 2) * 128]; afn[1] = pa[(kk + 2) * 128 + 32];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bfn[tn] = pb[(kk + 2) * BN + tn * 32];
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ABOVE this pair's MFMAs
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tm], bf[tn], acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 2; ++q) af[q] = afn[q];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bf[tn] = bfn[tn];
        }
    }
#undef MI32_RB_ISSUE
    MI32_RB_STAMP(2);

#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int col = col0 + wc * (BN / 2) + tn * 32 + lcol;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int grow = row0 + wr * 64 + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhalf;
                const unsigned boff = (unsigned)grow * ((unsigned)ld * 4u) + (unsigned)col * 4u;
                __builtin_nontemporal_store(acc[tm][tn][reg],
                                            reinterpret_cast<float *>(reinterpret_cast<char *>(dst) + boff));
            }
            // the next block's first sub-panels, compact and transposed: registers 4q .. 4q+3 are 4 consecutive rows
#pragma unroll
            for (int q = 0; q < 4; ++q)
                panel_export_store4(ex, tstride, b, np, col, row0 + wr * 64 + tm * 32 + 8 * q + 4 * lhalf,
                                    acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2],
                                    acc[tm][tn][4 * q + 3]);
        }
    MI32_RB_STAMP(3);
#ifdef MI32_RB_STAMPS
    if (g_rb_stamps && threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_rb_stamps[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + 4] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
}

template <int BK, int WPS, int BN = 128, bool PF = false>
__global__ __launch_bounds__(256, WPS) void gj_rank_bw2_kernel(
    const float *__restrict__ src_all, float *__restrict__ dst_all, const float *__restrict__ g_all, size_t gstride,
    const float *__restrict__ gk_all, size_t gkstride, const float *__restrict__ ub_all, const float *__restrict__ xs_all,
    int np, int ld, size_t mstride, int c0, int kdim, const int *__restrict__ map_all, int copy_panel, PanelExport ex,
    size_t tstride, int skip_lo, int skip_hi, const int *__restrict__ guard)
{
    extern __shared__ __attribute__((aligned(16))) float rb_smem[];
    if (guard != nullptr && __builtin_amdgcn_readfirstlane(guard[blockIdx.y]) == MI32_RUNTIME_ERROR) return;  // given up
    int rt, ct;
    rb_tile_of(blockIdx.x, np / 128, np / BN, rb_strip_width(kdim, BN), rt, ct);
    rank_bw2_tile<BK, BN, PF>(src_all, dst_all, g_all, gstride, gk_all, gkstride, ub_all, xs_all, np, ld, mstride, c0, kdim,
                              map_all, copy_panel, ex, tstride, skip_lo, skip_hi, blockIdx.y, rt, ct, rb_smem);
}

// Persistent, residency-limited flavour for the look-ahead half (see blocked_invert): gridDim.x workgroups
// walk all the tiles.  It is launched with enough dynamic LDS that only ONE workgroup fits on a CU and with
// fewer workgroups than CUs, so a known number of CUs stays entirely free for the critical-path kernels
// of the main stream (the panel kernel needs a whole CU); stream priorities cannot give that guarantee
// and a CU mask serialises the queues.
template <int BK, bool PF = false>
__global__ __launch_bounds__(256, 1) void gj_rank_bw2_persistent_kernel(
    const float *__restrict__ src_all, float *__restrict__ dst_all, const float *__restrict__ g_all, size_t gstride,
    const float *__restrict__ gk_all, size_t gkstride, const float *__restrict__ ub_all, const float *__restrict__ xs_all,
    int np, int ld, size_t mstride, int c0, int kdim, const int *__restrict__ map_all, int copy_panel, PanelExport ex,
    size_t tstride, int skip_lo, int skip_hi, const int *__restrict__ guard)
{
    extern __shared__ __attribute__((aligned(16))) float rb_smem[];
    if (guard != nullptr && __builtin_amdgcn_readfirstlane(guard[blockIdx.y]) == MI32_RUNTIME_ERROR) return;  // given up
    const int T = np / 128;
    for (int id = blockIdx.x; id < T * T; id += gridDim.x) {
        int rt, ct;
        rb_tile_of(id, T, T, rb_strip_width(kdim, 128), rt, ct);
        rank_bw2_tile<BK, 128, PF>(src_all, dst_all, g_all, gstride, gk_all, gkstride, ub_all, xs_all, np, ld, mstride, c0,
                                   kdim, map_all, copy_panel, ex, tstride, skip_lo, skip_hi, blockIdx.y, rt, ct, rb_smem);
        __syncthreads();  // the next tile re-uses the LDS buffers and maps
    }
}

}  // namespace mi32
