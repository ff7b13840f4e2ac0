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
//                              pivot steps on them: column arg-max (DPP + one LDS
//                              atomic), row swap (= exchange of two position
//                              labels), IEEE-division normalise, eliminate.
//                              Result: G_s = the W transformed columns (the
//                              inverse columns of these pivots).
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
// The panel kernel is a single workgroup on the critical path of all N pivot
// steps, so it only ever touches COMPACT, TRANSPOSED panels: it reads
// Pt[c][row] (W x np, written by whichever wide kernel produced those columns)
// and writes Gt[c][row], both with fully coalesced 16-byte accesses.  The wide
// update kernels, which hold those values anyway, export the next sub-panel's
// columns into Pt and materialise G_s into the row-major matrix.  (Letting the
// one workgroup gather 64-byte chunks of np rows itself cost 18 us per launch,
// more than its 16 pivot steps.)
//
// The working matrix is the N x N in-place form (see mi32_sweep.hip), padded
// with an identity block to a multiple of 128 so that no tile needs bounds
// checks: inv(diag(A, I)) = diag(inv(A), I); a real column only ever takes its
// pivot from the real rows, so the padding is never swapped into the matrix.
#include <cstdlib>
#include <utility>

#include "mi32_internal.h"
#include "mi32_rank_bw.h"

namespace mi32 {

typedef float float16v __attribute__((ext_vector_type(16)));

// k-tile depth and waves/SIMD of the rank-bw update (mi32_rank_bw.h; tunable at build time)
#ifndef MI32_BW_BK
#define MI32_BW_BK 16
#endif
#ifndef MI32_BW_WPS
#define MI32_BW_WPS 3
#endif
static constexpr int kMaxBW = 512;  // widest outer block (rows of the transposed panel Gk)

static constexpr int kMaxW = 32;  // widest sub-panel (columns kept in registers)

// Panel-kernel geometry: NT threads hold the rows at or below the block x w columns in registers, rpt rows
// each (1024 threads leave <= 128 VGPRs per lane, i.e. rpt * w <= 64 floats of slab).
// Thread geometry of a panel launch that holds `nrows` rows: the smallest that fits (fewer waves and fewer
// rows per lane both shorten a pivot step).
static void panel_geometry(const BlockedPlan &p, int nrows, int &nt, int &rpt)
{
    rpt = 1;
    if (nrows <= 256) nt = 256;
    else if (nrows <= 512) nt = 512;
    else {
        nt = p.nthreads_panel;
        while (rpt * nt < nrows) rpt *= 2;
    }
}

BlockedPlan make_blocked_plan(int n, int w, int bw)
{
    BlockedPlan p;
    p.n = n;
    p.np = (n + 127) & ~127;
    // Row stride: np + 64 floats (256 B): keeps rows 256-B aligned and avoids a power-of-two stride.
    p.ld = p.np + 64;
    int nt = (p.np >= 2048) ? 1024 : 512;
    if (const char *e = std::getenv("MI32_PANEL_THREADS")) {
        const int v = std::atoi(e);
        if (v == 512 || v == 1024) nt = v;
    }
    int rpt = 1;
    while (rpt * nt < p.np) rpt *= 2;
    if (rpt * 4 > 128 && nt == 512) {  // does not fit with 512 threads: use 1024
        nt = 1024;
        rpt = 1;
        while (rpt * nt < p.np) rpt *= 2;
    }
    p.nthreads_panel = nt;
    p.rpt = rpt;
    if (w <= 0) w = 16;  // 32 is selectable where it fits, but measured slower (4.6 vs 4.2 ms at 2048^2)
    w = (w >= 32) ? 32 : (w >= 16) ? 16 : (w >= 8 ? 8 : 4);
    p.w = w;
    if (bw <= 0) bw = 256;
    bw = (bw + 127) & ~127;
    if (bw > kMaxBW) bw = kMaxBW;
    if (bw > p.np) bw = p.np;
    p.bw = bw;
    p.nblk = (p.np + bw - 1) / bw;
    for (int b = 0; b < p.nblk && b < 128; ++b) {
        int bnt, brpt;
        panel_geometry(p, p.np - b * bw, bnt, brpt);
        int wmax = ((bnt == 1024) ? 64 : 128) / brpt;  // floats of slab per thread
        if (wmax > kMaxW) wmax = kMaxW;
        int wb = w < wmax ? w : wmax;                  // wmax < 4 (np > 16384) is rejected by blocked_supported()
        wb = (wb >= 32) ? 32 : (wb >= 16) ? 16 : (wb >= 8 ? 8 : 4);
        p.wblk[b] = (unsigned char)wb;
    }
    return p;
}
bool blocked_supported(int n) { return n > 0 && ((n + 127) & ~127) <= 16384; }

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
struct BlockedWs {
    float *m0, *m1;   // the two working copies, np x ld each
    float *pt[2], *gt;  // compact transposed panels, kMaxW x np each: panel kernel input (two, alternating) / output
    float *prn;         // kMaxW x kMaxW per matrix: the normalised pivot rows of the current sub-panel
    float *gk;        // the block's panel G transposed, bw x np: A operand of the rank-bw update
    size_t gkstride;  // floats per matrix in gk
    int *submap, *rowsrc[2], *orig, *invp;  // rowsrc is double-buffered across blocks (look-ahead)
    size_t mstride;   // floats per matrix in m0/m1
    size_t tstride;   // floats per matrix in pt/gt
};
static size_t blocked_carve(const BlockedPlan &p, int batch, void *base, BlockedWs *o)
{
    const size_t mbytes = align256((size_t)p.np * p.ld * sizeof(float));
    const size_t tbytes = align256((size_t)kMaxW * p.np * sizeof(float));
    const size_t ibytes = align256((size_t)p.np * sizeof(int) * batch);
    char *c = (char *)base;
    size_t off = 0;
    if (o) { o->m0 = (float *)(c + off); o->mstride = mbytes / sizeof(float); o->tstride = tbytes / sizeof(float); }
    off += mbytes * batch;
    if (o) o->m1 = (float *)(c + off);
    off += mbytes * batch;
    if (o) o->pt[0] = (float *)(c + off);
    off += tbytes * batch;
    if (o) o->pt[1] = (float *)(c + off);
    off += tbytes * batch;
    if (o) o->gt = (float *)(c + off);
    off += tbytes * batch;
    if (o) o->prn = (float *)(c + off);
    off += align256((size_t)kMaxW * kMaxW * sizeof(float) * batch);
    const size_t gkbytes = align256((size_t)(p.bw < kMaxBW ? p.bw : kMaxBW) * p.np * sizeof(float));
    if (o) { o->gk = (float *)(c + off); o->gkstride = gkbytes / sizeof(float); }
    off += gkbytes * batch;
    if (o) o->submap = (int *)(c + off);
    off += ibytes;
    if (o) o->rowsrc[0] = (int *)(c + off);
    off += ibytes;
    if (o) o->rowsrc[1] = (int *)(c + off);
    off += ibytes;
    if (o) o->orig = (int *)(c + off);
    off += ibytes;
    if (o) o->invp = (int *)(c + off);
    off += ibytes;
    return off;
}
size_t blocked_workspace_bytes(const BlockedPlan &p, int batch) { return blocked_carve(p, batch, nullptr, nullptr); }

// ---- init: A -> diag(A, I) in the first working copy (makeAugmentedMatrix counterpart,
//      mat_inv_32.cpp:177-192) + the compact copy of the first sub-panel's columns ---------
__global__ __launch_bounds__(256) void blocked_init_kernel(const float *__restrict__ in, int n, int np, int ld,
                                                            size_t mstride, float *__restrict__ m0,
                                                            float *__restrict__ pt_all, size_t tstride, int w,
                                                            int *__restrict__ orig, int *__restrict__ submap,
                                                            int *__restrict__ status)
{
    const int b = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i0 = blockIdx.y * 16;
    const float *a = in + (size_t)b * n * n;
    float *m = m0 + (size_t)b * mstride;
    float *pt = pt_all + (size_t)b * tstride;
    if (j < np) {
#pragma unroll 4
        for (int u = 0; u < 16; ++u) {
            const int i = i0 + u;
            if (i >= np) break;
            float v;
            if (i < n && j < n) v = a[(size_t)i * n + j];
            else v = (i == j) ? 1.0f : 0.0f;
            m[(size_t)i * ld + j] = v;
            if (j < w) pt[(size_t)j * np + i] = v;
        }
    }
    if (blockIdx.y == 0 && j < np) {
        orig[(size_t)b * np + j] = j;
        submap[(size_t)b * np + j] = j;  // the panel kernels only ever rewrite the positions at or below their block
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && status) status[b] = MI32_OK;
}

// ---- wave-level arg-max helpers (DPP, no LDS traffic) ----------------------------
// Canonical gfx9 wave64 reduction: quad_perm x2, row_half_mirror, row_mirror, then
// row_bcast15 / row_bcast31 fold the four rows; lane 63 ends up with the total.  Each stage
// is ONE instruction (v_max_u32 / v_min_u32 with a DPP source); hipcc's update_dpp builtin
// emits v_mov_dpp + op + copy per stage, and this chain sits on the critical path of every
// pivot step.  The s_nop covers the VALU-write -> DPP-read hazard (2 wait states), which the
// compiler does not pad inside an asm statement.
#define MI32_DPP_REDUCE(OP, V)                                                        \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
                 "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"     \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"          \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"        \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"        \
                 "s_nop 1"                                                                          \
                 : "+v"(V))
__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    MI32_DPP_REDUCE("v_max_u32_dpp", v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    MI32_DPP_REDUCE("v_min_u32_dpp", v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float lane_bcast(float v, int srclane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srclane));
}

// ---- the panel: W pivot steps on an (np x W) register-resident slab -----------
// Each thread keeps RPT rows of the panel in registers for the whole kernel: row
// CONTENTS never move between threads.  What a row swap changes is only an integer
// label pos[k] = the position (row index of the working matrix) that the content of
// register row k currently occupies:
//   pivotElements (mat_inv_32.cpp:154-173)  ==  exchange of two labels.
// submap[position] = the row of the source copy X whose data now belongs at that
// position tells the rank-k updates where every other column's data still lives.

// Diagnostic builds (tools/panel_probe.hip, -DMI32_STAMPS) record s_memtime at the phase
// boundaries of every step; in the product build the macro expands to nothing.
#ifdef MI32_STAMPS
#define MI32_STAMP(step, slot_)                                                           \
    do {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if (stamp_buf && threadIdx.x == blockDim.x - 64 && blockIdx.x == 0)               \
            stamp_buf[(step) * 8 + (slot_)] = __builtin_amdgcn_s_memtime();               \
        __builtin_amdgcn_sched_barrier(0);                                                \
    } while (0)
#define MI32_STAMP_PARAM , unsigned long long *stamp_buf
#define MI32_STAMP_ARG , stamp_buf
#else
#define MI32_STAMP(step, slot_) do { } while (0)
#define MI32_STAMP_PARAM
#define MI32_STAMP_ARG
#endif

template <int NW, int W>
struct __attribute__((aligned(16))) PanelShared {
    float cand[NW][W];          // per wave: its best candidate row as found (wave-private scratch)
    float prn[2][NW][W];        // per step parity, per wave: that row NORMALISED (candidate pivot row)
    unsigned long long key[W];  // one cross-wave arg-max word per step, zeroed at kernel start
    float prn_all[W][W];        // the normalised pivot row of every step, exported for the rows above the block
};

// which matrix row register row k of thread tid holds: V consecutive rows per thread so that the
// compact panel is loaded and stored with one 4*V-byte access per column
template <int NT, int RPT>
__device__ __forceinline__ int panel_row(int tid, int k)
{
    constexpr int V = RPT < 4 ? RPT : 4;
    return (k / V) * (V * NT) + V * tid + (k % V);
}

// One pivot step (column c0 + R of the working matrix, R a compile-time constant) with ONE workgroup
// barrier.
//
// The step is a chain of dependent, mostly scalar and cross-lane operations executed by in-order waves:
// s_memtime stamps (tools/panel_probe.hip) show ~3500 cycles per step even with ONE wave per SIMD, of
// which the 16 FMAs per row are ~5 %.  What a step costs is the NUMBER of instructions every wave runs
// between two barriers, so the step is written to be short rather than clever:
//  * ONE pass over the lane's rows finds its best candidate under the exact order of the reference's scan
//    (largest |a|, lowest position among equals; mat_inv_32.cpp:121-127): a 64-bit comparison of
//    {bits(|a|), ~position}.  The whole state of a row is ONE register npl[k]: ~position (top bit set)
//    while the row can still be chosen, its position itself (top bit clear) once it cannot -- rows above
//    the block, rows already used as a pivot in this panel, rows beyond the matrix.  A dead row's |a| key
//    is masked to 0 and its small npl loses every tie against a live row;
//  * one DPP max over the 32-bit |a| keys, one ballot; only a genuine tie between lanes pays for a second
//    DPP reduction over the positions;
//  * every wave SPECULATES: the lane that holds the wave's best candidate writes that row to LDS, lanes
//    0..W-1 divide one element each by the candidate's pivot-column entry (IEEE division, the identity
//    column's entry becomes 1/pivot) and publish the NORMALISED row next to a 64-bit arg-max key
//    (ds_max_u64).  After the single barrier the key's low bits name the winning wave and its row is read
//    straight from LDS: no second barrier and no division on the post-barrier path;
//  * pivotElements (mat_inv_32.cpp:154-173) is an exchange of two position labels, done branch-free by
//    every lane (no table of who holds which position);
//  * a NaN is never special-cased in the search: its bit pattern wins the unsigned max, the step then has
//    a NaN pivot and the winning wave flags the matrix as singular -- the result is poisoned either way.
template <int NT, int RPT, int W, int R>
__device__ __forceinline__ void panel_step(float (&a)[RPT][W], unsigned (&npl)[RPT], PanelShared<NT / 64, W> &sh,
                                           int lane, int wave_u, int c0, bool wave_active,
                                           bool &singular MI32_STAMP_PARAM)
{
    constexpr int par = R & 1;
    const int slot = c0 + R;

    // -- maxPivot over this lane's rows
    MI32_STAMP(R, 0);
    float col[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) col[k] = a[k][R];
    int own_lane = -1, own_k = 0;
    if (wave_active) {
        unsigned mkey = 0u, mnp = 0u;
        int kb = 0;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const unsigned lm = (unsigned)((int)npl[k] >> 31);                    // all ones while live
            const unsigned key = __float_as_uint(col[k]) & lm & 0x7fffffffu;
            const bool better = (((unsigned long long)key << 32) | npl[k]) > (((unsigned long long)mkey << 32) | mnp);
            mkey = better ? key : mkey;
            mnp = better ? npl[k] : mnp;
            kb = better ? k : kb;
        }
        MI32_STAMP(R, 1);
        const unsigned wm = wave_max_u32(mkey);
        // lanes that hold the wave maximum and a real candidate: almost always exactly one
        unsigned long long hit = __ballot(mkey == wm && (int)mnp < 0);
        MI32_STAMP(R, 2);
        if (hit != 0ull) {  // this wave has a candidate
            if ((hit & (hit - 1ull)) != 0ull) {  // tie between lanes: lowest position = largest ~position
                const unsigned hv = (mkey == wm && (int)mnp < 0) ? mnp : 0u;
                const unsigned hmax = wave_max_u32(hv);  // all lanes take part: never under a lane condition
                hit = __ballot(hv == hmax);              // hmax != 0: at least two lanes hold a live candidate
            }
            own_lane = __ffsll((long long)hit) - 1;
            own_k = __builtin_amdgcn_readlane(kb, own_lane);
            const unsigned wi = ~(unsigned)__builtin_amdgcn_readlane((int)mnp, own_lane);
            // the candidate row, as found, into this wave's scratch slot (the holder lane writes it)
#pragma unroll
            for (int k = 0; k < RPT; ++k)
                if (own_k == k) {
                    if (lane == own_lane) {
#pragma unroll
                        for (int c = 0; c < W; c += 4)
                            *reinterpret_cast<float4 *>(&sh.cand[wave_u][c]) =
                                make_float4(a[k][c], a[k][c + 1], a[k][c + 2], a[k][c + 3]);
                    }
                }
            // fixRow, speculatively: lanes 0..W-1 divide one element each (IEEE); identity entry -> 1/piv.
            // One wave's LDS operations execute in order, so no s_barrier is needed between the holder
            // lane's store and these loads -- only the compiler must not reorder here.
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float cpiv = sh.cand[wave_u][R];
            const float num = (lane < W) ? ((lane == R) ? 1.0f : sh.cand[wave_u][lane]) : 0.0f;
            const float qv = num / cpiv;
            if (lane < W) sh.prn[par][wave_u][lane] = qv;
            if (lane == 0)
                atomicMax(&sh.key[R], ((unsigned long long)wm << 32) |
                                          (unsigned long long)(((0xFFFFFu - wi) << 8) | (unsigned)wave_u));
        }
    }
    MI32_STAMP(R, 3);
    __syncthreads();
    MI32_STAMP(R, 4);
    const unsigned long long key = sh.key[R];
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(key & 0xFFFFFFFFull));
    const int p = (int)(0xFFFFFu - (lo >> 8));
    const int wv = (int)(lo & 0xFFu);
    if (key == 0ull) singular = true;  // cannot happen (position `slot` is always a live candidate); never trust it
    float prn[W];  // prn[R] = 1/piv (the identity column's entry), prn[c] = normalised pivot row
#pragma unroll
    for (int c = 0; c < W; c += 4) {
        const float4 t = *reinterpret_cast<const float4 *>(&sh.prn[par][wv][c]);
        prn[c] = t.x; prn[c + 1] = t.y; prn[c + 2] = t.z; prn[c + 3] = t.w;
    }

    MI32_STAMP(R, 5);
    // -- fixColumn on the slab, branch-free; the pivot column holds the implicit identity column, whose
    //    entry is 0 in every row but the pivot row.  The pivot row itself is overwritten right after.
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const float f = col[k];
#pragma unroll
        for (int c = 0; c < W; ++c)
            a[k][c] = (c == R) ? __builtin_fmaf(-f, prn[R], 0.0f) : __builtin_fmaf(-f, prn[c], a[k][c]);
    }
    MI32_STAMP(R, 6);
    // -- pivotElements == exchange of two position labels: the row that held `slot` takes p ...
    if (p != slot) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) npl[k] = (npl[k] == ~(unsigned)slot) ? ~(unsigned)p : npl[k];
    }
    // ... and the winner's candidate row (its wave knows lane and row) becomes the pivot row: normalised
    // values, label `slot`, no longer a candidate
    if (wave_u == wv) {
        // this wave's scratch slot still holds the winning row as found: its pivot entry decides "singular"
        const float cpiv = sh.cand[wave_u][R];
        if (cpiv == 0.0f || cpiv != cpiv) singular = true;
        if (lane < W) sh.prn_all[R][lane] = sh.prn[par][wv][lane];
#pragma unroll
        for (int k = 0; k < RPT; ++k)
            if (own_k == k) {
                if (lane == own_lane) {
#pragma unroll
                    for (int c = 0; c < W; ++c) a[k][c] = prn[c];
                    npl[k] = (unsigned)slot;
                }
            }
    }
    MI32_STAMP(R, 7);
}

template <int NT, int RPT, int W, int... Rs>
__device__ __forceinline__ void panel_steps(float (&a)[RPT][W], unsigned (&npl)[RPT], PanelShared<NT / 64, W> &sh,
                                            int lane, int wave_u, int c0, bool wave_active,
                                            bool &singular MI32_STAMP_PARAM, std::integer_sequence<int, Rs...>)
{
    (panel_step<NT, RPT, W, Rs>(a, npl, sh, lane, wave_u, c0, wave_active, singular MI32_STAMP_ARG), ...);
}

// Handles the rows [row_lo, np) of the sub-panel -- the rows that can still be chosen as pivots (row_lo = c0).
// The rows above the block never take part in a search and never move: their part of G_s follows from the
// W normalised pivot rows alone, which this kernel exports (prn_out) and the in-block update kernel applies
// with the same fmaf sequence (gj_rank_update_kernel, COMPACT_G).
template <int NT, int RPT, int W>
__global__ __launch_bounds__(NT) void gj_panel_kernel(const float *__restrict__ pt_all, float *__restrict__ gt_all,
                                                       int np, int n, size_t tstride, int c0, int row_lo,
                                                       int *__restrict__ submap_all, int *__restrict__ rowsrc_all,
                                                       int *__restrict__ orig_all, int first_in_block,
                                                       float *__restrict__ prn_out_all,
                                                       int *__restrict__ status MI32_STAMP_PARAM)
{
    constexpr int V = RPT < 4 ? RPT : 4;
    typedef float vecV __attribute__((ext_vector_type(V)));
    __shared__ PanelShared<NT / 64, W> sh;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float *pt = pt_all + (size_t)b * tstride;
    float *gt = gt_all + (size_t)b * tstride;
    if (tid < W) sh.key[tid] = 0ull;

    float a[RPT][W];
    unsigned npl[RPT];
#pragma unroll
    for (int g = 0; g < RPT / V; ++g) {
        const int row = row_lo + panel_row<NT, RPT>(tid, g * V);  // first of V consecutive rows
#pragma unroll
        for (int c = 0; c < W; ++c) {
            vecV v;
            if (row < np) v = *reinterpret_cast<const vecV *>(pt + (size_t)c * np + row);
            else v = (vecV)(0.0f);
#pragma unroll
            for (int j = 0; j < V; ++j) a[g * V + j][c] = v[j];
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int r = row + j;
            // A candidate is a row of the matrix at or below the block.  A real column (slot < n) may only take
            // its pivot from the real rows: the identity padding holds exact zeros there, which can tie only
            // with an all-zero column, and then the lowest position -- a real row -- wins the tie.
            // Rows beyond the matrix (r >= np) are dead and are never written back.
            npl[g * V + j] = (r >= c0 && r < np) ? ~(unsigned)r : (unsigned)r;
        }
    }
    // the row maps this kernel will permute: fetched now, so their latency hides behind the steps,
    // and parked in thread-private LDS slots (the 1024-thread instances have no registers to spare)
    int *submap = submap_all + (size_t)b * np;
    int *rowsrc = rowsrc_all + (size_t)b * np;
    int *orig = orig_all + (size_t)b * np;
    extern __shared__ int s_park[];  // [2][RPT][NT]
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int row = row_lo + panel_row<NT, RPT>(tid, k);
        s_park[k * NT + tid] = (first_in_block || row >= np) ? row : rowsrc[row];  // composite map so far
        s_park[(RPT + k) * NT + tid] = row < np ? orig[row] : 0;
    }
    // rows above the block keep their place: identity entries in the maps the update kernels read
    if (first_in_block)
        for (int i = tid; i < row_lo; i += NT) rowsrc[i] = i;
    // the positions the previous sub-panel retired (at most kMaxW of them; identity is right for every earlier one)
    if (tid < kMaxW && c0 - kMaxW + tid >= 0) submap[c0 - kMaxW + tid] = c0 - kMaxW + tid;
    bool singular = false;
    __syncthreads();  // sh.key[] zeroed before any wave's first atomicMax; all map reads issued
    // a wave takes part in the pivot search only if at least one of its rows lies in or below the block
    const int wave_last_tid = (__builtin_amdgcn_readfirstlane(tid) | 63);
    const bool wave_active = row_lo + panel_row<NT, RPT>(wave_last_tid, RPT - 1) >= c0;
    panel_steps<NT, RPT, W>(a, npl, sh, lane, wave_u, c0, wave_active, singular MI32_STAMP_ARG,
                            std::make_integer_sequence<int, W>{});
    int pos[RPT];  // final position of every register row
#pragma unroll
    for (int k = 0; k < RPT; ++k) pos[k] = (int)(npl[k] ^ (unsigned)((int)npl[k] >> 31));

    // -- the W normalised pivot rows, for the rows above the block
    __syncthreads();
    for (int i = tid; i < W * W; i += NT) prn_out_all[(size_t)b * (kMaxW * kMaxW) + i] = sh.prn_all[i / W][i % W];
    // -- G_s, compact and by register row (coalesced); the row maps, by position
#pragma unroll
    for (int g = 0; g < RPT / V; ++g) {
        const int row = row_lo + panel_row<NT, RPT>(tid, g * V);
        if (row < np) {
#pragma unroll
            for (int c = 0; c < W; ++c) {
                vecV v;
#pragma unroll
                for (int j = 0; j < V; ++j) v[j] = a[g * V + j][c];
                *reinterpret_cast<vecV *>(gt + (size_t)c * np + row) = v;
            }
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const int k = g * V + j;
                submap[pos[k]] = row + j;  // position pos[k] now holds X's row (row + j)
            }
        }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        if (row_lo + panel_row<NT, RPT>(tid, k) < np) {
            rowsrc[pos[k]] = s_park[k * NT + tid];
            orig[pos[k]] = s_park[(RPT + k) * NT + tid];
        }
    }
    // only the wave that won a step has looked at that step's pivot: any wave may raise the flag
    if (singular && lane == 0 && status) status[b] = MI32_SINGULAR;
}

// One pivot step of a row that is not a candidate, for the in-block update kernel: the row's BK panel entries
// are spread over the 4 threads of a quad (BK/4 consecutive columns each); its current entry in column R
// lives in thread R / (BK/4) and is broadcast with one quad_perm DPP move.
template <int BK, int R>
__device__ __forceinline__ void above_rows_step(float (&v)[BK / 4], const float *s_prn, int q4)
{
    constexpr int CPT = BK / 4;
    constexpr int kQuad = (R / CPT) * 0x55;  // quad_perm:[q,q,q,q]
    const float f = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[R % CPT]), kQuad, 0xf, 0xf, false));
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = q4 * CPT + j;
        const float base = (c == R) ? 0.0f : v[j];
        v[j] = __builtin_fmaf(-f, s_prn[R * BK + c], base);
    }
}
template <int BK, int... Rs>
__device__ __forceinline__ void above_rows_steps(float (&v)[BK / 4], const float *s_prn, int q4,
                                                 std::integer_sequence<int, Rs...>)
{
    (above_rows_step<BK, Rs>(v, s_prn, q4), ...);
}

// ---- rank-k update on the fp32 matrix cores ------------------------------------
//   dst[i][j] = (i in [c0,c0+kdim) ? 0 : src[map[i]][j]) + sum_k G[i][k] * src[map[c0+k]][j]
// for the columns j of this tile that are not panel columns.  256 threads = 4
// waves in a 2x2 arrangement; each wave owns (BM/2)x(BN/2) as 32x32 MFMA tiles.
// A = G is staged into LDS as [k][row] so that the 32 lanes of a half-wave read 32
// consecutive floats; B (= pivot rows, row-major) is staged as it lies.  One
// accumulation chain per output element, k ascending: bit-for-bit the fmaf chain
// of oracle/gj_oracle.c's blocked restatement.
//
// COMPACT_G (the in-block update, kdim == BK == w): G_s comes from the panel
//   kernel's compact output gt[k][map[i]]; the column-tile-0 workgroups also
//   materialise it into dst[i][c0 + k] (row-major), where every later update
//   expects it.
// !COMPACT_G (the rank-bw update): G is the block's panel, row-major in g_all.
// Either flavour exports the columns [pt_col, pt_col + pt_w) it has just
// computed into the compact transposed panel pt_out (the next sub-panel's input).
template <int BM, int BN, int BK, bool COMPACT_G>
__global__ __launch_bounds__(256) void gj_rank_update_kernel(const float *__restrict__ src_all,
                                                              float *__restrict__ dst_all,
                                                              const float *__restrict__ g_all, size_t gstride,
                                                              int np, int ld, size_t mstride, int c0, int kdim,
                                                              int col_lo, const int *__restrict__ map_all,
                                                              int copy_panel, float *__restrict__ pt_out_all,
                                                              size_t tstride, int pt_col, int pt_w, int skip_lo,
                                                              int skip_hi, const float *__restrict__ pt_in_all,
                                                              const float *__restrict__ prn_all, int above_hi)
{
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int PADA = (32 / BK) > 0 ? (32 / BK) : 1;
    constexpr int LDA = BM + PADA;
    constexpr int LDB = BN + 4;
    __shared__ float s_a[BK * LDA];
    __shared__ __attribute__((aligned(16))) float s_b[BK * LDB];
    __shared__ int s_map[BM];
    __shared__ __attribute__((aligned(16))) float s_prn[COMPACT_G ? BK * BK : 4];  // the sub-panel's normalised pivot rows

    const int b = blockIdx.z;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int row0 = blockIdx.y * BM;
    const int col0 = col_lo + blockIdx.x * BN;
    const float *src = src_all + (size_t)b * mstride;
    float *dst = dst_all + (size_t)b * mstride;
    const float *g = g_all + (size_t)b * gstride;
    const int *map = map_all + (size_t)b * np;
    float *pt_out = pt_out_all + (size_t)b * tstride;

    if (col0 >= skip_lo && col0 < skip_hi) return;  // these columns belong to the other half of a split update
    if (!COMPACT_G && col0 >= c0 && col0 + BN <= c0 + kdim) {
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
    if constexpr (COMPACT_G) {
        if (row0 < above_hi)  // some of this tile's rows lie above the block: their G_s is computed here
            for (int i = tid; i < BK * BK; i += 256) s_prn[i] = prn_all[(size_t)b * (kMaxW * kMaxW) + i];
    }
    __syncthreads();

    // accumulators start from the (row-mapped) old values; rows of the block start from 0
    float16v acc[TM][TN], cin[TM][TN];
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
                const float cval = in_block ? 0.0f : src[(size_t)s_map[lr] * ld + col];
                if constexpr (COMPACT_G) {
                    acc[tm][tn][reg] = cval;  // in-block update: the chain starts from the old value
                } else {
                    cin[tm][tn][reg] = cval;  // rank-bw update: sum from zero, old value added at the end
                    acc[tm][tn][reg] = 0.0f;
                }
            }
        }

    for (int kt = 0; kt < kdim; kt += BK) {
        if constexpr (COMPACT_G) {
            // stage A = G_s, [k][row]; 4 threads per row, BK/4 columns each (kdim == BK, kt == 0).
            //  * rows at or below the block: the panel kernel's compact output gt[k][map[row]];
            //  * rows above the block (never candidates, never moved): the panel kernel did not touch them.
            //    Their G_s is the row's W entries pt_in[k][row] taken through the W pivot steps with the
            //    exported normalised pivot rows -- fixColumn (mat_inv_32.cpp:28-38) on one row, the very
            //    fmaf sequence the panel kernel applies to a row that is not a candidate.
            static_assert(BM * 4 == 256 && BK % 4 == 0, "4 threads per row");
            constexpr int CPT = BK / 4;
            const int rr = tid >> 2, q4 = tid & 3;
            const int grow = row0 + rr;
            float v[CPT];
            if (grow >= above_hi) {
#pragma unroll
                for (int j = 0; j < CPT; ++j) v[j] = g[(size_t)(q4 * CPT + j) * np + s_map[rr]];
            } else {
                const float *pt_in = pt_in_all + (size_t)b * tstride;
#pragma unroll
                for (int j = 0; j < CPT; ++j) v[j] = pt_in[(size_t)(q4 * CPT + j) * np + grow];
                above_rows_steps<BK>(v, s_prn, q4, std::make_integer_sequence<int, BK>{});
            }
#pragma unroll
            for (int j = 0; j < CPT; ++j) s_a[(q4 * CPT + j) * LDA + rr] = v[j];
        } else {
            // stage A: BM x BK of the row-major panel, transposed
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
        if constexpr (COMPACT_G) {
            // materialise G_s into the row-major working copy (column tile 0 only; kdim == BK)
            if (blockIdx.x == 0) {
                for (int idx = tid; idx < BM * (BK / 4); idx += 256) {
                    const int rr = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
                    *reinterpret_cast<float4 *>(dst + (size_t)(row0 + rr) * ld + c0 + k4) =
                        make_float4(s_a[(k4 + 0) * LDA + rr], s_a[(k4 + 1) * LDA + rr], s_a[(k4 + 2) * LDA + rr],
                                    s_a[(k4 + 3) * LDA + rr]);
                }
            }
        }
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
            if (col >= c0 && col < c0 + kdim) continue;  // panel column: holds G, not an update result
            const bool exp = (col >= pt_col && col < pt_col + pt_w);  // next sub-panel's column
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int grow = row0 + wr * WM + tm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhalf;
                float v = acc[tm][tn][reg];
                if constexpr (!COMPACT_G) v += cin[tm][tn][reg];
                dst[(size_t)grow * ld + col] = v;
                if (exp) pt_out[(size_t)(col - pt_col) * np + grow] = v;
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
// Whole rows go through LDS: the global read (all np columns of R rows) and the global write (n columns)
// are both coalesced; the column gather happens inside LDS.  (A direct gather from global memory read
// 4 scattered bytes per lane: 1.35 ms for 64 x 2048^2, i.e. 1.5 TB/s.)
__global__ __launch_bounds__(256) void unpermute_columns_ld_kernel(const float *__restrict__ w_all, int ld, int np,
                                                                    size_t wstride, const int *__restrict__ invp,
                                                                    int istride, int n, int rows_per_block,
                                                                    float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float s_rows[];  // [rows_per_block][np]
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const float *w = w_all + (size_t)b * wstride;
    float *o = out + (size_t)b * n * n;
    const int i0 = blockIdx.x * rows_per_block;
    const int nr = (n - i0 < rows_per_block) ? (n - i0) : rows_per_block;
    for (int r = 0; r < nr; ++r)
        for (int c4 = tid * 4; c4 < np; c4 += 1024)
            *reinterpret_cast<float4 *>(&s_rows[(size_t)r * np + c4]) =
                *reinterpret_cast<const float4 *>(w + (size_t)(i0 + r) * ld + c4);
    __syncthreads();
    for (int j = tid; j < n; j += 256) {
        const int c = invp[(size_t)b * istride + j];
        for (int r = 0; r < nr; ++r) o[(size_t)(i0 + r) * n + j] = s_rows[(size_t)r * np + c];
    }
}

#ifndef MI32_STAMPS
template <int NT, int RPT, int W>
static void launch_panel(const BlockedPlan &p, const BlockedWs &ws, int c0, int sub, int *rowsrc, int batch,
                         int *d_status, hipStream_t stream)
{
    hipLaunchKernelGGL((gj_panel_kernel<NT, RPT, W>), dim3(batch), dim3(NT), 2 * RPT * NT * sizeof(int), stream,
                       ws.pt[sub & 1], ws.gt, p.np, p.n, ws.tstride, c0, c0, ws.submap, rowsrc, ws.orig, sub == 0,
                       ws.prn, d_status);
}

// The panel kernel handles the np - c0 rows at or below the block: the smallest thread geometry that holds
// them (fewer waves and fewer rows per lane both shorten a pivot step).
static bool dispatch_panel(const BlockedPlan &p, const BlockedWs &ws, int w, int c0, int sub, int *rowsrc, int batch,
                           int *d_status, hipStream_t stream)
{
    int nt, rpt;
    panel_geometry(p, p.np - c0, nt, rpt);
#define MI32_PANEL_CASE(T, R, WW)                                                  \
    if (nt == T && rpt == R && w == WW) {                                          \
        launch_panel<T, R, WW>(p, ws, c0, sub, rowsrc, batch, d_status, stream);   \
        return true;                                                               \
    }
    MI32_PANEL_CASE(256, 1, 32) MI32_PANEL_CASE(256, 1, 16) MI32_PANEL_CASE(256, 1, 8) MI32_PANEL_CASE(256, 1, 4)
    MI32_PANEL_CASE(512, 1, 32) MI32_PANEL_CASE(512, 2, 32) MI32_PANEL_CASE(512, 4, 32) MI32_PANEL_CASE(1024, 1, 32)
    MI32_PANEL_CASE(1024, 2, 32)
    MI32_PANEL_CASE(512, 1, 16) MI32_PANEL_CASE(512, 2, 16) MI32_PANEL_CASE(512, 4, 16) MI32_PANEL_CASE(512, 8, 16)
    MI32_PANEL_CASE(512, 1, 8) MI32_PANEL_CASE(512, 2, 8) MI32_PANEL_CASE(512, 4, 8) MI32_PANEL_CASE(512, 8, 8)
    MI32_PANEL_CASE(512, 1, 4) MI32_PANEL_CASE(512, 2, 4) MI32_PANEL_CASE(512, 4, 4) MI32_PANEL_CASE(512, 8, 4)
    MI32_PANEL_CASE(1024, 1, 16) MI32_PANEL_CASE(1024, 2, 16) MI32_PANEL_CASE(1024, 4, 16)
    MI32_PANEL_CASE(1024, 1, 8) MI32_PANEL_CASE(1024, 2, 8) MI32_PANEL_CASE(1024, 4, 8) MI32_PANEL_CASE(1024, 8, 8)
    MI32_PANEL_CASE(1024, 1, 4) MI32_PANEL_CASE(1024, 2, 4) MI32_PANEL_CASE(1024, 4, 4) MI32_PANEL_CASE(1024, 8, 4)
    MI32_PANEL_CASE(1024, 16, 4)
#undef MI32_PANEL_CASE
    return false;
}

// in-block update: columns [C0, C0+kb) of the block, K = w, G_s from the compact panel; exports the
// next sub-panel's columns (if it lies in this block) into pt
static void launch_inner_update(const BlockedPlan &p, const BlockedWs &ws, int w, const float *x, float *y, int c0,
                                int C0, int kb, int sub, int batch, hipStream_t stream)
{
    const dim3 grid(kb / 64, p.np / 64, batch);
    const int next = c0 + w;
    const int pt_col = (next < C0 + kb) ? next : -(1 << 30);
    // reads sub-panel `sub`'s input panel (rows above the block) and writes the next one's: two buffers
#define MI32_INNER(BKV)                                                                                             \
    hipLaunchKernelGGL((gj_rank_update_kernel<64, 64, BKV, true>), grid, dim3(256), 0, stream, x, y, ws.gt,          \
                       ws.tstride, p.np, p.ld, ws.mstride, c0, BKV, C0, ws.submap, 0, ws.pt[(sub + 1) & 1], ws.tstride, \
                       pt_col, w, 0, 0, ws.pt[sub & 1], ws.prn, c0)
    if (w == 32) MI32_INNER(32);
    else if (w == 16) MI32_INNER(16);
    else if (w == 8) MI32_INNER(8);
    else MI32_INNER(4);
#undef MI32_INNER
}

// Look-ahead: the rank-bw update of block b is split into (A) the columns of block b+1, which the next
// panel phase needs at once, and (B) all other columns.  (A) stays on the main stream; (B) runs on a
// second stream and overlaps with block b+1's panel phase, which is latency bound on a few CUs.  The
// next rank-bw update (and the final un-permutation) wait for (B) through an event.
hipError_t blocked_invert(const BlockedPlan &p, const float *d_a, float *d_inv, int batch, int *d_status, void *wsp,
                          const BlockedExec &ex)
{
    BlockedWs ws;
    blocked_carve(p, batch, wsp, &ws);
    const int np = p.np;
    hipStream_t stream = ex.stream;
    Profiler *prof = ex.prof;
    // look-ahead pays when the GPU is otherwise idle during the panel phase: a single large matrix
    // (measured: 8192^2 51 -> 45 ms, 16384^2 399 -> 330 ms, 4096^2 11.2 -> 11.0 ms, 2048^2 4.2 -> 4.4 ms)
    const bool lookahead = ex.aux != nullptr && ex.n_events >= 4 && ex.aux_workgroups > 0 && batch == 1 && np >= 4096;
    hipError_t e;
    {
        ProfScope ps(prof, KC_INIT, stream);
        hipLaunchKernelGGL(blocked_init_kernel, dim3((np + 255) / 256, (np + 15) / 16, batch), dim3(256), 0, stream,
                           d_a, p.n, np, p.ld, ws.mstride, ws.m0, ws.pt[0], ws.tstride, p.wblk[0], ws.orig, ws.submap, d_status);
    }
    if (lookahead) {  // whatever still runs on the second stream from an earlier call shares this workspace
        if ((e = hipEventRecord(ex.events[0], ex.aux)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(stream, ex.events[0], 0)) != hipSuccess) return e;
    }
    // dynamic LDS of the rank-bw kernels: operand stages + maps; the persistent flavour asks for more than half
    // a CU's LDS so that at most one of its workgroups is resident per CU
    const size_t lds_persistent = 84 * 1024;
    {
        static bool attr_set_dev[64] = {};  // function attributes are per device
        int dev = 0;
        (void)hipGetDevice(&dev);
        bool &attr_set = attr_set_dev[dev & 63];
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)gj_rank_bw2_kernel<MI32_BW_BK, MI32_BW_WPS>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)rank_bw2_lds_bytes<MI32_BW_BK>(kMaxBW));
            (void)hipFuncSetAttribute((const void *)gj_rank_bw2_persistent_kernel<MI32_BW_BK>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_persistent);
            attr_set = true;
        }
    }
    float *cur = ws.m0, *oth = ws.m1;
    bool pending_b = false;  // a (B) half is in flight on the second stream
    int blk = 0, ev = 0;
    for (int C0 = 0; C0 < np; C0 += p.bw, ++blk) {
        const int kb = (C0 + p.bw <= np) ? p.bw : np - C0;
        int *rowsrc = ws.rowsrc[blk & 1];
        const int w = p.wblk[blk];                                       // sub-panel width of this block
        const int w_next = (blk + 1 < p.nblk) ? p.wblk[blk + 1] : w;     // ... and of the next one
        float *x = cur, *y = oth;  // the block's panel columns alternate between the two copies
        for (int s = 0; s * w < kb; ++s) {
            const int c0 = C0 + s * w;
            {
                ProfScope ps(prof, KC_PANEL, stream);
                if (!dispatch_panel(p, ws, w, c0, s, rowsrc, batch, d_status, stream)) return hipErrorInvalidValue;
            }
            {
                ProfScope ps(prof, KC_UPDATE_IN, stream);
                launch_inner_update(p, ws, w, x, y, c0, C0, kb, s, batch, stream);
            }
            float *t = x; x = y; y = t;
        }
        // x now holds the block's G; every other column is still valid in `cur` only
        if (kb < np) {
            const int next = C0 + kb;  // first column of the next block
            const bool has_next = next < np;
            const int kb_next = has_next ? ((next + p.bw <= np) ? p.bw : np - next) : 0;
            const int pt_col = has_next ? next : -(1 << 30);
            const int copy = (x != oth) ? 1 : 0;
            if (pending_b) {  // this update reads all of `cur` and overwrites `oth`: the previous (B) must be done
                if ((e = hipStreamWaitEvent(stream, ex.events[ev], 0)) != hipSuccess) return e;
                pending_b = false;
            }
            {   // A operand of the rank-bw update: the block's panel, k-major (mi32_rank_bw.h)
                ProfScope ps(prof, KC_TRANSPOSE, stream);
                hipLaunchKernelGGL(gj_panel_transpose_kernel, dim3(np / 64, kb / 64, batch), dim3(256), 0, stream, x,
                                   ws.mstride, np, p.ld, C0, ws.gk, ws.gkstride);
            }
            if (lookahead && has_next) {
                {   // (A): the next block's columns, on the main stream; exports the next sub-panel
                    ProfScope ps(prof, KC_UPDATE_OUT, stream);
                    // small tiles: only kb_next columns, so 64x64 gives 4x the workgroups of 128x128
                    hipLaunchKernelGGL((gj_rank_update_kernel<64, 64, 32, false>), dim3(kb_next / 64, np / 64, batch),
                                       dim3(256), 0, stream, cur, oth, x, ws.mstride, np, p.ld, ws.mstride, C0, kb, next,
                                       rowsrc, copy, ws.pt[0], ws.tstride, pt_col, w_next, 0, 0, nullptr, nullptr, 0);
                }
                // (B): everything else, on the second stream, after this block's panel phase
                ev = (ev + 1) % (ex.n_events / 2);
                hipEvent_t e_panel = ex.events[ex.n_events / 2 + ev];
                if ((e = hipEventRecord(e_panel, stream)) != hipSuccess) return e;
                if ((e = hipStreamWaitEvent(ex.aux, e_panel, 0)) != hipSuccess) return e;
                {
                    ProfScope ps(prof, KC_UPDATE_OUT, ex.aux);
                    // persistent flavour: aux_workgroups (< number of CUs) workgroups, with so much dynamic LDS
                    // that one CU holds at most one of them -> the remaining CUs stay free for the main stream
                    hipLaunchKernelGGL((gj_rank_bw2_persistent_kernel<MI32_BW_BK>), dim3(ex.aux_workgroups, batch),
                                       dim3(256), lds_persistent, ex.aux, cur, oth, x, ws.mstride, ws.gk, ws.gkstride, np,
                                       p.ld, ws.mstride, C0, kb, rowsrc, copy, ws.pt[0], ws.tstride, -(1 << 30), w_next, next,
                                       next + kb_next);
                }
                if ((e = hipEventRecord(ex.events[ev], ex.aux)) != hipSuccess) return e;
                pending_b = true;
            } else {
                ProfScope ps(prof, KC_UPDATE_OUT, stream);
                hipLaunchKernelGGL((gj_rank_bw2_kernel<MI32_BW_BK, MI32_BW_WPS>), dim3((np / 128) * (np / 128), batch),
                                   dim3(256), rank_bw2_lds_bytes<MI32_BW_BK>(kb), stream, cur, oth, x, ws.mstride, ws.gk,
                                   ws.gkstride, np, p.ld, ws.mstride, C0, kb, rowsrc, copy, ws.pt[0], ws.tstride, pt_col,
                                   w_next, 0, 0);
            }
            float *t = cur; cur = oth; oth = t;
        } else {
            cur = x;  // single block: the panel is the whole matrix
        }
    }
    if (pending_b) {
        if ((e = hipStreamWaitEvent(stream, ex.events[ev], 0)) != hipSuccess) return e;
    }
    ProfScope ps(prof, KC_FINISH, stream);
    // over ALL np entries: orig is a permutation of [0, np), so every invp[j] is defined and in range
    hipLaunchKernelGGL(invert_perm_ld_kernel, dim3((np + 255) / 256, batch), dim3(256), 0, stream, ws.orig, ws.invp,
                       np, np);
    {
        int rpb = (64 * 1024) / (np * (int)sizeof(float));  // rows per workgroup: at most 64 KiB of LDS
        if (rpb < 1) rpb = 1;
        if (rpb > 8) rpb = 8;
        hipLaunchKernelGGL(unpermute_columns_ld_kernel, dim3((p.n + rpb - 1) / rpb, batch), dim3(256),
                           (size_t)rpb * np * sizeof(float), stream, cur, p.ld, np, ws.mstride, ws.invp, np, p.n, rpb,
                           d_inv);
    }
    return hipGetLastError();
}
#endif  // !MI32_STAMPS

}  // namespace mi32
