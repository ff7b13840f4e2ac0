// mi32_blocked.hip -- blocked Gauss-Jordan with delayed rank-k updates on the
// fp32 matrix cores (gfx950: v_mfma_f32_32x32x2_f32).
//
// Same elimination as mi32_sweep.hip (and as the reference's step loop,
// /root/reference/Matlab/mat_inv_32/mat_inv_32/mat_inv_32.cpp:317-362), with the
// column updates of a block of pivots delayed:
//
//   for each outer block K = [C0, C0+kb) of pivot columns:
//     for each sub-panel s, Ks = [c0, c0+W) of K:
//       panel(s)   -- ONE workgroup per matrix holds the rows that can still be
//                     chosen as pivots, W columns each, in registers and runs the W
//                     pivot steps on them: column arg-max (DPP + one LDS atomic), row
//                     swap (= exchange of two position labels), IEEE-division
//                     normalise, eliminate.  Result: G_s = the W transformed columns
//                     (the inverse columns of these pivots).
//       update(s)  -- every other column j of the block (fp32 MFMA, K = W):
//                     M[i][j] = (i in Ks ? 0 : M[src(i)][j]) + sum_k G_s[i][k] * M[src(c0+k)][j]
//     rank-bw update (K = kb, mi32_rank_bw.h) -- every column outside the block, same
//                     formula with the block's composite G and row map.
//
// panel(s) and update(s-1) are ONE launch (gj_subpanel_kernel: workgroup 0 of a
// matrix is the panel, the others are update tiles): update(s-1) no longer sits
// between two panels on the critical path of the N pivot steps.  What panel(s)
// needs from update(s-1) -- its own W columns -- it computes itself in a prologue
// (the same k-ascending fmaf chain the MFMA update runs), from the panel input
// that update(s-2) exported one launch earlier.
//
// Row swaps are never applied as data movement of their own: the updates read
// their C rows and their B (pivot-row) operand THROUGH a row map and write
// out-of-place into the second working copy, so swap + snapshot + update are
// one launch and no launch has a read-after-write hazard between workgroups.
// The two working copies alternate roles exactly like the reference's
// ping-pong buffers (mat_inv_32.cpp:318,353-360).
//
// The panel workgroup only ever touches COMPACT, TRANSPOSED panels: it reads
// Pt[c][row] (W x np, exported by whichever wide kernel produced those columns)
// and writes Gt[c][row], with coalesced accesses.  (Letting the one workgroup
// gather 64-byte chunks of np rows itself cost 18 us per launch, more than its 16
// pivot steps.)
//
// Row orders.  "Order after t" = the rows arranged by the position they hold after
// sub-panel t's swaps.  update(t) writes the working copy and its exports in order
// after t.  panel(s) therefore receives its input Pt_s (exported by update(s-2)) in
// order after s-2 and the previous panel's output Gt_{s-1} in that same order;
// what it needs on top is every row's position after s-1, its label at entry
// (invsub_{s-1}).  It writes Gt_s by those labels, i.e. in order after s-1, which
// is the order update(s) reads the working copy in.
//
// The working matrix is the N x N in-place form (see mi32_sweep.hip), padded
// with an identity block to a multiple of 128 so that no tile needs bounds
// checks: inv(diag(A, I)) = diag(inv(A), I); a real column only ever takes its
// pivot from the real rows, so the padding is never swapped into the matrix.
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <utility>

#include "mi32_internal.h"
#include "mi32_rank_bw.h"

namespace mi32 {

typedef float float16v __attribute__((ext_vector_type(16)));

// k-tile depth and waves/SIMD of the rank-bw update (mi32_rank_bw.h; tunable at build time)
#ifndef MI32_BW_BK
#define MI32_BW_BK 16
#endif
#ifndef MI32_BW_PF
#define MI32_BW_PF 0   // (round 2 fetched the old values of a tile under its last k-tiles; the accumulation from the old
                       // value, round 3, needs them in front of the k-loop)
#endif
#ifndef MI32_BW_WPS
#define MI32_BW_WPS (MI32_BW_PF ? 2 : 3)   // the prefetch keeps 64 more registers live: two workgroups per CU
#endif
static constexpr int kMaxBW = 512;  // widest outer block (rows of the transposed panel Gk)

static constexpr int kMaxW = 32;  // widest sub-panel (columns kept in registers)
// A panel of more than kPanelGroupRows candidate rows is shared by up to kMaxPanelGroups workgroups (one CU
// each, <= 4 rows per lane at 1024 threads) that exchange every step's local winner through global memory.
static constexpr int kMaxPanelGroups = 4;
static constexpr int kPanelGroupRows = 4096;
static constexpr int kXchGranules = 2 * kMaxPanelGroups * 32;  // 8-byte granules per matrix: [parity][group][32]
// how long a workgroup of a shared panel waits for a partner's record before it gives the matrix up
// (MI32_RUNTIME_ERROR, output poisoned with NaN): 0.25 s of the 100 MHz s_memrealtime clock
static constexpr unsigned long long kPanelXchTimeoutTicks = 25000000ull;

// Panel-kernel geometry: NT threads hold the rows at or below the block x w columns in registers, rpt rows
// each (1024 threads leave <= 128 VGPRs per lane, i.e. rpt * w <= 64 floats of slab).
// Thread geometry of a panel launch that holds `nrows` rows: the smallest that fits (fewer waves and fewer
// rows per lane both shorten a pivot step).
static void panel_geometry(const BlockedPlan &p, int nrows, int &nt, int &rpt)
{
    rpt = 1;
    if (p.multi_panel && nrows > kPanelGroupRows) {  // shared by ceil(nrows / 4096) workgroups of 1024 x 4 rows
        nt = 1024;
        rpt = 4;
        return;
    }
    if (nrows <= 256) nt = 256;
    else if (nrows <= 512) nt = 512;
    else {
        nt = p.nthreads_panel;
        while (rpt * nt < nrows) rpt *= 2;
    }
}

BlockedPlan make_blocked_plan(int n, int w, int bw, int batch)
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
    if (rpt > 8 && nt == 512) {  // no 512-thread instance holds more than 8 rows per lane: use 1024
        nt = 1024;
        rpt = 1;
        while (rpt * nt < p.np) rpt *= 2;
    }
    p.nthreads_panel = nt;
    p.rpt = rpt;
    // Multi-workgroup panels need every workgroup of a panel resident at once and a whole CU each; with the
    // look-ahead kernel holding all but 16 (32 below 8192 rows) CUs that is safe for a few matrices (MI32_MULTI_PANEL=0 turns it off).
    p.multi_panel = (nt == 1024 && p.np > kPanelGroupRows && batch * kMaxPanelGroups <= 16) ? 1 : 0;
    if (const char *e = std::getenv("MI32_MULTI_PANEL")) p.multi_panel = p.multi_panel && std::atoi(e) != 0;
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
// (the update tiles address a matrix with 32-bit byte offsets from its base: mi32_rank_bw.h, inblock_update_body)
static_assert(16384ull * (16384 + 64) * sizeof(float) < (1ull << 32), "a working copy must stay below 4 GiB");
bool blocked_supported(int n) { return n > 0 && ((n + 127) & ~127) <= 16384; }

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
struct BlockedWs {
    float *m0, *m1;     // the two working copies, np x ld each
    float *pt[3];       // compact transposed panel inputs, kMaxW x np each: sub-panel s of a block uses pt[s % 3]
    float *gt[2];       // compact transposed panel outputs G_s (the new values of the sub-panel's own columns): gt[s & 1]
    float *mt[2];       // compact transposed MULTIPLIERS of sub-panel s, kMaxW x mtld each: mt[s & 1]
    float *aux[2];      // per sub-panel, kAuxFloats per matrix: the W normalised pivot rows; the previous sub-panel's
                        // pivot rows as its steps saw them (U_{s-1}) restricted to this sub-panel's columns; the W x W
                        // multipliers of the sub-panel's own pivot rows
    unsigned long long *xch;  // exchange granules of the multi-workgroup panels, kXchGranules per matrix
    float *mf[2];       // the block's NEGATED multipliers, np x bw row-major BY BLOCK-START ROW INDEX (never permuted);
                        // double-buffered across blocks (the look-ahead half reads block b's while block b+1 runs)
    float *ub[2];       // the block's pivot rows as their own steps saw them (U), bw x np: B operand of the rank-bw update
    float *xs[2];       // the block's pivot rows after their own sub-panel's last step, bw x np: where their
                        // accumulation starts in the rank-bw update
    float *xst;         // gj_block_strip_kernel: the pivot rows of the groups a call leaves to the next one
    float *gk;          // the block's multipliers transposed and in final row order, bw x np: its A operand
    size_t gkstride;    // floats per matrix in gk / ub
    size_t mfstride;    // floats per matrix in mf
    int *submap[2], *invsub[2];  // per sub-panel: position after s -> index in order after s-1, and its inverse
    int *rowsrc[4], *orig, *invp;  // rowsrc[2 * (blk & 1) + (fused ? s & 1 : 0)]: double-buffered across blocks
                                   // (look-ahead) and, in fused blocks, across sub-panels (update(s-1) reads the map
                                   // panel(s) rewrites in the same launch)
    size_t mstride;     // floats per matrix in m0/m1
    size_t tstride;     // floats per matrix in pt/gt
    size_t mtstride;    // floats per matrix in mt
    int mtld;           // row stride of mt: every register row of a panel workgroup has a slot (rows >= np too)
    size_t pt_bstride;  // floats between pt[i] and pt[i + 1]
};
static constexpr int kAuxFloats = 2 * kMaxW * kMaxW;
static size_t blocked_carve(const BlockedPlan &p, int batch, void *base, BlockedWs *o)
{
    const size_t mbytes = align256((size_t)p.np * p.ld * sizeof(float));
    const size_t tbytes = align256((size_t)kMaxW * p.np * sizeof(float));
    const size_t ibytes = align256((size_t)p.np * sizeof(int) * batch);
    const size_t abytes = align256((size_t)kAuxFloats * sizeof(float) * batch);
    const int mtld = 2 * p.np + 256;  // row_lo + NT * RPT <= 2 np + 256 for every panel geometry
    const size_t mtbytes = align256((size_t)kMaxW * mtld * sizeof(float));
    char *c = (char *)base;
    size_t off = 0;
    if (o) {
        o->m0 = (float *)(c + off);
        o->mstride = mbytes / sizeof(float);
        o->tstride = tbytes / sizeof(float);
        o->pt_bstride = tbytes * batch / sizeof(float);
        o->mtstride = mtbytes / sizeof(float);
        o->mtld = mtld;
    }
    off += mbytes * batch;
    if (o) o->m1 = (float *)(c + off);
    off += mbytes * batch;
    for (int i = 0; i < 3; ++i) {
        if (o) o->pt[i] = (float *)(c + off);
        off += tbytes * batch;
    }
    for (int i = 0; i < 2; ++i) {
        if (o) o->gt[i] = (float *)(c + off);
        off += tbytes * batch;
    }
    for (int i = 0; i < 2; ++i) {
        if (o) o->mt[i] = (float *)(c + off);
        off += mtbytes * batch;
    }
    for (int i = 0; i < 2; ++i) {
        if (o) o->aux[i] = (float *)(c + off);
        off += abytes;
    }
    if (o) o->xch = (unsigned long long *)(c + off);
    off += align256((size_t)kXchGranules * sizeof(unsigned long long) * batch);
    const size_t gkbytes = align256((size_t)(p.bw < kMaxBW ? p.bw : kMaxBW) * p.np * sizeof(float));
    if (o) { o->gk = (float *)(c + off); o->gkstride = gkbytes / sizeof(float); }
    off += gkbytes * batch;
    for (int i = 0; i < 2; ++i) {
        if (o) o->ub[i] = (float *)(c + off);
        off += gkbytes * batch;
        if (o) o->xs[i] = (float *)(c + off);
        off += gkbytes * batch;
    }
    if (o) o->xst = (float *)(c + off);
    off += gkbytes * batch;
    const size_t mfbytes = align256((size_t)p.np * p.bw * sizeof(float));
    if (o) o->mfstride = mfbytes / sizeof(float);
    for (int i = 0; i < 2; ++i) {
        if (o) o->mf[i] = (float *)(c + off);
        off += mfbytes * batch;
    }
    int **maps[10] = {o ? &o->submap[0] : nullptr, o ? &o->submap[1] : nullptr, o ? &o->invsub[0] : nullptr,
                      o ? &o->invsub[1] : nullptr, o ? &o->rowsrc[0] : nullptr, o ? &o->rowsrc[1] : nullptr,
                      o ? &o->rowsrc[2] : nullptr, o ? &o->rowsrc[3] : nullptr,
                      o ? &o->orig : nullptr,      o ? &o->invp : nullptr};
    for (int i = 0; i < 10; ++i) {
        if (o) *maps[i] = (int *)(c + off);
        off += ibytes;
    }
    return off;
}
size_t blocked_workspace_bytes(const BlockedPlan &p, int batch) { return blocked_carve(p, batch, nullptr, nullptr); }

// Shared panels only: true when matrix b was given up (see SubpanelArgs::guard).  Wave-uniform.
__device__ __forceinline__ bool matrix_given_up(const int *guard, int b)
{
    return guard != nullptr && __builtin_amdgcn_readfirstlane(guard[b]) == MI32_RUNTIME_ERROR;
}

// ---- init: A -> diag(A, I) in the first working copy (makeAugmentedMatrix counterpart,
//      mat_inv_32.cpp:177-192) + the compact copies of the first two sub-panels' columns ------
__global__ __launch_bounds__(256) void blocked_init_kernel(const float *__restrict__ in, int n, int np, int ld,
                                                            size_t mstride, float *__restrict__ m0, PanelExport ex,
                                                            size_t tstride, int *__restrict__ orig,
                                                            int *__restrict__ status)
{
    const int b = blockIdx.z;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i0 = blockIdx.y * 16;
    const float *a = in + (size_t)b * n * n;
    float *m = m0 + (size_t)b * mstride;
    bool nonfinite = false;  // boundary rule: a NaN / inf anywhere in the input is an invalid matrix
    if (j < np) {
#pragma unroll 4
        for (int u = 0; u < 16; ++u) {
            const int i = i0 + u;
            if (i >= np) break;
            float v;
            if (i < n && j < n) v = a[(size_t)i * n + j];
            else v = (i == j) ? 1.0f : 0.0f;
            nonfinite = nonfinite || (v - v != 0.0f);
            m[(size_t)i * ld + j] = v;
            panel_export_store(ex, tstride, b, np, j, i, v);
        }
    }
    if (blockIdx.y == 0 && j < np) orig[(size_t)b * np + j] = j;
    // status[b] was zeroed (MI32_OK) by the host before this launch; every writer stores the same value
    if (nonfinite && status) status[b] = MI32_SINGULAR;
}

// Diagnostic builds (make stamps -> lib/libmat_inv_32_stamps.so, tools/panel_stamps.py) record s_memtime at the
// phase boundaries of every panel launch (wave 0 of workgroup 0); in the product build the macro expands to nothing.
#ifdef MI32_PANEL_STAMPS
__device__ unsigned long long *g_panel_stamps;  // [1024 launches][64 slots]
#define MI32_PSTAMP(TAG_, SLOT_)                                                                             \
    do {                                                                                                     \
        if (g_panel_stamps && threadIdx.x == 0 && blockIdx.x == 0)                                           \
            g_panel_stamps[(size_t)((TAG_) & 1023u) * 64 + (SLOT_)] = __builtin_amdgcn_s_memtime();          \
    } while (0)
#else
#define MI32_PSTAMP(TAG_, SLOT_) do { } while (0)
#endif

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

// ---- the pivot-row strip: BK pivot steps on the BK pivot rows alone, one column per quad ------------
// Every column outside a (sub-)panel sees that panel's BK pivot steps as
//     u_m = x[row of step m] / pivot_m                       fixRow,    mat_inv_32.cpp:138-150
//     x[i] = fmaf(-f_m[i], u_m, x[i])   for every other row  fixColumn, mat_inv_32.cpp:28-38
// for m = 0 .. BK-1 in order, f_m[i] = the entry row i had in the pivot column when step m ran (the panel keeps
// these multipliers).  u_m only depends on the BK pivot rows themselves: the strip runs the BK steps on them -- BK
// dependent IEEE divisions -- and leaves u_m (what every other row multiplies with) and the pivot rows' values
// after the last step.  The 4 lanes of a quad share one column: lane g holds the rows BK/4 * g ... of it, the row
// of step M is broadcast with one quad_perm DPP move.  s_lt[m * LT + row] = -f_m[row] (own step: -pivot).
// The multipliers of a step do not depend on the chain: they are read from LDS kStripAhead steps early into a
// rotating window of registers, and scheduling barriers keep hipcc from sinking the reads back down to their uses
// (left alone it puts two dependent LDS round trips, ~250 cycles, into each of the BK dependent steps).
//
// The division x / pivot (fixRow, mat_inv_32.cpp:149: IEEE, correctly rounded) is the other half of a step's
// latency: hipcc expands it into v_div_scale x2, v_rcp, 6 fma, v_div_fmas, v_div_fixup -- 11 dependent instructions
// of which only five depend on x once the operands need no scaling.  The strip splits it: the pivots' part
// (reciprocal and its Newton step: the very instructions of the expansion, on the unscaled pivot) is computed once,
// BK pivots in BK lanes, before the chain starts; the chain keeps q0 = x r1, e1 = fma(-d, q0, x), q1 = fma(e1, r1, q0),
// e2 = fma(-d, q1, x), q = fma(e2, r1, q1).  v_div_scale leaves both operands alone and v_div_fixup returns q as it is
// exactly when (ISA, V_DIV_SCALE_F32 / V_DIV_FIXUP_F32) neither is zero, denormal, infinite or NaN, the exponents are
// less than 96 apart, the numerator's biased exponent is above 23 and the denominator's below 253: the fast path is
// taken for 2^-47 <= |.| < 2^48 on both sides -- bit for bit the full expansion's result -- and for an exact zero
// numerator.  Whether every operand was in range is collected beside the chain (no branch per step); if one was not,
// the whole strip is run again from the saved rows with the expansion itself (strip_steps_full_division).
static constexpr int kStripAhead = 3;
__device__ __forceinline__ bool strip_div_in_range(float v)
{
    return __builtin_fabsf(v) >= 0x1p-47f && __builtin_fabsf(v) < 0x1p48f;
}
// The multipliers of step M for this lane's rows, read kStripAhead steps early (ALIGNED: one 16-byte LDS read)
template <int BK, int M, bool ALIGNED>
__device__ __forceinline__ void strip_fetch(float (&nfw)[kStripAhead + 1][BK / 4], const float *s_lt, int LT, int g)
{
    constexpr int CPT = BK / 4;
    if constexpr (M < BK) {
        if constexpr (ALIGNED && CPT == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(s_lt + M * LT + CPT * g);
            nfw[M % (kStripAhead + 1)][0] = v.x;
            nfw[M % (kStripAhead + 1)][1] = v.y;
            nfw[M % (kStripAhead + 1)][2] = v.z;
            nfw[M % (kStripAhead + 1)][3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < CPT; ++j) nfw[M % (kStripAhead + 1)][j] = s_lt[M * LT + CPT * g + j];
        }
    }
}
// One step on the fast division.  dv / rv: lane m (mod BK) holds -pivot_m and the refined reciprocal of pivot_m; the
// step's pair reaches every lane as two scalars (v_readlane: off the chain).  `ok` collects whether every numerator
// was in the fast path's range (or an exact zero: its five instructions return a zero of either sign, -0.0 == 0.0, and
// nothing downstream can tell them apart but the sign of another zero) -- no branch inside the chain.
template <int BK, int M, bool ALIGNED>
__device__ __forceinline__ void strip_step(float (&x)[BK / 4], float (&uu)[BK], float (&nfw)[kStripAhead + 1][BK / 4],
                                           float dv, float rv, bool &ok, const float *s_lt, int LT, int g)
{
    constexpr int CPT = BK / 4;
    constexpr int kQuad = (M / CPT) * 0x55;  // quad_perm:[q,q,q,q]
    strip_fetch<BK, M + kStripAhead, ALIGNED>(nfw, s_lt, LT, g);
    const float dneg = lane_bcast(dv, M);
    const float r1 = lane_bcast(rv, M);
    __builtin_amdgcn_sched_barrier(0);
    const float xm = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[M % CPT]), kQuad, 0xf, 0xf, false));
    float q = xm * r1;
    float e = __builtin_fmaf(dneg, q, xm);
    q = __builtin_fmaf(e, r1, q);
    e = __builtin_fmaf(dneg, q, xm);
    const float u = __builtin_fmaf(e, r1, q);
    ok = ok && (strip_div_in_range(xm) || xm == 0.0f);
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const float upd = __builtin_fmaf(nfw[M % (kStripAhead + 1)][j], u, x[j]);
        x[j] = (j == M % CPT && g == M / CPT) ? u : upd;
    }
    uu[M] = u;
    __builtin_amdgcn_sched_barrier(0);
}
// The same BK steps with the compiler's own IEEE division (v_div_scale / v_div_fmas / v_div_fixup): taken when a pivot
// or a numerator lies outside the fast path's range -- rare, and then for the whole strip.
template <int BK>
__device__ __forceinline__ void strip_steps_full_division(float (&x)[BK / 4], float (&uu)[BK], const float *s_lt, int LT, int g)
{
    constexpr int CPT = BK / 4;
#pragma unroll 1
    for (int m = 0; m < BK; ++m) {
        const int src = (threadIdx.x & 60) | (m / CPT);  // the quad's lane that holds row m
        float xm = 0.0f;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const float v = __shfl(x[j], src, 64);
            xm = (j == m % CPT) ? v : xm;
        }
        const float u = xm / -s_lt[m * LT + m];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const float upd = __builtin_fmaf(s_lt[m * LT + CPT * g + j], u, x[j]);
            x[j] = (j == m % CPT && g == m / CPT) ? u : upd;
        }
#pragma unroll
        for (int k = 0; k < BK; ++k) uu[k] = (k == m) ? u : uu[k];
    }
}
// u_m is stored at the end, by one lane of each quad: no LDS store between the steps.
template <int BK, bool ALIGNED, int... Ms>
__device__ __forceinline__ void strip_steps_t(float (&x)[BK / 4], const float *s_lt, int LT, int g, float *s_u, int LDU,
                                              std::integer_sequence<int, Ms...>)
{
    constexpr int CPT = BK / 4;
    float uu[BK];
    float nfw[kStripAhead + 1][BK / 4];
    float x0[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) x0[j] = x[j];
    // the pivots' part of the divisions, once per strip: lane m (mod BK) takes pivot m
    const int lane_m = (int)(threadIdx.x & 63) % BK;
    const float dv = s_lt[lane_m * LT + lane_m];  // -pivot
    const float r = __builtin_amdgcn_rcpf(-dv);
    const float e0 = __builtin_fmaf(dv, r, 1.0f);
    const float rv = __builtin_fmaf(e0, r, r);
    bool ok = strip_div_in_range(dv);
    strip_fetch<BK, 0, ALIGNED>(nfw, s_lt, LT, g);
    strip_fetch<BK, 1, ALIGNED>(nfw, s_lt, LT, g);
    strip_fetch<BK, 2, ALIGNED>(nfw, s_lt, LT, g);
    static_assert(kStripAhead == 3, "the three fetches above");
    (strip_step<BK, Ms, ALIGNED>(x, uu, nfw, dv, rv, ok, s_lt, LT, g), ...);
    if (!__all(ok)) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) x[j] = x0[j];
        strip_steps_full_division<BK>(x, uu, s_lt, LT, g);
    }
    if (g == 0) {
#pragma unroll
        for (int m = 0; m < BK; ++m) s_u[m * LDU] = uu[m];
    }
}
// Every caller keeps s_lt 16-byte aligned with LT a multiple of 4 (PanelShared::lt, UpdateTileShared::s_lt,
// OStripShared::s_lt, the block strip kernel's s_lt + s0): a step's multipliers are one 16-byte LDS read at W = 16.
template <int BK, int... Ms>
__device__ __forceinline__ void strip_steps(float (&x)[BK / 4], const float *s_lt, int LT, int g, float *s_u, int LDU,
                                            std::integer_sequence<int, Ms...> seq)
{
    strip_steps_t<BK, true>(x, s_lt, LT, g, s_u, LDU, seq);
}

// ---- the panel: W pivot steps on a register-resident slab ----------------------
// Each thread keeps RPT rows of the panel in registers for the whole kernel: row
// CONTENTS never move between threads.  What a row swap changes is only an integer
// label = the position (row index of the working matrix) that the content of a
// register row currently occupies:
//   pivotElements (mat_inv_32.cpp:154-173)  ==  exchange of two labels.
// submap[position] = where the data that now belongs at that position lies in the
// previous order tells the rank-k updates where every other column's data still lives.

template <int NW, int W>
struct __attribute__((aligned(16))) PanelShared {
    float cand[NW][W];          // per wave: its best candidate row as found (wave-private scratch)
    float prn[2][NW][W];        // per step parity, per wave: that row NORMALISED (candidate pivot row)
    unsigned long long key[W];  // one cross-wave arg-max word per step, zeroed at kernel start
    float prn_all[W][W];        // the normalised pivot row of every step, exported for the rows above the block
    float bprev[W][W];          // the previous sub-panel's W pivot rows, restricted to this sub-panel's columns
    float uprev[W][W];          // ... as that sub-panel's own steps saw them (u_m of the strip)
    float lt[W][W + 4];         // -multipliers of those W pivot rows, [step][row]; at the end: this sub-panel's own
    unsigned gx[2][kMaxPanelGroups][W + 2];  // multi-workgroup panels: every workgroup's winner of this step
    int lost;                   // multi-workgroup panels: a partner timed out (sticky; zeroed at kernel start)
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
// What a workgroup of a multi-workgroup panel knows about the others (MULTI instances only).
struct PanelGroup {
    int ngroups, grp;           // workgroups sharing this panel, and which one this is
    unsigned long long *xch;    // this matrix's exchange granules, [2][kMaxPanelGroups][32] x {payload, tag}
    unsigned tag_base;          // unique per launch within a call (<< 8 | step + 1 = the tag of a step)
    bool timed_out;
};

// V floats to sbase (wave-uniform) + voff bytes (per lane): global_store with a scalar base.  (Inline asm: hipcc does
// not insert the wait state between a store of more than 8 bytes and the overwrite of its data registers here.)
template <int V, typename T>
__device__ __forceinline__ void mt_store(float *sbase, unsigned voff, T v)
{
#ifdef MI32_TIMING_NO_MT_STORE  // timing-only builds (wrong results): what do the multiplier stores cost the panel?
    return;
#endif
    if constexpr (V == 1) asm volatile("global_store_dword %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase));
    else if constexpr (V == 2) asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(voff), "v"(v), "s"(sbase));
    else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase));
}

// Every step stores its multiplier column straight away: mtp = this lane's first slot of Mt row 0 (slab order), or,
// LBL (fused instances: the rows' labels at entry differ from their slab index), mt_base + moff[k] per row.
template <int NT, int RPT, int W, int R, bool MULTI, bool LBL>
__device__ __forceinline__ void panel_step(float (&a)[RPT][W], unsigned (&npl)[RPT],
                                           const int (&moff)[LBL ? RPT : 1], PanelShared<NT / 64, W> &sh,
                                           int wave_u, int c0, bool wave_active, bool &singular, PanelGroup &pg,
                                           float *mtp, int mtld)
{
    constexpr int par = R & 1;
    const int slot = c0 + R;
    // The lane id is recomputed in every step (two v_mbcnt, opaque to the optimiser): a `lane` carried through
    // the 16 unrolled steps is the first thing the 128-VGPR instances spill, and every path of the step reads
    // it -- a scratch reload in front of each compare on the critical path.
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));

    constexpr bool kSub = (R == W / 2);  // diagnostic builds: phase stamps inside one representative step
    if (kSub) MI32_PSTAMP(pg.tag_base, 32);
    // -- maxPivot over this lane's rows
    float col[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) col[k] = a[k][R];
    // the multiplier column of this step (mat_inv_32.cpp:30: what fixColumn reads before it overwrites the column);
    // the pivot row's own entry is the pivot
    // One store per step, fire and forget -- written so that NOTHING of it lives in vector registers across the
    // steps: the base is scalar (global_store ... saddr form), the 32-bit lane offset is recomputed from the lane id
    // (fused instances: one kept offset per row).  A pointer kept in VGPRs is the first thing the 128-VGPR instances
    // spill, and its reload's s_waitcnt vmcnt(0) then waits for the previous step's store to be acknowledged by the
    // memory system: +0.9 us per pivot step (measured: 28.5 -> 43 us per 16-step launch).
    if constexpr (LBL) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) mt_store<1>(mtp + (size_t)R * mtld, (unsigned)moff[k] * 4u, col[k]);
    } else {
        constexpr int V = RPT < 4 ? RPT : 4;
        typedef float mvecV __attribute__((ext_vector_type(V)));
        const unsigned voff = (unsigned)(wave_u * 64 + lane) * (4u * V);
#pragma unroll
        for (int g = 0; g < RPT / V; ++g) {
            mvecV v;
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] = col[g * V + j];
            if constexpr (V == 1) mt_store<1>(mtp + (size_t)R * mtld + g * (V * NT), voff, v[0]);
            else mt_store<V>(mtp + (size_t)R * mtld + g * (V * NT), voff, v);
        }
    }
    int own_lane = -1, own_k = 0;
    bool cand_bad = false;  // this wave's candidate has a zero / NaN / infinite pivot entry
    float qv = 0.0f;        // lanes 0..W-1: this wave's candidate row, normalised (kept for the export if it wins)
    if (wave_active) {
        unsigned mkey = 0u, mnp = 0u;
        int kb = 0;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const unsigned lm = (unsigned)((int)npl[k] >> 31);                    // all ones while live
            const unsigned key = __float_as_uint(col[k]) & lm & 0x7fffffffu;
            // (two 32-bit compares, not one 64-bit compare: the register pairs a v_cmp_gt_u64 needs cost the 128-VGPR
            // instances copies and spills in the middle of the steps)
            const bool better = key > mkey || (key == mkey && npl[k] > mnp);
            mkey = better ? key : mkey;
            mnp = better ? npl[k] : mnp;
            kb = better ? k : kb;
        }
        if (kSub) MI32_PSTAMP(pg.tag_base, 33);
        const unsigned wm = wave_max_u32(mkey);
        if (kSub) MI32_PSTAMP(pg.tag_base, 34);
        // lanes that hold the wave maximum and a real candidate: almost always exactly one
        unsigned long long hit = __ballot(mkey == wm && (int)mnp < 0);
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
            if (kSub) MI32_PSTAMP(pg.tag_base, 35);
            // fixRow, speculatively: lanes 0..W-1 divide one element each (IEEE); identity entry -> 1/piv.
            // One wave's LDS operations execute in order, so no s_barrier is needed between the holder
            // lane's store and these loads -- only the compiler must not reorder here.
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const float cpiv = sh.cand[wave_u][R];
            const float num = (lane < W) ? ((lane == R) ? 1.0f : sh.cand[wave_u][lane]) : 0.0f;
            if (kSub) MI32_PSTAMP(pg.tag_base, 36);
            qv = num / cpiv;
            cand_bad = (cpiv == 0.0f || cpiv - cpiv != 0.0f);
            if (kSub) MI32_PSTAMP(pg.tag_base, 37);
            if (lane < W) sh.prn[par][wave_u][lane] = qv;
            if (lane == 0)
                atomicMax(&sh.key[R], ((unsigned long long)wm << 32) |
                                          (unsigned long long)(((0xFFFFFu - wi) << 8) | (unsigned)wave_u));
        }
    }
    if (kSub) MI32_PSTAMP(pg.tag_base, 38);
    __syncthreads();
    if (kSub) MI32_PSTAMP(pg.tag_base, 39);
    // (Tried: reading every wave's candidate in the same LDS round as the arg-max word and picking the winner's W
    // entries out of their lanes with v_readlane into SGPRs -- no dependent second read, no spills in the 128-VGPR
    // instances.  16 v_readlane per wave and step cost more than the LDS round trip they replace once 2 or 4
    // waves share a SIMD: 2905 -> 3556 cycles per step at 4096 rows, 1876 -> 1800 with one wave per SIMD.)
    unsigned long long key = sh.key[R];
    unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(key & 0xFFFFFFFFull));
    if (kSub) MI32_PSTAMP(pg.tag_base, 40);
    float prn[W];  // prn[R] = 1/piv (the identity column's entry), prn[c] = normalised pivot row
    bool my_group_won = true;
    if constexpr (MULTI) {
        // -- the workgroups of this panel exchange their local winners: W normalised entries + the 64-bit key,
        //    as 8-byte {payload, tag} granules, each written by ONE agent-scope store and polled with agent-scope
        //    loads (a granule is its own flag: MI355X_MICROARCH.md, handoff-1to1).  The tag is unique per step
        //    and launch and the buffers alternate with the step parity: a workgroup can only be one step ahead.
        //    Every spin is bounded: on a time-out the matrix is flagged and the step goes on with what it has.
        const unsigned tag = (pg.tag_base << 8) | (unsigned)(R + 1);
        unsigned long long *xq = pg.xch + (size_t)par * (kMaxPanelGroups * 32);
        if (wave_u == 0 && lane < W + 2) {
            const int lwv = (int)(lo & 0xFFu);
            unsigned payload;
            if (lane < W) payload = __float_as_uint(sh.prn[par][lwv][lane]);
            else if (lane == W) payload = (lo & ~0xFFu) | ((unsigned)pg.grp << 4) | (unsigned)lwv;
            else payload = (unsigned)(key >> 32);
            __hip_atomic_store(&xq[pg.grp * 32 + lane], ((unsigned long long)tag << 32) | payload, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        if (wave_u < pg.ngroups) {  // wave g collects workgroup g's record (its own workgroup's too)
            const unsigned long long *src = xq + wave_u * 32;
            unsigned long long v = 0ull;
            // Every spin is bounded in TIME (s_memrealtime: 100 MHz).  A partner that has not shown up after
            // kPanelXchTimeoutTicks is given up for good: its record counts as "no candidate" (key 0) in this and
            // every later step -- all labels stay valid positions of the present rows -- the matrix is flagged
            // MI32_RUNTIME_ERROR, skipped by every later launch of the call and handed out NaN-filled.
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            for (;;) {
                if (lane < W + 2) v = __hip_atomic_load(&src[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ok = (lane >= W + 2) || ((unsigned)(v >> 32) == tag);
                if (__all(ok)) break;
                if (pg.timed_out || __builtin_amdgcn_s_memrealtime() - t_start > kPanelXchTimeoutTicks) {
                    v = 0ull;
                    if (lane == 0) sh.lost = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane < W + 2) sh.gx[par][wave_u][lane] = (unsigned)v;
        }
        __syncthreads();
        if (sh.lost != 0) pg.timed_out = true;  // workgroup-uniform from here on
        int gw = 0;
        key = ((unsigned long long)sh.gx[par][0][W + 1] << 32) | sh.gx[par][0][W];
#pragma unroll
        for (int g = 1; g < kMaxPanelGroups; ++g)
            if (g < pg.ngroups) {
                const unsigned long long kg = ((unsigned long long)sh.gx[par][g][W + 1] << 32) | sh.gx[par][g][W];
                if (kg > key) { key = kg; gw = g; }
            }
        gw = __builtin_amdgcn_readfirstlane(gw);
        lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(key & 0xFFFFFFFFull));
        my_group_won = (gw == pg.grp);
#pragma unroll
        for (int c = 0; c < W; c += 4) {
            const uint4 t = *reinterpret_cast<const uint4 *>(&sh.gx[par][gw][c]);
            prn[c] = __uint_as_float(t.x); prn[c + 1] = __uint_as_float(t.y);
            prn[c + 2] = __uint_as_float(t.z); prn[c + 3] = __uint_as_float(t.w);
        }
        if (wave_u == 0 && lane < W) sh.prn_all[R][lane] = __uint_as_float(sh.gx[par][gw][lane]);
    }
    int p = (int)(0xFFFFFu - (lo >> 8));
    if constexpr (MULTI) {
        // no record at all (only after a partner was lost and this workgroup has no candidate left): keep the
        // label a valid position -- nothing out of range may ever reach the row maps
        if (key == 0ull) p = slot;
    }
    const int wv = MULTI ? (int)(lo & 0xFu) : (int)(lo & 0xFFu);
    if (key == 0ull) singular = true;  // cannot happen (position `slot` is always a live candidate); never trust it
    if constexpr (!MULTI) {
#pragma unroll
        for (int c = 0; c < W; c += 4) {
            const float4 t = *reinterpret_cast<const float4 *>(&sh.prn[par][wv][c]);
            prn[c] = t.x; prn[c + 1] = t.y; prn[c + 2] = t.z; prn[c + 3] = t.w;
        }
    }
    if (kSub) { asm volatile("" ::"v"(prn[0]), "v"(prn[W - 1])); MI32_PSTAMP(pg.tag_base, 41); }
    // -- fixColumn on the slab, branch-free; the pivot column holds the implicit identity column, whose
    //    entry is 0 in every row but the pivot row.  The pivot row itself is overwritten right after.
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const float f = col[k];
#pragma unroll
        for (int c = 0; c < W; ++c)
            a[k][c] = (c == R) ? __builtin_fmaf(-f, prn[R], 0.0f) : __builtin_fmaf(-f, prn[c], a[k][c]);
    }
    if (kSub) { asm volatile("" ::"v"(a[0][0]), "v"(a[RPT - 1][W - 1])); MI32_PSTAMP(pg.tag_base, 42); }
    // -- pivotElements == exchange of two position labels: the row that held `slot` takes p ...
    if (p != slot) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) npl[k] = (npl[k] == ~(unsigned)slot) ? ~(unsigned)p : npl[k];
    }
    // ... and the winner's candidate row (its wave knows lane and row) becomes the pivot row: normalised
    // values, label `slot`, no longer a candidate
    if (my_group_won && wave_u == wv) {
        // the winner's own pivot entry decides "singular" (zero, NaN or infinite pivot); its normalised row is
        // still in lanes 0..W-1 -- no LDS read on the slowest wave's way to the next barrier
        if (cand_bad) singular = true;
        if (!MULTI && lane < W) sh.prn_all[R][lane] = qv;
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
    MI32_PSTAMP(pg.tag_base, 3 + R);
}

template <int NT, int RPT, int W, bool MULTI, bool LBL, int... Rs>
__device__ __forceinline__ void panel_steps(float (&a)[RPT][W], unsigned (&npl)[RPT],
                                            const int (&moff)[LBL ? RPT : 1], PanelShared<NT / 64, W> &sh,
                                            int wave_u, int c0, bool wave_active, bool &singular, PanelGroup &pg,
                                            float *mtp, int mtld, std::integer_sequence<int, Rs...>)
{
    (panel_step<NT, RPT, W, Rs, MULTI, LBL>(a, npl, moff, sh, wave_u, c0, wave_active, singular, pg, mtp, mtld), ...);
}

// Everything one fused sub-panel launch needs (passed by value).
struct SubpanelArgs {
    int np, n, ld, batch;
    size_t mstride, tstride;
    // ---- panel(s): workgroup b < batch of the grid (absent when panel_on == 0)
    int panel_on;
    int c0;        // first column of sub-panel s
    int has_prev;  // update(s-1) is still pending on this sub-panel's columns: apply it in the prologue
    int c0_prev;   // first column of sub-panel s-1
    int row_lo;    // the workgroup holds the rows [row_lo, np) of its input order
    int first_in_block;
    const float *pt_in;      // Pt_s: this sub-panel's columns, updates up to s-2 applied, order after s-2
    const float *mt_prev;    // Mt_{s-1}: the multipliers of sub-panel s-1, same order
    float *gt_out;           // Gt_s, order after s-1
    float *mt_out;           // Mt_s: the multipliers of this sub-panel's W steps (fused: order after s-1; else slab order = the same)
    size_t mtstride; int mtld;
    const int *submap_prev;  // submap of sub-panel s-1: position after s-1 -> index in order after s-2
    const int *invsub_prev;  // index in order after s-2 -> position after s-1 (the row's label at entry)
    int *submap_out;         // position after s -> index in order after s-1
    int *invsub_out;         // its inverse
    const int *rowsrc_in;    // position after s-1 -> row index at the start of the block
    int *rowsrc_out;         // the same after s (fused blocks: the other buffer -- update(s-1) still reads rowsrc_in)
    int *rowsrc_alt;         // fused blocks, first sub-panel: the second buffer, whose rows above the block are set too
    int *orig;
    float *aux_out;          // [kAuxFloats] per matrix: normalised pivot rows of s; U_{s-1} x columns of s
    int *status;
    const int *guard;        // non-null for plans with shared panels: status words; a matrix flagged
                             // MI32_RUNTIME_ERROR (a panel lost a partner: its row maps are not to be trusted)
                             // is skipped by every later launch and comes out as NaN
    int ngroups;             // workgroups per panel (> 1: MULTI instances, kPanelGroupRows rows each)
    unsigned long long *xch; // [batch][kXchGranules] exchange granules of the multi-workgroup panels
    unsigned tag_base;       // unique per panel launch within a call
    // ---- update(t), t = s-1: the other workgroups (absent when upd_on == 0)
    int upd_on;
    int u_c0;        // first column of sub-panel t
    int u_has_prev;  // sub-panel t itself had a pending update (t >= 1 within its block)
    int u_above_hi;  // positions below this were not in panel(t): their G_t is computed by the update tile
    int u_panel_hi;  // ... and positions from this on neither (np with pivoting; no-pivot variant: only the W pivot rows
                     // go through the "panel", gj_diag_panel_kernel)
    int C0, kb;      // the outer block
    const float *x;  // working copy in order after t-1
    float *y;        // working copy written in order after t
    const float *u_gt;     // Gt_t
    const float *u_mt;     // Mt_t
    const int *u_rowsrc;   // position after t -> row index at the start of the block
    float *u_mf;           // the block's negated multipliers by block-start row index, [np][mf_ld]
    size_t mfstride; int mf_ld;
    const int *u_submap;   // submap_t
    const float *u_pt_in;  // Pt_t (for the rows above the block)
    const float *u_aux;    // aux_t
    PanelExport u_exp;     // the columns of sub-panel t+2 -> its compact panel input
    int upd_wgs;           // workgroups of the launch that run update tiles
    // ---- strip(t) of the columns outside the block (absent when os_on == 0): uses the u_ fields of sub-panel t
    int os_on;
    int os_first, os_ntiles;  // the tiles' columns: os_ntiles x 64 from os_first on (os_first == 0: the block's own are skipped)
    const float *os_cur;   // the working copy the columns outside the block are still valid in
    float *os_ub, *os_xs;  // the block's u rows / its pivot rows after their own sub-panel, kb x np each
    size_t ubstride;
    int drop_groups;       // tests only: panel workgroups left out of a multi-workgroup panel launch
};

template <int NW, int W>
constexpr size_t panel_shared_bytes()
{
    return (sizeof(PanelShared<NW, W>) + 15) & ~(size_t)15;
}

// panel(s) of one matrix: the whole workgroup.  smem: panel_shared_bytes + 2 * RPT * NT ints.
// FUSED = false compiles the pending-update prologue (and the labels-at-entry indirection) out: the instances
// with 4 and more rows per lane have no registers to spare for code they never run.
// MULTI: the panel is shared by A.ngroups workgroups; this one (grp) holds the rows
// [row_lo + grp * NT * RPT, row_lo + (grp + 1) * NT * RPT) and takes part in the per-step exchange (panel_step).
template <int NT, int RPT, int W, bool FUSED, bool MULTI>
__device__ __forceinline__ void panel_body(const SubpanelArgs &A, int b, int grp, unsigned char *smem)
{
    static_assert(!(FUSED && MULTI), "multi-workgroup panels are never fused");
    if (matrix_given_up(A.guard, b)) return;
    const bool has_prev = FUSED && A.has_prev;
    constexpr int V = RPT < 4 ? RPT : 4;
    constexpr int NW = NT / 64;
    typedef float vecV __attribute__((ext_vector_type(V)));
    typedef int ivecV __attribute__((ext_vector_type(V)));
    PanelShared<NW, W> &sh = *reinterpret_cast<PanelShared<NW, W> *>(smem);
    int *s_park = reinterpret_cast<int *>(smem + panel_shared_bytes<NW, W>());  // [2][RPT][NT]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int np = A.np, c0 = A.c0;
    const int row_lo = A.row_lo + (MULTI ? grp * (NT * RPT) : 0);  // first row THIS workgroup holds
    const float *pt = A.pt_in + (size_t)b * A.tstride;
    const int *invsub_prev = A.invsub_prev + (size_t)b * np;
    float *mt = A.mt_out + (size_t)b * A.mtstride;
    const int mtld = A.mtld;
    if (tid < W) sh.key[tid] = 0ull;
    if (tid == 0) sh.lost = 0;
    MI32_PSTAMP(A.tag_base, 0);

    // -- the slab and every row's label at entry (its position after the previous sub-panel's swaps)
    float a[RPT][W];
    unsigned npl[RPT];
#pragma unroll
    for (int g = 0; g < RPT / V; ++g) {
        const int row = row_lo + panel_row<NT, RPT>(tid, g * V);  // first of V consecutive rows
        ivecV p0;
#pragma unroll
        for (int j = 0; j < V; ++j) p0[j] = row + j;
        if (has_prev && row < np) p0 = *reinterpret_cast<const ivecV *>(invsub_prev + row);
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
            // A candidate is a row of the matrix at or below the block.  A real column (slot < n) may only take
            // its pivot from the real rows: the identity padding holds exact zeros there, which can tie only
            // with an all-zero column, and then the lowest position -- a real row -- wins the tie.
            // Rows beyond the matrix (row >= np) are dead and are never written back.
            const bool live = (row + j < np) && (p0[j] >= c0);
            npl[g * V + j] = live ? ~(unsigned)p0[j] : (unsigned)p0[j];
        }
    }
    // the row maps this workgroup will permute: fetched now (by label), so their latency hides behind the
    // steps, and parked in thread-private LDS slots (the 1024-thread instances have no registers to spare)
    const int *rowsrc_in = A.rowsrc_in + (size_t)b * np;
    int *rowsrc = A.rowsrc_out + (size_t)b * np;
    int *orig = A.orig + (size_t)b * np;
    // All 2 * RPT loads are requested before the first is used (addresses clamped instead of guarded: behind a
    // condition hipcc waits for each load before it issues the next -- eight dependent round trips in front of the
    // first pivot step of a 4-rows-per-lane panel).
    {
        int pr[RPT], po[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int row = row_lo + panel_row<NT, RPT>(tid, k);
            const int p0 = (int)(npl[k] ^ (unsigned)((int)npl[k] >> 31));
            const int pi = row < np ? p0 : 0;
            pr[k] = rowsrc_in[pi];
            po[k] = orig[pi];
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int row = row_lo + panel_row<NT, RPT>(tid, k);
            const int p0 = (int)(npl[k] ^ (unsigned)((int)npl[k] >> 31));
            s_park[k * NT + tid] = (A.first_in_block || row >= np) ? p0 : pr[k];  // composite map so far
            s_park[(RPT + k) * NT + tid] = row < np ? po[k] : 0;
        }
    }
    // rows above the block keep their place: identity entries in the maps the update kernels read
    if (A.first_in_block && grp == 0) {
        for (int i = tid; i < row_lo; i += NT) rowsrc[i] = i;
        if (A.rowsrc_alt != nullptr) {
            int *alt = A.rowsrc_alt + (size_t)b * np;
            for (int i = tid; i < row_lo; i += NT) alt[i] = i;
        }
    }

#ifdef MI32_PANEL_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    MI32_PSTAMP(A.tag_base, 1);
    if (has_prev) {
        // -- update(s-1) on this sub-panel's columns, which nobody has applied yet: the W pivot steps of s-1 as
        //    every column outside that sub-panel sees them (strip_step above).  First the strip on the W pivot rows
        //    of s-1, which this workgroup holds (one wave: W columns x 4 lanes); then every other row takes
        //      a[row][c] = fmaf(-f_m[row], u_m[c], a[row][c]),  m ascending
        //    -- one fmaf per element and step from the old value: the reference's own order (mat_inv_32.cpp:28-38).
        constexpr int CPT = W / 4;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int rel = (int)npl[k] - A.c0_prev;  // dead rows carry their position itself
            if ((unsigned)rel < (unsigned)W && row_lo + panel_row<NT, RPT>(tid, k) < np) {
#pragma unroll
                for (int c = 0; c < W; ++c) sh.bprev[rel][c] = a[k][c];
            }
        }
        {   // the multipliers of those W rows in the W steps of s-1, negated, [step][row]: Mt_{s-1} is stored by the
            // rows' labels at the entry of panel(s-1), submap_{s-1} says which label the pivot row of each step had
            const float *mtp = A.mt_prev + (size_t)b * A.mtstride;
            const int *smp = A.submap_prev + (size_t)b * np;
            for (int i = tid; i < W * W; i += NT)
                sh.lt[i % W][i / W] = -mtp[(size_t)(i % W) * mtld + smp[A.c0_prev + i / W]];
        }
        __syncthreads();
        if (tid < 4 * W) {
            const int c = tid >> 2, g = tid & 3;
            float x[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) x[j] = sh.bprev[CPT * g + j][c];
            strip_steps<W>(x, &sh.lt[0][0], W + 4, g, &sh.uprev[0][c], W, std::make_integer_sequence<int, W>{});
#pragma unroll
            for (int j = 0; j < CPT; ++j) sh.bprev[CPT * g + j][c] = x[j];
        }
        __syncthreads();
        const float *mtp = A.mt_prev + (size_t)b * A.mtstride;
#ifndef MI32_PRO_REGS
#define MI32_PRO_REGS 16  // (32: hipcc hoists every LDS read of the round and spills them -- 892 B of scratch per lane in the 1024 x 2 instance)
#endif
        constexpr int KC = (MI32_PRO_REGS / RPT) < 1 ? 1 : ((MI32_PRO_REGS / RPT) > W ? W : (MI32_PRO_REGS / RPT));  // k's per round of loads
        // a ROLLED loop over the rounds: unrolled, hipcc hoists every round's loads to the top and the whole of
        // Mt_{s-1} (RPT * W registers) is live beside the slab
#pragma unroll 1
        for (int k0 = 0; k0 < W; k0 += KC) {
            vecV gv[KC][RPT / V];
#pragma unroll
            for (int kk = 0; kk < KC; ++kk)
#pragma unroll
                for (int g = 0; g < RPT / V; ++g) {
                    const int row = row_lo + panel_row<NT, RPT>(tid, g * V);
                    gv[kk][g] = (row < np) ? *reinterpret_cast<const vecV *>(mtp + (size_t)(k0 + kk) * mtld + row)
                                           : (vecV)(0.0f);
                }
#pragma unroll
            for (int kk = 0; kk < KC; ++kk)
#pragma unroll
                for (int c4 = 0; c4 < W; c4 += 4) {
                    const float4 bq = *reinterpret_cast<const float4 *>(&sh.uprev[k0 + kk][c4]);
#pragma unroll
                    for (int g = 0; g < RPT / V; ++g)
#pragma unroll
                        for (int j = 0; j < V; ++j) {
                            const float nf = -gv[kk][g][j];
                            a[g * V + j][c4 + 0] = __builtin_fmaf(nf, bq.x, a[g * V + j][c4 + 0]);
                            a[g * V + j][c4 + 1] = __builtin_fmaf(nf, bq.y, a[g * V + j][c4 + 1]);
                            a[g * V + j][c4 + 2] = __builtin_fmaf(nf, bq.z, a[g * V + j][c4 + 2]);
                            a[g * V + j][c4 + 3] = __builtin_fmaf(nf, bq.w, a[g * V + j][c4 + 3]);
                        }
                }
        }
        // the pivot rows of s-1 themselves: what the strip left
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int rel = (int)npl[k] - A.c0_prev;
            if ((unsigned)rel < (unsigned)W && row_lo + panel_row<NT, RPT>(tid, k) < np) {
#pragma unroll
                for (int c = 0; c < W; ++c) a[k][c] = sh.bprev[rel][c];
            }
        }
    }
    bool singular = false;
    __syncthreads();  // sh.key[] zeroed before any wave's first atomicMax; all map reads issued
    PanelGroup pg = {A.ngroups, grp, A.xch + (size_t)b * kXchGranules, A.tag_base, false};
    MI32_PSTAMP(A.tag_base, 2);
    // Mt_s: by the rows' labels at entry (order after s-1, what update(s) and the next fused panel index it by).
    // Unfused panels hold their rows in that very order: this lane's first slot in row 0 of Mt (rows beyond np land
    // in the padding of the mtld-wide rows).  Fused panels hold them in the order after s-2: one offset per row.
    int moff[FUSED ? RPT : 1];
    if constexpr (FUSED) {
#pragma unroll
        for (int g = 0; g < RPT / V; ++g) {
            const int row = row_lo + panel_row<NT, RPT>(tid, g * V);
#pragma unroll
            for (int j = 0; j < V; ++j) moff[g * V + j] = row + j;  // rows beyond np: a slot in the row's padding
            if (has_prev && row < np) {
                const ivecV p0 = *reinterpret_cast<const ivecV *>(invsub_prev + row);
#pragma unroll
                for (int j = 0; j < V; ++j) moff[g * V + j] = p0[j];
            }
        }
    }
    float *mt_lane = FUSED ? mt : mt + row_lo;  // wave-uniform; the lane's part is added by the store
    panel_steps<NT, RPT, W, MULTI, FUSED>(a, npl, moff, sh, wave_u, c0, true, singular, pg, mt_lane, mtld,
                                          std::make_integer_sequence<int, W>{});
    // The thread index, recomputed behind an opaque instruction: everything the epilogue addresses hangs on it, so
    // hipcc cannot compute those addresses in front of the steps and carry them through (it spilled them).
    int tid_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(tid_e));
    tid_e += wave_u * 64;
    int pos[RPT];  // final position of every register row
#pragma unroll
    for (int k = 0; k < RPT; ++k) pos[k] = (int)(npl[k] ^ (unsigned)((int)npl[k] >> 31));

    // -- for the rows above the block (update(s) computes their G_s): the W normalised pivot rows, and the
    //    pivot rows of s-1 restricted to this sub-panel's columns
    __syncthreads();
    MI32_PSTAMP(A.tag_base, 48);
    float *aux = A.aux_out + (size_t)b * kAuxFloats;
    if (grp == 0)
        for (int i = tid_e; i < W * W; i += NT) {
            aux[i] = sh.prn_all[i / W][i % W];
            if (has_prev) aux[kMaxW * kMaxW + i] = sh.uprev[i / W][i % W];
        }
    // -- G_s by label at entry (order after s-1: what update(s) reads the working copy in); the row maps
    float *gt = A.gt_out + (size_t)b * A.tstride;
    int *submap = A.submap_out + (size_t)b * np;
    int *invsub = A.invsub_out + (size_t)b * np;
    // positions retired since this map buffer was last written: identity (any earlier position already is)
    if (grp == 0 && tid_e < 4 * kMaxW && row_lo - 4 * kMaxW + tid_e >= 0)
        submap[row_lo - 4 * kMaxW + tid_e] = row_lo - 4 * kMaxW + tid_e;
#pragma unroll
    for (int g = 0; g < RPT / V; ++g) {
        const int row = row_lo + panel_row<NT, RPT>(tid_e, g * V);
        if (row < np) {
            ivecV p0;  // the labels at entry, again (no registers were kept for them)
#pragma unroll
            for (int j = 0; j < V; ++j) p0[j] = row + j;
            if (has_prev) p0 = *reinterpret_cast<const ivecV *>(invsub_prev + row);
            bool contiguous = (p0[0] % V) == 0;
#pragma unroll
            for (int j = 1; j < V; ++j) contiguous = contiguous && (p0[j] == p0[0] + j);
            if (contiguous) {
#pragma unroll
                for (int c = 0; c < W; ++c) {
                    vecV v;
#pragma unroll
                    for (int j = 0; j < V; ++j) v[j] = a[g * V + j][c];
                    *reinterpret_cast<vecV *>(gt + (size_t)c * np + p0[0]) = v;
                }
            } else {
#pragma unroll
                for (int c = 0; c < W; ++c)
#pragma unroll
                    for (int j = 0; j < V; ++j) gt[(size_t)c * np + p0[j]] = a[g * V + j][c];
            }
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const int k = g * V + j;
                submap[pos[k]] = p0[j];  // position pos[k] now holds what lies at index p0[j] of the order after s-1
                invsub[p0[j]] = pos[k];
                rowsrc[pos[k]] = s_park[k * NT + tid_e];
                orig[pos[k]] = s_park[(RPT + k) * NT + tid_e];
            }
        }
    }
    // only the wave that won a step has looked at that step's pivot: any wave may raise the flag
#ifdef MI32_PANEL_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MI32_PSTAMP(A.tag_base, 49);
    if (g_panel_stamps && threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long *q = g_panel_stamps + (size_t)(A.tag_base & 1023u) * 64;
        q[50] = __builtin_amdgcn_s_memrealtime();
        q[51] = ((unsigned long long)NT << 32) | ((unsigned)RPT << 16) | ((unsigned)(FUSED ? 1 : 0) << 8) | (unsigned)W;
        q[52] = (unsigned long long)(np - row_lo);
    }
#endif
    // (atomicMax: a later "singular" must not hide "a partner workgroup never showed up")
    if (singular && lane == 0 && A.status) atomicMax(&A.status[b], (int)MI32_SINGULAR);
    if (pg.timed_out && lane == 0 && A.status) atomicMax(&A.status[b], (int)MI32_RUNTIME_ERROR);
}

// One pivot step of a row that is not a candidate, for the in-block update tiles: the row's BK panel entries
// are spread over the 4 threads of a quad (BK/4 consecutive columns each); its current entry in column R
// lives in thread R / (BK/4) and is broadcast with one quad_perm DPP move.  That entry is the row's multiplier of
// the step (fm: kept by the thread that owns column R).
template <int BK, int R>
__device__ __forceinline__ void above_rows_step(float (&v)[BK / 4], float (&fm)[BK / 4], const float *s_prn, int q4)
{
    constexpr int CPT = BK / 4;
    constexpr int kQuad = (R / CPT) * 0x55;  // quad_perm:[q,q,q,q]
    const float f = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[R % CPT]), kQuad, 0xf, 0xf, false));
    fm[R % CPT] = (q4 == R / CPT) ? f : fm[R % CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = q4 * CPT + j;
        const float base = (c == R) ? 0.0f : v[j];
        v[j] = __builtin_fmaf(-f, s_prn[R * BK + c], base);
    }
}
template <int BK, int... Rs>
__device__ __forceinline__ void above_rows_steps(float (&v)[BK / 4], float (&fm)[BK / 4], const float *s_prn, int q4,
                                                 std::integer_sequence<int, Rs...>)
{
    (above_rows_step<BK, Rs>(v, fm, s_prn, q4), ...);
}

typedef float float16v __attribute__((ext_vector_type(16)));

// ---- update(t): the in-block rank-W update on the fp32 matrix cores ---------------
// For the columns j of the block that are not sub-panel t's own, 64 x 64 tiles, 256 threads = 4 waves in a 2x2
// arrangement per tile (NG tiles per workgroup), one 32x32 MFMA tile per wave:
//   strip   : the tile's 64 columns of the W pivot rows of t run the W steps (strip_step): u_m[j] and the pivot
//             rows' new values;
//   update  : y[i][j] = x[map[i]][j] - sum_m f_m[i] * u_m[j] for every other row, ONE accumulation chain per output
//             element starting from the old value, m ascending (v_mfma_f32_32x32x2_f32 with the old value as its
//             C operand is that fmaf chain): exactly the operations the reference's step loop applies to the
//             element, in its order (mat_inv_32.cpp:28-38,317-362) -- bit for bit oracle/gj_oracle.c's
//             gjo_matrix_inv_32_inplace.
// The column-tile-0 workgroups also materialise sub-panel t's own columns G_t into y[i][c0 + k] (row-major) and
// the rows' negated multipliers into mf[block-start row][c0 - C0 + k], where the rank-bw update finds them.
template <int BK>
struct __attribute__((aligned(16))) UpdateTileShared {
    static constexpr int LDA = 64 + ((32 / BK) > 0 ? (32 / BK) : 1);
    static constexpr int LDB = 64 + 4;
    static constexpr int LT = BK + 4;
    float s_b[BK * LDB];     // pivot rows (through the row map) x 64 columns; after the strip: u_m
    float s_xs[BK * LDB];    // the pivot rows after the W steps
    float s_lt[BK * LT];     // -multipliers of the W pivot rows, [step][row]
    float s_prn[BK * BK];    // sub-panel t's normalised pivot rows
    float s_bprev[BK * BK];  // u_m of sub-panel t-1 restricted to sub-panel t's columns
    float s_a[BK * LDA];     // -multipliers of the tile's rows, [k][row]
    int s_map[64];
    int s_pmap[BK];          // where the W pivot rows lie in the order before the sub-panel's swaps
    int s_rs[64];            // the tile's rows' indices at the start of the block
};

template <int BK, int NG>
__device__ __forceinline__ void inblock_update_body(const SubpanelArgs &A, int u, unsigned char *smem)
{
    typedef UpdateTileShared<BK> TS;
    constexpr int LDA = TS::LDA, LDB = TS::LDB, LT = TS::LT;
    constexpr int CPT = BK / 4;
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;
    TS &T = reinterpret_cast<TS *>(smem)[grp];
    const int np = A.np, ld = A.ld, c0 = A.u_c0;
    const int tiles_x = A.kb / 64;
    const int wgs_per_matrix = tiles_x * (np / 64) / NG;
    const int b = u / wgs_per_matrix;
    if (matrix_given_up(A.guard, b)) return;
    const int id = (u % wgs_per_matrix) * NG + grp;
    const int tx = id % tiles_x, ty = id / tiles_x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int row0 = ty * 64;
    const int col0 = A.C0 + tx * 64;
    const float *src = A.x + (size_t)b * A.mstride;
    float *dst = A.y + (size_t)b * A.mstride;
    const float *g = A.u_gt + (size_t)b * A.tstride;
    const float *mt = A.u_mt + (size_t)b * A.mtstride;
    const int mtld = A.mtld;
    const int *map = A.u_submap + (size_t)b * np;
    const bool some_above = row0 < A.u_above_hi || row0 + 64 > A.u_panel_hi;  // some of this tile's rows were not in panel(t)

    // Two dependent rounds of global loads in all: the maps first, then everything they index (old values,
    // multipliers, pivot rows) -- requested into registers back to back, before the first of them is needed.
    if (tid < 64) T.s_map[tid] = map[row0 + tid];
    else if (tid < 64 + BK) T.s_pmap[tid - 64] = map[c0 + tid - 64];
    else if (tid >= 128 && tid < 192) T.s_rs[tid - 128] = (A.u_rowsrc + (size_t)b * np)[row0 + tid - 128];
    if (some_above) {
        const float *aux = A.u_aux + (size_t)b * kAuxFloats;
        for (int i = tid; i < BK * BK; i += 256) {
            T.s_prn[i] = aux[i];
            if (A.u_has_prev) T.s_bprev[i] = aux[kMaxW * kMaxW + i];
        }
    }
    __syncthreads();
    // 32-bit byte offsets from the matrix's (scalar) base: one v_mad_u32_u24 per access instead of a 64-bit
    // multiply-add pair (np <= 16384: the last byte of a matrix lies below 2^31)
    const unsigned ld4 = (unsigned)ld * 4u;
    const char *srcb = reinterpret_cast<const char *>(src);
    // (1) the W pivot rows' own multipliers: Mt_t[step][index of the row of step kk in order after t-1]
    constexpr int NLT = (BK * BK + 255) / 256;
    float lval[NLT];
#pragma unroll
    for (int q = 0; q < NLT; ++q) {
        const int i = tid + q * 256;
        lval[q] = (i < BK * BK) ? mt[(size_t)(i % BK) * mtld + T.s_pmap[i / BK]] : 0.0f;
    }
    // (2) the W pivot rows (through the row map) x 64 columns
    constexpr int NBQ = (BK * 16 + 255) / 256;
    float4 bq[NBQ];
#pragma unroll
    for (int q = 0; q < NBQ; ++q) {
        const int idx = tid + q * 256;
        if (idx < BK * 16)
            bq[q] = *reinterpret_cast<const float4 *>(
                srcb + ((unsigned)T.s_pmap[idx / 16] * ld4 + (unsigned)(col0 + (idx % 16) * 4) * 4u));
    }
    // (3) the accumulators start from the (row-mapped) old values
    float16v acc;
    const int lcol = lane & 31;
    const int lhalf = lane >> 5;
    {
        const unsigned col4 = (unsigned)(col0 + wc * 32 + lcol) * 4u;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int lr = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhalf;
            acc[reg] = *reinterpret_cast<const float *>(srcb + ((unsigned)T.s_map[lr] * ld4 + col4));
        }
    }
    {
        // (4) stage A = the tile's rows' multipliers, negated, [k][row]; 4 threads per row, BK/4 columns each.
        //  * rows that were in panel(t): its compact output mt[k][map[row]] (and gt[k][map[row]] = the row's new
        //    entries in sub-panel t's own columns);
        //  * rows above (never candidates, never moved): the row's W entries of the panel input Pt_t, brought up
        //    to date with update(t-1) where that was still pending (the chain of the panel prologue: old value -
        //    sum_m f_m[row] * u_m[c], f_m as materialised in mf), then taken through the W pivot steps with the
        //    exported normalised pivot rows -- fixColumn (mat_inv_32.cpp:28-38) on one row, the very fmaf
        //    sequence the panel applies to a dead row; the entry the row holds in the pivot column when a step
        //    runs is its multiplier.
        const int rr = tid >> 2, q4 = tid & 3;
        const int grow = row0 + rr;
        float v[CPT], fm[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) { v[j] = 0.0f; fm[j] = 0.0f; }
        float *mfrow = A.u_mf + (size_t)b * A.mfstride + (size_t)T.s_rs[rr] * A.mf_ld + (c0 - A.C0);
        if (grow >= A.u_above_hi && grow < A.u_panel_hi) {
#pragma unroll
            for (int j = 0; j < CPT; ++j) fm[j] = mt[(size_t)(q4 * CPT + j) * mtld + T.s_map[rr]];
            if (tx == 0) {
#pragma unroll
                for (int j = 0; j < CPT; ++j) v[j] = g[(size_t)(q4 * CPT + j) * np + T.s_map[rr]];
            }
        } else {
            const float *pt_in = A.u_pt_in + (size_t)b * A.tstride;
#pragma unroll
            for (int j = 0; j < CPT; ++j) v[j] = pt_in[(size_t)(q4 * CPT + j) * np + grow];
            if (A.u_has_prev) {
                const float *mp = mfrow - BK;  // -f_m of sub-panel t-1 for this row: same width, same block
#pragma unroll
                for (int k4 = 0; k4 < BK; k4 += 4) {
                    const float4 gq = *reinterpret_cast<const float4 *>(mp + k4);
                    const float gk4[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int j = 0; j < CPT; ++j)
                            v[j] = __builtin_fmaf(gk4[kk], T.s_bprev[(k4 + kk) * BK + q4 * CPT + j], v[j]);
                }
            }
            above_rows_steps<BK>(v, fm, T.s_prn, q4, std::make_integer_sequence<int, BK>{});
        }
        // everything requested; now into LDS
#pragma unroll
        for (int q = 0; q < NLT; ++q) {
            const int i = tid + q * 256;
            if (i < BK * BK) T.s_lt[(i % BK) * LT + i / BK] = -lval[q];
        }
#pragma unroll
        for (int q = 0; q < NBQ; ++q) {
            const int idx = tid + q * 256;
            if (idx < BK * 16) *reinterpret_cast<float4 *>(&T.s_b[(idx / 16) * LDB + (idx % 16) * 4]) = bq[q];
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) T.s_a[(q4 * CPT + j) * LDA + rr] = -fm[j];
        if (tx == 0) {  // materialise: G_t into the row-major working copy, -f into the block's multiplier matrix
#pragma unroll
            for (int j = 0; j < CPT; ++j) {
                dst[(size_t)grow * ld + c0 + q4 * CPT + j] = v[j];
                mfrow[q4 * CPT + j] = -fm[j];
            }
        }
    }
    __syncthreads();
    {   // the strip: column tid >> 2 of the W pivot rows, rows CPT * (tid & 3) ... in this lane
        const int c = tid >> 2, q4 = tid & 3;
        float x[CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) x[j] = T.s_b[(CPT * q4 + j) * LDB + c];
        strip_steps<BK>(x, T.s_lt, LT, q4, &T.s_b[c], LDB, std::make_integer_sequence<int, BK>{});
#pragma unroll
        for (int j = 0; j < CPT; ++j) T.s_xs[(CPT * q4 + j) * LDB + c] = x[j];
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
        const float af = T.s_a[(kk + lhalf) * LDA + wr * 32 + lcol];
        const float bf = T.s_b[(kk + lhalf) * LDB + wc * 32 + lcol];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
    }
    {
        const int col = col0 + wc * 32 + lcol;
        if (!(col >= c0 && col < c0 + BK)) {  // sub-panel t's own columns hold G_t, not an update result
            // the W pivot rows of t are rows c0 .. c0+W-1 of the new order: they take what the strip left
            if (row0 + wr * 32 < c0 + BK && row0 + wr * 32 + 32 > c0) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int rel = row0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhalf - c0;
                    if ((unsigned)rel < (unsigned)BK) acc[reg] = T.s_xs[rel * LDB + wc * 32 + lcol];
                }
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int grow = row0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhalf;
                *reinterpret_cast<float *>(reinterpret_cast<char *>(dst) + ((unsigned)grow * ld4 + (unsigned)col * 4u)) = acc[reg];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)  // registers 4q .. 4q+3 are 4 consecutive rows: one 16-byte store
                panel_export_store4(A.u_exp, A.tstride, b, np, col, row0 + wr * 32 + 8 * q + 4 * lhalf, acc[4 * q],
                                    acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
    }
}

// ---- strip(t): what the columns OUTSIDE the block see of sub-panel t's W pivot steps ---------------
// One 256-thread group per 64-column tile outside the block.  The W pivot rows of t have not been touched by the
// block's earlier sub-panels in these columns (their update is delayed to the end of the block), so their values at
// the start of the block first take the block's earlier steps,
//     x[kk][j] = fmaf(-f_m[row kk], u_m[j], x[kk][j]),  m = 0 .. c0 - C0 - 1 ascending        (mat_inv_32.cpp:28-38)
// (u_m: left in ub by the strips of the earlier sub-panels; -f_m: the block's multiplier matrix mf), then run their
// own W steps (strip_step: W dependent IEEE divisions).  Out: ub[c0 - C0 + m][j] = u_m[j], the pivot row of step m
// as fixColumn sees it = the B operand of the block's rank-bw update, and xs[c0 - C0 + kk][j] = pivot row kk after
// the sub-panel's last step = where that row's accumulation starts in the rank-bw update (which applies the later
// sub-panels' steps to it and nothing else: gj_mult_transpose_kernel masks the rest).
// The tiles ride in the launch of the NEXT panel (or in the block's last in-block update): off the chain of pivot steps.
template <int BK>
struct __attribute__((aligned(16))) OStripShared {
    static constexpr int MC = 32;       // earlier steps per round of loads
    static constexpr int LDU = 64 + 4;
    static constexpr int LT = BK + 4;
    float s_ub[MC * LDU];   // u_m of a round x 64 columns
    float s_mf[MC * LT];    // -f_m of the W pivot rows in a round, [m][row]
    float s_x[BK * LDU];    // the W pivot rows x 64 columns at the start of the block; after the strip: u_m
    float s_xs[BK * LDU];   // the W pivot rows after the sub-panel's last step
    float s_lt[BK * LT];    // -multipliers of the W pivot rows in the W steps of t, [step][row]
    int s_q[BK], s_idx[BK]; // their row index at the start of the block / in the order before t's swaps
};

template <int BK>
__device__ __forceinline__ void ostrip_body(const SubpanelArgs &A, int tile, unsigned char *smem_group, int tid)
{
    typedef OStripShared<BK> S;
    constexpr int MC = S::MC, LDU = S::LDU, LT = S::LT, CPT = BK / 4;
    S &T = *reinterpret_cast<S *>(smem_group);
    const int np = A.np, ld = A.ld, c0 = A.u_c0, C0 = A.C0, kb = A.kb;
    const int tiles = A.os_ntiles;
    // The 256-thread groups of a wider workgroup run different tiles and share the workgroup's barriers: no group
    // leaves early.  A group past the last tile repeats the last one without storing; a given-up matrix (its row
    // maps still hold valid positions) is computed and not stored.
    bool store_ok = tile < tiles * A.batch;
    if (!store_ok) tile = tiles * A.batch - 1;
    const int b = tile / tiles;
    store_ok = store_ok && !matrix_given_up(A.guard, b);
    int col0 = A.os_first + (tile % tiles) * 64;
    if (A.os_first == 0 && col0 >= C0) col0 += kb;  // all columns but the block's own
    const float *cur = A.os_cur + (size_t)b * A.mstride;
    const float *mt = A.u_mt + (size_t)b * A.mtstride;
    const float *mf = A.u_mf + (size_t)b * A.mfstride;
    float *ub = A.os_ub + (size_t)b * A.ubstride;
    float *xs = A.os_xs + (size_t)b * A.ubstride;
    const int K = c0 - C0;  // the block's steps before this sub-panel
    // (every group executes the same number of barriers: K is the same for all of them)
    if (tid < BK) {
        T.s_idx[tid] = (A.u_submap + (size_t)b * np)[c0 + tid];
        T.s_q[tid] = (A.u_rowsrc + (size_t)b * np)[c0 + tid];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < (BK * 16 + 255) / 256; ++q) {
        const int idx = tid + q * 256;
        if (idx < BK * 16)
            *reinterpret_cast<float4 *>(&T.s_x[(idx / 16) * LDU + (idx % 16) * 4]) =
                *reinterpret_cast<const float4 *>(cur + (size_t)T.s_q[idx / 16] * ld + col0 + (idx % 16) * 4);
    }
    for (int i = tid; i < BK * BK; i += 256)
        T.s_lt[(i % BK) * LT + i / BK] = -mt[(size_t)(i % BK) * A.mtld + T.s_idx[i / BK]];
    __syncthreads();
    const int c = tid >> 2, g = tid & 3;
    float x[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) x[j] = T.s_x[(CPT * g + j) * LDU + c];
    for (int m0 = 0; m0 < K; m0 += MC) {
        const int mc = (K - m0 < MC) ? (K - m0) : MC;  // a multiple of BK
#pragma unroll
        for (int q = 0; q < MC * 16 / 256; ++q) {
            const int idx = tid + q * 256;
            if (idx < mc * 16)
                *reinterpret_cast<float4 *>(&T.s_ub[(idx / 16) * LDU + (idx % 16) * 4]) =
                    *reinterpret_cast<const float4 *>(ub + (size_t)(m0 + idx / 16) * np + col0 + (idx % 16) * 4);
        }
        for (int idx = tid; idx < BK * (mc / 4); idx += 256) {
            const int kk = idx / (mc / 4), m4 = (idx % (mc / 4)) * 4;
            const float4 v = *reinterpret_cast<const float4 *>(mf + (size_t)T.s_q[kk] * A.mf_ld + m0 + m4);
            T.s_mf[(m4 + 0) * LT + kk] = v.x;
            T.s_mf[(m4 + 1) * LT + kk] = v.y;
            T.s_mf[(m4 + 2) * LT + kk] = v.z;
            T.s_mf[(m4 + 3) * LT + kk] = v.w;
        }
        __syncthreads();
        for (int mm = 0; mm < mc; mm += BK) {
#pragma unroll
            for (int i = 0; i < BK; ++i) {
                const float u = T.s_ub[(mm + i) * LDU + c];
#pragma unroll
                for (int j = 0; j < CPT; ++j) x[j] = __builtin_fmaf(T.s_mf[(mm + i) * LT + CPT * g + j], u, x[j]);
            }
        }
        __syncthreads();
    }
    strip_steps<BK>(x, T.s_lt, LT, g, &T.s_x[c], LDU, std::make_integer_sequence<int, BK>{});
#pragma unroll
    for (int j = 0; j < CPT; ++j) T.s_xs[(CPT * g + j) * LDU + c] = x[j];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < (BK * 16 + 255) / 256; ++q) {
        const int idx = tid + q * 256;
        if (idx < BK * 16 && store_ok) {
            const int kk = idx / 16, c4 = (idx % 16) * 4;
            *reinterpret_cast<float4 *>(ub + (size_t)(K + kk) * np + col0 + c4) =
                *reinterpret_cast<const float4 *>(&T.s_x[kk * LDU + c4]);
            *reinterpret_cast<float4 *>(xs + (size_t)(K + kk) * np + col0 + c4) =
                *reinterpret_cast<const float4 *>(&T.s_xs[kk * LDU + c4]);
        }
    }
}

// ---- the no-pivot variant's "panel" (matrix_inversion_no_pivots.cpp:10: findCrr / fixRow / fixColumn, no search,
//      no swap): the W x W diagonal block alone ------------------------------------------------------------------
// Without a pivot search the W pivot rows of a sub-panel are known in advance -- rows c0 .. c0+W-1 -- and what every
// OTHER row does in the W steps depends on those rows only: it is the update tiles that take each of them through the
// steps (above_rows_step, with the normalised pivot rows this kernel exports), thousands of rows in parallel on the
// whole chip instead of one workgroup.  This kernel runs the W steps on the W x W block of the pivot rows themselves
// (one thread per entry, two LDS hand-overs per step) and leaves what the panel kernel leaves for its rows: their new
// entries (gt), their multipliers (mt; own step: the pivot), the normalised pivot rows (aux) and the status.
// Workgroups past the matrices are strip(t) tiles, as in the other panel launches.
template <int W>
__global__ __launch_bounds__(256) void gj_diag_panel_kernel(SubpanelArgs A)
{
    constexpr size_t kBytes = sizeof(OStripShared<W>) > (3 * W * W + 2 * W) * sizeof(float) ? sizeof(OStripShared<W>)
                                                                                          : (3 * W * W + 2 * W) * sizeof(float);
    __shared__ __attribute__((aligned(16))) unsigned char dp_smem[kBytes];
    if ((int)blockIdx.x >= A.batch) {
        ostrip_body<W>(A, (int)blockIdx.x - A.batch, dp_smem, threadIdx.x);
        return;
    }
    float *s_d = reinterpret_cast<float *>(dp_smem);  // [W][W] the block
    float *s_prn = s_d + W * W;                       // [W][W] normalised pivot rows
    float *s_mt = s_prn + W * W;                      // [W][W] multipliers [step][row]
    const int b = blockIdx.x, tid = threadIdx.x, np = A.np, c0 = A.c0;
    if (matrix_given_up(A.guard, b)) return;
    const float *pt = A.pt_in + (size_t)b * A.tstride;
    for (int i = tid; i < W * W; i += 256) s_d[i] = pt[(size_t)(i % W) * np + c0 + i / W];  // s_d[row][col]
    __syncthreads();
    bool singular = false;
    for (int m = 0; m < W; ++m) {
        const float piv = s_d[m * W + m];
        if (piv == 0.0f || piv - piv != 0.0f) singular = true;
        // fixRow (IEEE division); the identity column's entry 1 becomes 1/piv
        for (int c = tid; c < W; c += 256) s_prn[m * W + c] = (c == m ? 1.0f : s_d[m * W + c]) / piv;
        __syncthreads();
        // fixColumn on the other W-1 rows of the block; the pivot column holds the implicit identity column (0)
        constexpr int EPT = (W * W + 255) / 256;
        float vv[EPT];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int i = tid + q * 256;
            vv[q] = 0.0f;
            if (i < W * W) {
                const int k = i / W, c = i % W;
                const float f = s_d[k * W + m];
                if (k == m) vv[q] = s_prn[m * W + c];
                else vv[q] = __builtin_fmaf(-f, s_prn[m * W + c], (c == m) ? 0.0f : s_d[i]);
                if (c == 0) s_mt[m * W + k] = f;  // own step: the pivot itself
            }
        }
        __syncthreads();  // every thread has read column m of its rows before anyone overwrites it
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int i = tid + q * 256;
            if (i < W * W) s_d[i] = vv[q];
        }
        __syncthreads();
    }
    float *gt = A.gt_out + (size_t)b * A.tstride;
    float *mt = A.mt_out + (size_t)b * A.mtstride;
    float *aux = A.aux_out + (size_t)b * kAuxFloats;
    for (int i = tid; i < W * W; i += 256) {
        gt[(size_t)(i % W) * np + c0 + i / W] = s_d[i];               // gt[col][row]
        mt[(size_t)(i / W) * A.mtld + c0 + i % W] = s_mt[i];           // mt[step][row]
        aux[i] = s_prn[i];
    }
    if (singular && tid == 0 && A.status) atomicMax(&A.status[b], (int)MI32_SINGULAR);
}

// ---- one launch per sub-panel: panel(s) || update(s-1) || strip(s-1) ------------------------------
template <int NT, int RPT, int W, bool FUSED>
constexpr size_t subpanel_lds_bytes(bool with_strip_tiles = true)
{
    const size_t pb = panel_shared_bytes<NT / 64, W>() + (size_t)2 * RPT * NT * sizeof(int);
    const size_t ub = FUSED ? sizeof(UpdateTileShared<W>) * (NT / 256) : 0;
    const size_t ob = with_strip_tiles ? sizeof(OStripShared<W>) * (NT / 256) : 0;  // (they cost the update tiles occupancy)
    const size_t m = pb > ub ? pb : ub;
    return m > ob ? m : ob;
}

// Workgroups [0, batch) (where panel_on) are the panels of sub-panel s; FUSED: the next A.upd_wgs are the update
// tiles of sub-panel s-1 (NT / 256 tiles each); the rest are strip tiles of sub-panel s-1 (NT / 256 each).
template <int NT, int RPT, int W, bool FUSED>
__global__ __launch_bounds__(NT) void gj_subpanel_kernel(SubpanelArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    const int npanel = A.panel_on ? A.batch : 0;
    int u = (int)blockIdx.x;
    if (u < npanel) {
        panel_body<NT, RPT, W, FUSED, false>(A, u, 0, sp_smem);
        return;
    }
    u -= npanel;
    if constexpr (FUSED) {
        if (u < A.upd_wgs) {
            inblock_update_body<W, NT / 256>(A, u, sp_smem);
            return;
        }
        u -= A.upd_wgs;
    }
    const int grp = threadIdx.x >> 8;
    ostrip_body<W>(A, u * (NT / 256) + grp, sp_smem + (size_t)grp * sizeof(OStripShared<W>), threadIdx.x & 255);
}

// A panel of more than kPanelGroupRows rows: A.ngroups workgroups per matrix (all must be resident at once:
// the host only uses this for small batches), kPanelGroupRows rows each; then the strip tiles of the sub-panel before.
template <int W>
__global__ __launch_bounds__(1024) void gj_panel_multi_kernel(SubpanelArgs A)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    static_assert(1024 * 4 == kPanelGroupRows, "1024 threads x 4 rows per lane");
    const int npanel = A.batch * A.ngroups - A.drop_groups;
    if ((int)blockIdx.x < npanel) {
        panel_body<1024, 4, W, false, true>(A, (int)blockIdx.x / A.ngroups, (int)blockIdx.x % A.ngroups, sp_smem);
        return;
    }
    const int grp = threadIdx.x >> 8;
    ostrip_body<W>(A, ((int)blockIdx.x - npanel) * 4 + grp, sp_smem + (size_t)grp * sizeof(OStripShared<W>),
                   threadIdx.x & 255);
}

// update(t) alone: one 64 x 64 tile per 256-thread workgroup; then (the block's last sub-panel) its strip tiles
template <int W>
__global__ __launch_bounds__(256) void gj_inblock_update_kernel(SubpanelArgs A)
{
    constexpr size_t kBytes = sizeof(UpdateTileShared<W>) > sizeof(OStripShared<W>) ? sizeof(UpdateTileShared<W>)
                                                                                    : sizeof(OStripShared<W>);
    __shared__ __attribute__((aligned(16))) unsigned char upd_smem[kBytes];
    if ((int)blockIdx.x < A.upd_wgs) inblock_update_body<W, 1>(A, (int)blockIdx.x, upd_smem);
    else ostrip_body<W>(A, (int)blockIdx.x - A.upd_wgs, upd_smem, threadIdx.x);
}

// ---- the block's pivot-row strips in ONE launch, for the columns strip(t) could not follow -----
// With the look-ahead, the columns outside the block are still being written by the previous block's second-stream
// update while this block's panels run: their strips can only start when that is done.  One workgroup per CT-column
// tile keeps the kb pivot rows x CT columns in the accumulator registers of its 16 waves (one 32 x 32 tile each) and
// runs the block's pivot steps on them, G (= the block's sub-panel width) at a time: the G rows of a group go
// through LDS and strip_step (u_m; G dependent IEEE divisions), then every LATER pivot row takes its G fmaf (one
// v_mfma_f32_32x32x2_f32 chain with the old value as C operand, k ascending).  Out, exactly what the strip(t) tiles
// leave: ub[m][j] = u_m[j], and xs[k][j] = pivot row k after its own sub-panel's last step.
// The groups [g_lo, g_hi) of one call: a block's strips can start before its last panels have run -- the rows of the
// groups still to come are parked in xst in between.  mf[q][m] = -f_m of the row whose index at the start of the
// block was q (own step: -pivot); map = rowsrc.
template <int CT, int G>
constexpr size_t block_strip_lds_bytes(int kb)
{
    return ((size_t)G * (CT + 4) + (size_t)G * (kb + 4) + (size_t)2 * G * (CT + 4)) * sizeof(float) + (size_t)kb * sizeof(int);
}
// kb <= 256 runs CT = 64, wider blocks CT = 32: at most 16 tiles of 32 x 32.  SNT threads: 1024 (16 waves, one tile
// each: a single matrix, where the launch is a chain of rounds on few workgroups) or 512 (8 waves, two tiles each:
// GPU-filling batches -- a round is latency, so two of these per CU, 4 waves per SIMD either way, do twice the tiles).
template <int CT, int G, int SNT>
__global__ __launch_bounds__(SNT, 4) void gj_block_strip_kernel(const float *__restrict__ src_all, size_t mstride, int np, int ld,
                                                              const float *__restrict__ mf_all, size_t mfstride, int mf_ld,
                                                              float *__restrict__ ub_all, float *__restrict__ xs_all,
                                                              float *__restrict__ xst_all, size_t ubstride, int C0, int kb,
                                                              const int *__restrict__ map_all, int col_lo, int col_hi,
                                                              int inside, int g_lo, int g_hi,
                                                              const int *__restrict__ guard)
{
    extern __shared__ __attribute__((aligned(16))) float bs_smem[];
    constexpr int LDX = CT + 4;
    constexpr int NT = SNT;
    constexpr int kStripTPW = 16 / (SNT / 64);  // tiles per wave
    constexpr int CTT = CT / 32;  // tiles per row of tiles
    const int LT = kb + 4;
    float *s_x = bs_smem;                  // [G][LDX]   the rows of the current group
    float *s_lt = s_x + G * LDX;           // [G][LT]    -f of the current G steps, [step][pivot row]
    float *s_u = s_lt + G * LT;            // [2][G][LDX]  u_m of the current G steps (and of the previous G)
    int *s_q = reinterpret_cast<int *>(s_u + 2 * G * LDX);  // [kb] block-start row index of every pivot row

    const int b = blockIdx.y;
    if (matrix_given_up(guard, b)) return;
    const int col0 = blockIdx.x * CT;
    if (col0 >= C0 && col0 < C0 + kb) return;  // the block's own columns are up to date already
    if ((col0 >= col_lo && col0 < col_hi) != (inside != 0)) return;  // the look-ahead splits the columns between two launches
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const float *src = src_all + (size_t)b * mstride;
    const float *mf = mf_all + (size_t)b * mfstride;
    float *ub = ub_all + (size_t)b * ubstride;
    float *xs = xs_all + (size_t)b * ubstride;
    float *xst = xst_all + (size_t)b * ubstride;
    const int *map = map_all + (size_t)b * np;
    const int lcol = lane & 31, lhalf = lane >> 5;
    const int ntiles = (kb / 32) * CTT;

    for (int i = tid; i < kb; i += NT) s_q[i] = map[C0 + i];
    __syncthreads();
    // this wave's tiles of the pivot rows: from the working copy (through the row map) or from where the call for
    // the earlier groups parked them
    float16v acc[kStripTPW];
    // (the source is chosen once, not per value: per value hipcc emits a branch pair and an LDS round trip for the map
    // entry in front of every load; 32-bit byte offsets from the scalar base)
    if (g_lo == 0) {
        const unsigned ld4 = (unsigned)ld * 4u;
        const char *srcb = reinterpret_cast<const char *>(src);
#pragma unroll
        for (int ti = 0; ti < kStripTPW; ++ti) {
            const int t = wave + ti * (NT / 64);
            if (t < ntiles) {
                const int rt = t / CTT;
                const unsigned col4 = (unsigned)(col0 + (t % CTT) * 32 + lcol) * 4u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int4 m4 = *reinterpret_cast<const int4 *>(&s_q[rt * 32 + 8 * q + 4 * lhalf]);
                    const int mm[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[ti][4 * q + j] = *reinterpret_cast<const float *>(srcb + ((unsigned)mm[j] * ld4 + col4));
                }
            }
        }
    } else {
#pragma unroll
        for (int ti = 0; ti < kStripTPW; ++ti) {
            const int t = wave + ti * (NT / 64);
            if (t < ntiles) {
                const int rt = t / CTT, col = col0 + (t % CTT) * 32 + lcol;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int r = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhalf;
                    acc[ti][reg] = xst[(size_t)r * np + col];
                }
            }
        }
    }
    // the multipliers of G steps, all kb pivot rows: requested one round ahead (registers), so that a round is the
    // strip and the update, not a dependent global round trip on top
    constexpr int NL = ((CT == 128 ? 128 : CT == 64 ? 256 : kMaxBW) * (G / 4) + NT - 1) / NT;
    float4 lreg[NL];
    auto load_l = [&](int s0) {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + i * NT;
            if (idx < kb * (G / 4))
                lreg[i] = *reinterpret_cast<const float4 *>(mf + (size_t)s_q[idx / (G / 4)] * mf_ld + s0 + (idx % (G / 4)) * 4);
        }
    };
    auto store_l = [&]() {
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int idx = tid + i * NT;
            if (idx < kb * (G / 4)) {
                const int k = idx / (G / 4), m4 = (idx % (G / 4)) * 4;
                s_lt[(m4 + 0) * LT + k] = lreg[i].x;
                s_lt[(m4 + 1) * LT + k] = lreg[i].y;
                s_lt[(m4 + 2) * LT + k] = lreg[i].z;
                s_lt[(m4 + 3) * LT + k] = lreg[i].w;
            }
        }
    };
    // u_m of G steps -> the rank-bw update's B operand.  Stored one round late, in front of the next request for
    // multipliers: a wave's memory operations complete in order, and the wait for those multipliers at the end of a
    // round must not have to wait for a store issued a moment ago to be acknowledged.
    auto store_u = [&](int s0) {
        const float *su = s_u + ((s0 / G) & 1) * G * LDX;
        for (int idx = tid; idx < G * (CT / 4); idx += NT) {
            const int m = idx / (CT / 4), c4 = (idx % (CT / 4)) * 4;
            *reinterpret_cast<float4 *>(ub + (size_t)(s0 + m) * np + col0 + c4) =
                *reinterpret_cast<const float4 *>(&su[m * LDX + c4]);
        }
    };
    // the rows [O, O + G) of a 32-row tile -> s_x (O a compile-time constant: no run-time index into the registers)
    auto park_group = [&](auto OFF, int rt_o) {
        constexpr int O = decltype(OFF)::value;
#pragma unroll
        for (int ti = 0; ti < kStripTPW; ++ti) {
            const int t = wave + ti * (NT / 64);
            if (t < ntiles && t / CTT == rt_o) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int r = (reg & 3) + 8 * (reg >> 2) + 4 * lhalf;
                    if (r >= O && r < O + G) s_x[(r - O) * LDX + (t % CTT) * 32 + lcol] = acc[ti][reg];
                }
            }
        }
    };
    load_l(g_lo * G);
    store_l();
    for (int gi = g_lo; gi < g_hi; ++gi) {
        const int s0 = gi * G;
        float *su = s_u + (gi & 1) * G * LDX;
        if (gi > g_lo) store_u(s0 - G);
        if (gi + 1 < g_hi) load_l(s0 + G);
        {
            const int rt_o = s0 / 32;
            switch ((s0 % 32) / G) {  // 32 / G cases
            case 0: park_group(std::integral_constant<int, 0>{}, rt_o); break;
            case 1: park_group(std::integral_constant<int, (G < 32 ? G : 0)>{}, rt_o); break;
            case 2: park_group(std::integral_constant<int, (2 * G < 32 ? 2 * G : 0)>{}, rt_o); break;
            case 3: park_group(std::integral_constant<int, (3 * G < 32 ? 3 * G : 0)>{}, rt_o); break;
            case 4: park_group(std::integral_constant<int, (4 * G < 32 ? 4 * G : 0)>{}, rt_o); break;
            case 5: park_group(std::integral_constant<int, (5 * G < 32 ? 5 * G : 0)>{}, rt_o); break;
            case 6: park_group(std::integral_constant<int, (6 * G < 32 ? 6 * G : 0)>{}, rt_o); break;
            default: park_group(std::integral_constant<int, (7 * G < 32 ? 7 * G : 0)>{}, rt_o); break;
            }
        }
        __syncthreads();  // the group's rows and s_lt are in LDS
        if (tid < 4 * CT) {
            constexpr int CPT = G / 4;
            const int c = tid >> 2, q4 = tid & 3;
            float x[CPT];
#pragma unroll
            for (int j = 0; j < CPT; ++j) x[j] = s_x[(CPT * q4 + j) * LDX + c];
            strip_steps<G>(x, s_lt + s0, LT, q4, &su[c], LDX, std::make_integer_sequence<int, G>{});
#pragma unroll
            for (int j = 0; j < CPT; ++j) s_x[(CPT * q4 + j) * LDX + c] = x[j];
        }
        __syncthreads();
        // the group's rows after their own sub-panel: where their accumulation starts in the rank-bw update
        for (int idx = tid; idx < G * (CT / 4); idx += NT) {
            const int k = idx / (CT / 4), c4 = (idx % (CT / 4)) * 4;
            *reinterpret_cast<float4 *>(xs + (size_t)(s0 + k) * np + col0 + c4) =
                *reinterpret_cast<const float4 *>(&s_x[k * LDX + c4]);
        }
        // every later pivot row: x[k][c] = fmaf(-f_m[k], u_m[c], x[k][c]), m ascending (rows of this and of earlier
        // groups in a tile take the same instructions: they are never read again)
#pragma unroll
        for (int ti = 0; ti < kStripTPW; ++ti) {
            const int t = wave + ti * (NT / 64);
            if (t < ntiles && (t / CTT) * 32 + 32 > s0 + G) {
                const int rt = t / CTT, ctl = t % CTT;
#pragma unroll
                for (int kk = 0; kk < G; kk += 2) {
                    const float af = s_lt[(kk + lhalf) * LT + rt * 32 + lcol];
                    const float bf = su[(kk + lhalf) * LDX + ctl * 32 + lcol];
                    acc[ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[ti], 0, 0, 0);
                }
            }
        }
        __syncthreads();  // before s_lt, s_x are overwritten
        if (gi + 1 < g_hi) store_l();
    }
    store_u((g_hi - 1) * G);
    if (g_hi * G < kb) {  // the rows of the groups still to come: parked for the next call
#pragma unroll
        for (int ti = 0; ti < kStripTPW; ++ti) {
            const int t = wave + ti * (NT / 64);
            if (t < ntiles && (t / CTT) * 32 + 32 > g_hi * G) {
                const int rt = t / CTT, col = col0 + (t % CTT) * 32 + lcol;
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int r = rt * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhalf;
                    xst[(size_t)r * np + col] = acc[ti][reg];
                }
            }
        }
    }
}

// Gk[k][row] = mf[map[row]][k], k < kdim: the block's negated multipliers, transposed and in the new row order
// (A operand of the rank-bw update); 64 x 64 tiles through LDS, both global sides coalesced.
// The block's own pivot rows (rows C0 .. C0+kdim-1 of the new order) enter the update with the value the strip of
// their sub-panel left (xs): everything up to the end of that sub-panel is applied already, so their multipliers of
// those steps are replaced by 0 -- fmaf(0, u, x) == x -- and only the later sub-panels' steps reach them.
__global__ __launch_bounds__(256) void gj_mult_transpose_kernel(const float *__restrict__ mf_all, size_t mfstride, int mf_ld,
                                                                 int np, const int *__restrict__ map_all,
                                                                 float *__restrict__ gk_all, size_t gkstride, int C0,
                                                                 int kdim, int w, const int *__restrict__ guard)
{
    __shared__ float t[64][65];
    __shared__ int s_q[64];
    const int b = blockIdx.z;
    if (matrix_given_up(guard, b)) return;
    const int row0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const float *mf = mf_all + (size_t)b * mfstride;
    float *gk = gk_all + (size_t)b * gkstride;
    if (tid < 64) s_q[tid] = (map_all + (size_t)b * np)[row0 + tid];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (tid >> 4) + 16 * q, c4 = (tid & 15) * 4;
        const float4 v = *reinterpret_cast<const float4 *>(mf + (size_t)s_q[r] * mf_ld + k0 + c4);
        const int rel = row0 + r - C0;  // a pivot row of the block: steps below `lim` are applied already
        const int lim = ((unsigned)rel < (unsigned)kdim) ? (rel / w + 1) * w : 0;
        t[r][c4] = (k0 + c4 < lim) ? 0.0f : v.x;
        t[r][c4 + 1] = (k0 + c4 + 1 < lim) ? 0.0f : v.y;
        t[r][c4 + 2] = (k0 + c4 + 2 < lim) ? 0.0f : v.z;
        t[r][c4 + 3] = (k0 + c4 + 3 < lim) ? 0.0f : v.w;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = (tid >> 4) + 16 * q, r4 = (tid & 15) * 4;
        *reinterpret_cast<float4 *>(gk + (size_t)(k0 + k) * np + row0 + r4) =
            make_float4(t[r4][k], t[r4 + 1][k], t[r4 + 2][k], t[r4 + 3][k]);
    }
}

// ---- rank-k update of the next block's columns (look-ahead half (A) of a rank-bw update) -----
//   dst[i][j] = src[map[i]][j] - sum_m f_m[i] * u_m[j]    for the rows i outside the block
// for the 64-column tiles starting at col_lo.  Same arithmetic as the rank-bw kernel of mi32_rank_bw.h (one fmaf
// chain per element from the old value, m ascending), on 64 x 64 tiles because the few columns of one block would
// otherwise make too few workgroups; -f is read from the block's multiplier matrix through the row map, u_m from ub.
template <int BK>
__global__ __launch_bounds__(256) void gj_rank_update_kernel(const float *__restrict__ src_all,
                                                              float *__restrict__ dst_all,
                                                              const float *__restrict__ mf_all, size_t mfstride, int mf_ld,
                                                              const float *__restrict__ ub_all,
                                                              const float *__restrict__ xs_all, size_t ubstride,
                                                              int np, int ld, size_t mstride, int c0, int kdim, int w,
                                                              int col_lo, const int *__restrict__ map_all,
                                                              PanelExport ex, size_t tstride,
                                                              const int *__restrict__ guard)
{
    constexpr int BM = 64, BN = 64;
    constexpr int PADA = (32 / BK) > 0 ? (32 / BK) : 1;
    constexpr int LDA = BM + PADA;
    constexpr int LDB = BN + 4;
    __shared__ float s_a[BK * LDA];
    __shared__ __attribute__((aligned(16))) float s_b[BK * LDB];
    __shared__ int s_map[BM];

    const int b = blockIdx.z;
    if (matrix_given_up(guard, b)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int row0 = blockIdx.y * BM;
    // the block's own pivot rows start from what the strip of their sub-panel left (xs) and only take the later
    // sub-panels' steps (see gj_mult_transpose_kernel)
    const bool tile_in_block = (row0 >= c0 && row0 < c0 + kdim);
    const int col0 = col_lo + blockIdx.x * BN;
    const float *src = src_all + (size_t)b * mstride;
    float *dst = dst_all + (size_t)b * mstride;
    const float *mf = mf_all + (size_t)b * mfstride;
    const float *ub = ub_all + (size_t)b * ubstride;
    const float *xs = xs_all + (size_t)b * ubstride;
    const int *map = map_all + (size_t)b * np;

    for (int i = tid; i < BM; i += 256) s_map[i] = map[row0 + i];
    __syncthreads();

    float16v acc;
    const int lcol = lane & 31;
    const int lhalf = lane >> 5;
    {
        const int col = col0 + wc * 32 + lcol;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int lr = wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhalf;
            acc[reg] = tile_in_block ? xs[(size_t)(row0 + lr - c0) * np + col] : src[(size_t)s_map[lr] * ld + col];
        }
    }
    for (int kt = 0; kt < kdim; kt += BK) {
        // stage A: BM x BK of the row-major multiplier matrix (rows through the map), transposed
#pragma unroll
        for (int q = 0; q < (BM * BK / 4 + 255) / 256; ++q) {
            const int idx = tid + q * 256;
            if (idx < BM * BK / 4) {
                const int rr = idx / (BK / 4), k4 = (idx % (BK / 4)) * 4;
                const float4 v = *reinterpret_cast<const float4 *>(mf + (size_t)s_map[rr] * mf_ld + kt + k4);
                const int lim = tile_in_block ? ((row0 + rr - c0) / w + 1) * w : 0;
                s_a[(k4 + 0) * LDA + rr] = (kt + k4 + 0 < lim) ? 0.0f : v.x;
                s_a[(k4 + 1) * LDA + rr] = (kt + k4 + 1 < lim) ? 0.0f : v.y;
                s_a[(k4 + 2) * LDA + rr] = (kt + k4 + 2 < lim) ? 0.0f : v.z;
                s_a[(k4 + 3) * LDA + rr] = (kt + k4 + 3 < lim) ? 0.0f : v.w;
            }
        }
        // stage B: BK rows of u x BN columns
#pragma unroll
        for (int q = 0; q < (BK * BN / 4 + 255) / 256; ++q) {
            const int idx = tid + q * 256;
            if (idx < BK * BN / 4) {
                const int kk = idx / (BN / 4), c4 = (idx % (BN / 4)) * 4;
                *reinterpret_cast<float4 *>(&s_b[kk * LDB + c4]) =
                    *reinterpret_cast<const float4 *>(ub + (size_t)(kt + kk) * np + col0 + c4);
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float af = s_a[(kk + lhalf) * LDA + wr * 32 + lcol];
            const float bf = s_b[(kk + lhalf) * LDB + wc * 32 + lcol];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    {
        const int col = col0 + wc * 32 + lcol;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int grow = row0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lhalf;
            dst[(size_t)grow * ld + col] = acc[reg];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            panel_export_store4(ex, tstride, b, np, col, row0 + wr * 32 + 8 * q + 4 * lhalf, acc[4 * q], acc[4 * q + 1],
                                acc[4 * q + 2], acc[4 * q + 3]);
    }
}

// ---- getInvertedMatrix counterpart: undo the column permutation ----------------
__global__ void invert_perm_ld_kernel(const int *__restrict__ orig, int *__restrict__ invp, int n, int istride,
                                      const int *__restrict__ guard)
{
    const int b = blockIdx.y;
    if (matrix_given_up(guard, b)) return;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < n) invp[(size_t)b * istride + orig[(size_t)b * istride + c]] = c;
}
// Whole rows go through LDS: the global read (all np columns of R rows) and the global write (n columns)
// are both coalesced; the column gather happens inside LDS.  (A direct gather from global memory read
// 4 scattered bytes per lane: 1.35 ms for 64 x 2048^2, i.e. 1.5 TB/s.)
__global__ __launch_bounds__(256) void unpermute_columns_ld_kernel(const float *__restrict__ w_all, int ld, int np,
                                                                    size_t wstride, const int *__restrict__ invp,
                                                                    int istride, int n, int rows_per_block,
                                                                    float *__restrict__ out,
                                                                    const int *__restrict__ status)
{
    extern __shared__ __attribute__((aligned(16))) float s_rows[];  // [rows_per_block][np]
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const float *w = w_all + (size_t)b * wstride;
    float *o = out + (size_t)b * n * n;
    const int i0 = blockIdx.x * rows_per_block;
    const int nr = (n - i0 < rows_per_block) ? (n - i0) : rows_per_block;
    // a matrix whose shared panel lost a partner workgroup went on with stale data: hand out NaN, not numbers
    if (status != nullptr && __builtin_amdgcn_readfirstlane(status[b]) == MI32_RUNTIME_ERROR) {
        for (int r = 0; r < nr; ++r)
            for (int j = tid; j < n; j += 256) o[(size_t)(i0 + r) * n + j] = __builtin_nanf("");
        return;
    }
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


// tests only: leave the last panel workgroup of every multi-workgroup panel launch out (see dispatch_subpanel)
static std::atomic<int> g_debug_drop_panel_group{0};
extern "C" int mi32_debug_drop_panel_group(int enable)
{
    g_debug_drop_panel_group.store(enable ? 1 : 0, std::memory_order_relaxed);
    return 0;
}

template <int NT, int RPT, int W, bool FUSED>
static hipError_t launch_subpanel(const SubpanelArgs &A, int nwgs, hipStream_t stream)
{
    constexpr size_t lds = subpanel_lds_bytes<NT, RPT, W, FUSED>();
    if (lds > 48 * 1024) {  // more dynamic LDS than the default limit: raise it once per device (any thread may be first)
        static std::once_flag once[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        hipError_t e = hipSuccess;
        std::call_once(once[dev & 63], [&] {
            e = hipFuncSetAttribute((const void *)gj_subpanel_kernel<NT, RPT, W, FUSED>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        });
        if (e != hipSuccess) return e;
    }
    const size_t lds_now = subpanel_lds_bytes<NT, RPT, W, FUSED>(A.os_on != 0);
    hipLaunchKernelGGL((gj_subpanel_kernel<NT, RPT, W, FUSED>), dim3(nwgs), dim3(NT), lds_now, stream, A);
    return hipSuccess;
}

// One launch of the sub-panel pipeline:
//  * panel(s) and update(s-1) together (fused blocks): the workgroup size is the panel's, the update tiles are
//    packed NT / 256 to a workgroup;
//  * panel(s) alone: the smallest thread geometry that holds its rows (fewer waves and fewer rows per lane both
//    shorten a pivot step);
//  * update(t) alone: 256-thread workgroups, one tile each.
static hipError_t dispatch_subpanel(const BlockedPlan &p, int w, const SubpanelArgs &A0, hipStream_t stream)
{
    SubpanelArgs A = A0;
    const int tiles = A.upd_on ? (A.kb / 64) * (p.np / 64) : 0;
    const int os_tiles = A.os_on ? A.batch * A.os_ntiles : 0;  // strip tiles of columns outside the block
    if (!A.panel_on) {
        A.upd_wgs = A.batch * tiles;
        const dim3 grid(A.upd_wgs + os_tiles);
        if (w == 32) hipLaunchKernelGGL((gj_inblock_update_kernel<32>), grid, dim3(256), 0, stream, A);
        else if (w == 16) hipLaunchKernelGGL((gj_inblock_update_kernel<16>), grid, dim3(256), 0, stream, A);
        else if (w == 8) hipLaunchKernelGGL((gj_inblock_update_kernel<8>), grid, dim3(256), 0, stream, A);
        else hipLaunchKernelGGL((gj_inblock_update_kernel<4>), grid, dim3(256), 0, stream, A);
        return hipSuccess;
    }
    if (A.ngroups > 1) {  // multi-workgroup panel: never fused, W = 16 (what the plan gives every block then)
        if (A.upd_on || w != 16) return hipErrorInvalidValue;
        constexpr size_t lds = subpanel_lds_bytes<1024, 4, 16, false>();
        // mi32_debug_drop_panel_group(1) (tests only, host side only): the last panel workgroup of the grid is never
        // launched, i.e. one panel loses a partner -- what a foreign kernel holding the CUs would cause
        A.drop_groups = g_debug_drop_panel_group.load(std::memory_order_relaxed) ? 1 : 0;
        const size_t lds_now = A.os_on ? lds : subpanel_lds_bytes<1024, 4, 16, false>(false);
        hipLaunchKernelGGL((gj_panel_multi_kernel<16>), dim3(A.batch * A.ngroups - A.drop_groups + (os_tiles + 3) / 4),
                           dim3(1024), lds_now, stream, A);
        return hipSuccess;
    }
    const bool fused = A.upd_on != 0;
    int nt, rpt;
    panel_geometry(p, p.np - A.row_lo, nt, rpt);
    A.upd_wgs = A.batch * (tiles / (nt / 256));
    const int nwgs = A.batch + A.upd_wgs + (os_tiles + nt / 256 - 1) / (nt / 256);
#define MI32_SUBPANEL_CASE(T, R, WW)                                                                   \
    if (nt == T && rpt == R && w == WW && !fused) return launch_subpanel<T, R, WW, false>(A, nwgs, stream);
#define MI32_SUBPANEL_FUSED(T, R, WW)                                                                  \
    if (nt == T && rpt == R && w == WW && fused) return launch_subpanel<T, R, WW, true>(A, nwgs, stream);
#ifdef MI32_EXPERIMENT_MIN  // compile-time experiments (tools/build_check.sh): the two instances C1 runs most
    MI32_SUBPANEL_CASE(1024, 4, 16) MI32_SUBPANEL_FUSED(1024, 2, 16)
#else
    MI32_SUBPANEL_CASE(256, 1, 32) MI32_SUBPANEL_CASE(256, 1, 16) MI32_SUBPANEL_CASE(256, 1, 8) MI32_SUBPANEL_CASE(256, 1, 4)
    MI32_SUBPANEL_CASE(512, 1, 32) MI32_SUBPANEL_CASE(512, 2, 32) MI32_SUBPANEL_CASE(512, 4, 32)
    MI32_SUBPANEL_CASE(1024, 1, 32) MI32_SUBPANEL_CASE(1024, 2, 32)
    MI32_SUBPANEL_CASE(512, 1, 16) MI32_SUBPANEL_CASE(512, 2, 16) MI32_SUBPANEL_CASE(512, 4, 16) MI32_SUBPANEL_CASE(512, 8, 16)
    MI32_SUBPANEL_CASE(512, 1, 8) MI32_SUBPANEL_CASE(512, 2, 8) MI32_SUBPANEL_CASE(512, 4, 8) MI32_SUBPANEL_CASE(512, 8, 8)
    MI32_SUBPANEL_CASE(512, 1, 4) MI32_SUBPANEL_CASE(512, 2, 4) MI32_SUBPANEL_CASE(512, 4, 4) MI32_SUBPANEL_CASE(512, 8, 4)
    MI32_SUBPANEL_CASE(1024, 1, 16) MI32_SUBPANEL_CASE(1024, 2, 16) MI32_SUBPANEL_CASE(1024, 4, 16)
    MI32_SUBPANEL_CASE(1024, 1, 8) MI32_SUBPANEL_CASE(1024, 2, 8) MI32_SUBPANEL_CASE(1024, 4, 8) MI32_SUBPANEL_CASE(1024, 8, 8)
    MI32_SUBPANEL_CASE(1024, 1, 4) MI32_SUBPANEL_CASE(1024, 2, 4) MI32_SUBPANEL_CASE(1024, 4, 4) MI32_SUBPANEL_CASE(1024, 8, 4)
    MI32_SUBPANEL_CASE(1024, 16, 4)
    // fused launches exist for the geometries of at most kFusedRows rows (blocked_invert)
    MI32_SUBPANEL_FUSED(256, 1, 32) MI32_SUBPANEL_FUSED(256, 1, 16) MI32_SUBPANEL_FUSED(256, 1, 8) MI32_SUBPANEL_FUSED(256, 1, 4)
    MI32_SUBPANEL_FUSED(512, 1, 32) MI32_SUBPANEL_FUSED(512, 2, 32) MI32_SUBPANEL_FUSED(512, 4, 32)
    MI32_SUBPANEL_FUSED(1024, 1, 32) MI32_SUBPANEL_FUSED(1024, 2, 32)
    MI32_SUBPANEL_FUSED(512, 1, 16) MI32_SUBPANEL_FUSED(512, 2, 16) MI32_SUBPANEL_FUSED(512, 4, 16)
    MI32_SUBPANEL_FUSED(1024, 1, 16) MI32_SUBPANEL_FUSED(1024, 2, 16)
    MI32_SUBPANEL_FUSED(512, 1, 8) MI32_SUBPANEL_FUSED(512, 2, 8) MI32_SUBPANEL_FUSED(512, 4, 8)
    MI32_SUBPANEL_FUSED(1024, 1, 8) MI32_SUBPANEL_FUSED(1024, 2, 8)
    MI32_SUBPANEL_FUSED(512, 1, 4) MI32_SUBPANEL_FUSED(512, 2, 4) MI32_SUBPANEL_FUSED(512, 4, 4)
    MI32_SUBPANEL_FUSED(1024, 1, 4) MI32_SUBPANEL_FUSED(1024, 2, 4)
#endif
#undef MI32_SUBPANEL_CASE
#undef MI32_SUBPANEL_FUSED
    return hipErrorInvalidValue;
}

template <int CT, int G, int SNT>
static hipError_t launch_block_strip_t(dim3 grid, size_t lds, hipStream_t st, const float *src, size_t mstride, int np, int ld,
                                       const float *mf, size_t mfstride, int mf_ld, float *ub, float *xs, float *xst,
                                       size_t ubstride, int C0, int kb, const int *map, int col_lo, int col_hi, int inside,
                                       int g_lo, int g_hi, const int *guard)
{
    static std::once_flag once[64];  // function attributes are per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipError_t e = hipSuccess;
    std::call_once(once[dev & 63], [&] {
        e = hipFuncSetAttribute((const void *)gj_block_strip_kernel<CT, G, SNT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)block_strip_lds_bytes<CT, G>(CT == 128 ? 128 : CT == 64 ? 256 : kMaxBW));
    });
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gj_block_strip_kernel<CT, G, SNT>), grid, dim3(SNT), lds, st, src, mstride, np, ld, mf, mfstride,
                       mf_ld, ub, xs, xst, ubstride, C0, kb, map, col_lo, col_hi, inside, g_lo, g_hi, guard);
    return hipSuccess;
}
// the block's strips, sub-panels [g_lo, g_hi), for the columns outside the block that lie in [col_lo, col_hi) (inside) /
// that do not
static hipError_t launch_block_strip(int w, int batch, hipStream_t st, const float *src, size_t mstride, int np, int ld,
                                     const float *mf, size_t mfstride, int mf_ld, float *ub, float *xs, float *xst,
                                     size_t ubstride, int C0, int kb, const int *map, int col_lo, int col_hi, int inside,
                                     int g_lo, int g_hi, const int *guard)
{
#define MI32_STRIP_CASE(GG)                                                                                               \
    if (w == GG && batch * (np / 64) > 512) {  /* GPU-filling: small workgroups */                                        \
        if (kb <= 128 && C0 % 128 == 0 && np % 128 == 0)  /* 128 columns: the strip uses all 512 threads */               \
            return launch_block_strip_t<128, GG, 512>(dim3(np / 128, batch), block_strip_lds_bytes<128, GG>(kb), st, src, mstride, \
                                                      np, ld, mf, mfstride, mf_ld, ub, xs, xst, ubstride, C0, kb, map, col_lo, \
                                                      col_hi, inside, g_lo, g_hi, guard);                                 \
        if (kb <= 256)                                                                                                    \
            return launch_block_strip_t<64, GG, 512>(dim3(np / 64, batch), block_strip_lds_bytes<64, GG>(kb), st, src, mstride, \
                                                     np, ld, mf, mfstride, mf_ld, ub, xs, xst, ubstride, C0, kb, map, col_lo, \
                                                     col_hi, inside, g_lo, g_hi, guard);                                  \
        return launch_block_strip_t<32, GG, 512>(dim3(np / 32, batch), block_strip_lds_bytes<32, GG>(kb), st, src, mstride, np, \
                                                 ld, mf, mfstride, mf_ld, ub, xs, xst, ubstride, C0, kb, map, col_lo, col_hi, \
                                                 inside, g_lo, g_hi, guard);                                              \
    }                                                                                                                     \
    if (w == GG) {                                                                                                        \
        if (kb <= 256)                                                                                                    \
            return launch_block_strip_t<64, GG, 1024>(dim3(np / 64, batch), block_strip_lds_bytes<64, GG>(kb), st, src, mstride, np, \
                                                ld, mf, mfstride, mf_ld, ub, xs, xst, ubstride, C0, kb, map, col_lo, col_hi, \
                                                inside, g_lo, g_hi, guard);                                               \
        return launch_block_strip_t<32, GG, 1024>(dim3(np / 32, batch), block_strip_lds_bytes<32, GG>(kb), st, src, mstride, np, ld, \
                                            mf, mfstride, mf_ld, ub, xs, xst, ubstride, C0, kb, map, col_lo, col_hi, inside, \
                                            g_lo, g_hi, guard);                                                           \
    }
    MI32_STRIP_CASE(16) MI32_STRIP_CASE(8) MI32_STRIP_CASE(4) MI32_STRIP_CASE(32)
#undef MI32_STRIP_CASE
    return hipErrorInvalidValue;
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
    // (measured in round 1: 8192^2 51 -> 45 ms, 16384^2 399 -> 330 ms, 4096^2 11.2 -> 11.0 ms, 2048^2 4.2 -> 4.4 ms; with
    // the half on CUs of its own, round 2: 3584^2 7.39 -> 7.00 ms, 3072^2 5.79 -> 5.63, 2560^2 4.32 -> 4.34, 2048^2 3.01 -> 3.14)
    // Round 3 (reference-order arithmetic: the pivot rows' strip per block sits between the block's last panel and its
    // rank-bw update, and rides in the panel launches only WITHOUT the second stream), with / without:
    // 4096^2 10.23 / 10.09, 4352^2 11.50 / 11.66, 5120^2 15.02 / 15.79, 8192^2 34.0 / 37.1 -> on above 4096 padded rows.
    int la_min = 4096 + 1;
    if (const char *ev = std::getenv("MI32_LOOKAHEAD_MIN")) la_min = std::atoi(ev) > 2048 ? std::atoi(ev) : 2048;
    const bool lookahead = ex.aux != nullptr && ex.n_events >= 4 && ex.aux_workgroups > 0 && batch == 1 && np >= la_min;
    hipError_t e;
    int fused_rows = 2048;  // see "Fused mode" below; fused instances exist for at most 2048 rows
    if (const char *ev = std::getenv("MI32_FUSED_ROWS")) fused_rows = std::atoi(ev) < 2048 ? std::atoi(ev) : 2048;
    const PanelExport no_export = {ws.pt[0], ws.pt_bstride, -(1 << 30), 1, 0};
    if (d_status) {  // MI32_OK; the init kernel flags non-finite input, the panels bad pivots and lost partners
        if ((e = hipMemsetAsync(d_status, 0, sizeof(int) * (size_t)batch, stream)) != hipSuccess) return e;
    }
    {
        // the first two sub-panels of the first block are exported as they are: the first has no pending
        // update at all, the second gets the first one's update in its panel's prologue
        ProfScope ps(prof, KC_INIT, stream);
        const PanelExport ex0 = {ws.pt[0], ws.pt_bstride, 0, ex.pivoting ? (int)p.wblk[0] : 16,
                                 (ex.pivoting && np <= fused_rows) ? 2 : 1};
        hipLaunchKernelGGL(blocked_init_kernel, dim3((np + 255) / 256, (np + 15) / 16, batch), dim3(256), 0, stream,
                           d_a, p.n, np, p.ld, ws.mstride, ws.m0, ex0, ws.tstride, ws.orig, d_status);
    }
    if (lookahead) {  // whatever still runs on the second stream from an earlier call shares this workspace
        if ((e = hipEventRecord(ex.events[0], ex.aux)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(stream, ex.events[0], 0)) != hipSuccess) return e;
    }
    // dynamic LDS of the rank-bw kernels: operand stages + maps; the persistent flavour asks for more than half
    // a CU's LDS so that at most one of its workgroups is resident per CU
    // "exclusive": nearly all of a CU's LDS, so that no workgroup of the main stream fits beside a look-ahead workgroup.
    // The dispatcher deals a grid's workgroups to the XCDs and shader engines in turn and puts each on the FIRST CU of
    // its engine that has room -- with 84 KB the in-block update tiles and the small panels land on the CUs the
    // look-ahead half keeps busy although whole CUs are idle (in-block update 12.2 instead of 6.4 us while the half
    // runs).  Where the half is short against the panel phase (up to ~8192 rows) it gets fewer CUs, all to itself.
    const size_t lds_persistent = (ex.aux_exclusive ? 156 : 84) * 1024;
    {
        static std::once_flag once[64];  // function attributes are per device; any thread may be the first
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::call_once(once[dev & 63], [] {
            (void)hipFuncSetAttribute((const void *)gj_rank_bw2_kernel<MI32_BW_BK, MI32_BW_WPS, 128, (MI32_BW_PF != 0)>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)rank_bw2_lds_bytes<MI32_BW_BK>(kMaxBW));
            (void)hipFuncSetAttribute((const void *)gj_rank_bw2_persistent_kernel<MI32_BW_BK>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
            (void)hipFuncSetAttribute((const void *)gj_panel_multi_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)subpanel_lds_bytes<1024, 4, 16, false>());
        });
    }
    // plans with shared panels: every launch skips a matrix whose panel lost a partner (SubpanelArgs::guard)
    const int *guard = (p.multi_panel && ex.pivoting) ? d_status : nullptr;
    unsigned panel_launches = 0;  // tags of the multi-workgroup panels' exchange granules: unique per launch
    if (p.multi_panel) {  // no stale tag of an earlier call may match
        if ((e = hipMemsetAsync(ws.xch, 0, (size_t)kXchGranules * sizeof(unsigned long long) * batch, stream)) != hipSuccess)
            return e;
    }
    // MI32_STRIP_ONE_LAUNCH=1 (diagnostic): every block's strips in one launch at its end, no strip(t) tiles
    const char *sol = std::getenv("MI32_STRIP_ONE_LAUNCH");
    // GPU-filling batches: the strip(t) tiles (256-thread groups, one global round trip per 32 earlier steps) cost
    // more than the one launch per block (measured 64 x 2048^2: 23.0 vs 21.7 ms)
    const bool one_launch_strips = !lookahead && ((sol && std::atoi(sol) != 0) || batch * ((np + 63) / 64) > 256);
    float *cur = ws.m0, *oth = ws.m1;
    bool pending_b = false;  // a (B) half is in flight on the second stream
    int blk = 0, ev = 0;
    for (int C0 = 0; C0 < np; C0 += p.bw, ++blk) {
        const int kb = (C0 + p.bw <= np) ? p.bw : np - C0;
        int **rsb = &ws.rowsrc[2 * (blk & 1)];
        float *mf = ws.mf[blk & 1], *ub = ws.ub[blk & 1], *xs = ws.xs[blk & 1];
        // sub-panel width of this block and of the next one (the no-pivot variant has no register-resident panel
        // whose rows would limit it)
        const int w = ex.pivoting ? (int)p.wblk[blk] : 16;
        const int w_next = !ex.pivoting ? 16 : (blk + 1 < p.nblk) ? (int)p.wblk[blk + 1] : w;
        const int S = kb / w;                                            // sub-panels of this block (even)
        // Fused mode: launch s = panel(s) || update(s-1), the panel applies update(s-1) to its own columns in a
        // prologue.  It pays while the panel workgroup holds at most 2 rows per lane (measured: 2048^2 3.23 ->
        // 3.06 ms, 1024^2 1.40 -> 1.27 ms); with more rows the prologue (rows x W x W fmaf on ONE CU) costs what
        // the update launch did (4096^2: 8.9 -> 9.5 ms), so those blocks keep panel(s) and update(s) apart.
        const bool fused = ex.pivoting && (np - C0) <= fused_rows;
        // The strip(t) tiles follow the block sub-panel by sub-panel in the columns outside it -- unless the look-ahead
        // is on: those columns are then still being written by the previous block's second-stream update while this
        // block's panels run (the next block's columns too: half (A) of the previous block covered THIS block's), and
        // the block's strips run in one launch at its end (gj_block_strip_kernel).
        const int os_first = 0;
        const int os_ntiles = (one_launch_strips || lookahead) ? 0 : (np - kb) / 64;
        const bool strips_at_end = one_launch_strips || lookahead;
        float *x = cur, *y = oth;  // the block's panel columns alternate between the two copies
        for (int s = 0; s <= S; ++s) {
            SubpanelArgs P = {};   // the panel half
            P.np = np; P.n = p.n; P.ld = p.ld; P.batch = batch;
            P.mstride = ws.mstride; P.tstride = ws.tstride;
            P.mtstride = ws.mtstride; P.mtld = ws.mtld;
            P.mfstride = ws.mfstride; P.mf_ld = p.bw;
            P.u_exp = no_export;
            P.guard = guard;
            SubpanelArgs U = P;    // the update half
            if (s < S) {
                P.panel_on = 1;
                P.c0 = C0 + s * w;
                P.has_prev = fused && (s > 0);
                P.c0_prev = P.c0 - w;
                P.row_lo = P.has_prev ? P.c0_prev : P.c0;  // fused: the W pivot rows of s-1 are needed once more
                P.first_in_block = (s == 0);
                P.pt_in = ws.pt[s % 3];
                P.mt_prev = ws.mt[(s + 1) & 1];
                P.gt_out = ws.gt[s & 1];
                P.mt_out = ws.mt[s & 1];
                P.submap_prev = ws.submap[(s + 1) & 1];
                P.invsub_prev = ws.invsub[(s + 1) & 1];
                P.submap_out = ws.submap[s & 1];
                P.invsub_out = ws.invsub[s & 1];
                if (!ex.pivoting) P.ngroups = 1;
                P.rowsrc_in = rsb[fused ? (s + 1) & 1 : 0];
                P.rowsrc_out = rsb[fused ? s & 1 : 0];
                P.rowsrc_alt = (fused && s == 0) ? rsb[1] : nullptr;
                P.orig = ws.orig;
                P.aux_out = ws.aux[s & 1];
                P.status = d_status;
                const int prow = np - P.row_lo;  // rows the panel holds
                P.ngroups = (p.multi_panel && prow > kPanelGroupRows) ? (prow + kPanelGroupRows - 1) / kPanelGroupRows : 1;
                P.xch = ws.xch;
                P.tag_base = ++panel_launches;
            }
            if (s > 0) {
                const int t = s - 1;
                U.upd_on = 1;
                U.u_c0 = C0 + t * w;
                U.u_has_prev = fused && (t > 0);
                U.u_above_hi = U.u_has_prev ? U.u_c0 - w : U.u_c0;  // = the first row panel(t) held
                U.u_panel_hi = ex.pivoting ? np : U.u_c0 + w;
                U.C0 = C0; U.kb = kb;
                U.x = x; U.y = y;
                U.u_gt = ws.gt[t & 1];
                U.u_mt = ws.mt[t & 1];
                U.u_rowsrc = ex.pivoting ? rsb[fused ? t & 1 : 0] : ws.orig;  // no pivoting: no row ever moves
                U.u_mf = mf;
                U.u_submap = ex.pivoting ? ws.submap[t & 1] : ws.orig;
                U.u_pt_in = ws.pt[t % 3];
                U.u_aux = ws.aux[t & 1];
                // fused: sub-panel t+2's columns (t+1's are brought up to date by its own panel);
                // otherwise sub-panel t+1's, fully up to date
                const int tx = fused ? t + 2 : t + 1;
                if (tx < S) U.u_exp = PanelExport{ws.pt[tx % 3], ws.pt_bstride, C0 + tx * w, w, 1};
            }
            if (s > 0 && os_ntiles > 0) {
                // strip(s-1) of the columns outside the block: rides in the launch of panel(s) (in the block's last
                // in-block update for the last sub-panel); it needs panel(s-1)'s output and the strips before it
                SubpanelArgs &O = (fused || s < S) ? P : U;
                O.os_on = 1;
                O.os_first = os_first; O.os_ntiles = os_ntiles;
                O.os_cur = cur;
                O.os_ub = ub; O.os_xs = xs; O.ubstride = ws.gkstride;
                O.u_c0 = U.u_c0; O.C0 = C0; O.kb = kb;
                O.u_mt = U.u_mt; O.u_submap = U.u_submap; O.u_rowsrc = U.u_rowsrc; O.u_mf = U.u_mf;
            }
            if (fused) {
                SubpanelArgs A = P;  // one launch: panel(s) || update(s-1) || strip(s-1)
                A.upd_on = U.upd_on; A.u_c0 = U.u_c0; A.u_has_prev = U.u_has_prev; A.u_above_hi = U.u_above_hi;
                A.u_panel_hi = U.u_panel_hi;
                A.C0 = U.C0; A.kb = U.kb; A.x = U.x; A.y = U.y; A.u_gt = U.u_gt; A.u_submap = U.u_submap;
                A.u_mt = U.u_mt; A.u_rowsrc = U.u_rowsrc; A.u_mf = U.u_mf;
                A.u_pt_in = U.u_pt_in; A.u_aux = U.u_aux; A.u_exp = U.u_exp;
                // a fused launch is accounted to the panel while there is one (it is the critical path)
                ProfScope ps(prof, A.panel_on ? KC_PANEL : KC_UPDATE_IN, stream);
                if ((e = dispatch_subpanel(p, w, A, stream)) != hipSuccess) return e;
            } else {
                if (U.upd_on) {  // update(s-1) first: panel(s) reads the columns it exports
                    ProfScope ps(prof, KC_UPDATE_IN, stream);
                    if ((e = dispatch_subpanel(p, w, U, stream)) != hipSuccess) return e;
                }
                if (P.panel_on && !ex.pivoting) {
                    // the no-pivot variant: the W x W diagonal block alone (+ the strip tiles that ride with a panel)
                    ProfScope ps(prof, KC_PANEL, stream);
                    const int os_tiles = P.os_on ? batch * P.os_ntiles : 0;
                    hipLaunchKernelGGL((gj_diag_panel_kernel<16>), dim3(batch + os_tiles), dim3(256), 0, stream, P);
                } else if (P.panel_on) {
                    ProfScope ps(prof, KC_PANEL, stream);
                    if ((e = dispatch_subpanel(p, w, P, stream)) != hipSuccess) return e;
                }
            }
            if (s > 0) { float *t2 = x; x = y; y = t2; }
        }
        // x now holds the block's own columns; every other column is still valid in `cur` only
        if (kb < np) {
            const int next = C0 + kb;  // first column of the next block
            const bool has_next = next < np;
            const int kb_next = has_next ? ((next + p.bw <= np) ? p.bw : np - next) : 0;
            // position after the block -> row index at its start
            const int *rowsrc = ex.pivoting ? rsb[fused ? (S - 1) & 1 : 0] : ws.orig;
            // the next block's first two sub-panels, fully updated, for its first two panels
            const PanelExport exn =
                has_next ? PanelExport{ws.pt[0], ws.pt_bstride, next, w_next, (ex.pivoting && (np - next) <= fused_rows) ? 2 : 1}
                         : no_export;
            const int copy = (x != oth) ? 1 : 0;
            if (pending_b) {  // this update reads all of `cur` and overwrites `oth`: the previous (B) must be done
                if ((e = hipStreamWaitEvent(stream, ex.events[ev], 0)) != hipSuccess) return e;
                pending_b = false;
            }
            const bool split_update = lookahead && has_next;
            auto launch_transpose = [&](hipStream_t st) {  // A operand of the rank-bw update, k-major (mi32_rank_bw.h)
                ProfScope ps(prof, KC_TRANSPOSE, st);
                hipLaunchKernelGGL(gj_mult_transpose_kernel, dim3(np / 64, kb / 64, batch), dim3(256), 0, st, mf,
                                   ws.mfstride, p.bw, np, rowsrc, ws.gk, ws.gkstride, C0, kb, w, guard);
            };
            if (split_update) {
                {   // (A): the next block's columns, on the main stream; exports the next sub-panels
                    {
                        ProfScope ps(prof, KC_TRANSPOSE, stream);
                        if ((e = launch_block_strip(w, batch, stream, cur, ws.mstride, np, p.ld, mf, ws.mfstride, p.bw, ub, xs,
                                                    ws.xst, ws.gkstride, C0, kb, rowsrc, next, next + kb_next, 1, 0, S,
                                                    guard)) != hipSuccess)
                            return e;
                    }
                    ProfScope ps(prof, KC_UPDATE_OUT, stream);
                    // small tiles: only kb_next columns, so 64x64 gives 4x the workgroups of 128x128
                    hipLaunchKernelGGL((gj_rank_update_kernel<32>), dim3(kb_next / 64, np / 64, batch), dim3(256), 0,
                                       stream, cur, oth, mf, ws.mfstride, p.bw, ub, xs, ws.gkstride, np, p.ld, ws.mstride, C0, kb,
                                       w, next, rowsrc, exn, ws.tstride, guard);
                }
                // (B): everything else, on the second stream, after this block's panel phase
                ev = (ev + 1) % (ex.n_events / 2);
                hipEvent_t e_panel = ex.events[ex.n_events / 2 + ev];
                if ((e = hipEventRecord(e_panel, stream)) != hipSuccess) return e;
                if ((e = hipStreamWaitEvent(ex.aux, e_panel, 0)) != hipSuccess) return e;
                {   // the strips of every column the strip(t) tiles could not follow
                    ProfScope ps(prof, KC_TRANSPOSE, ex.aux);
                    if ((e = launch_block_strip(w, batch, ex.aux, cur, ws.mstride, np, p.ld, mf, ws.mfstride, p.bw, ub, xs,
                                                ws.xst, ws.gkstride, C0, kb, rowsrc, next, next + kb_next, 0, 0, S,
                                                guard)) != hipSuccess)
                        return e;
                }
                launch_transpose(ex.aux);  // only half (B) reads the transposed multipliers: off the main stream
                {
                    ProfScope ps(prof, KC_UPDATE_OUT, ex.aux);
                    // persistent flavour: aux_workgroups (< number of CUs) workgroups, with so much dynamic LDS
                    // that one CU holds at most one of them -> the remaining CUs stay free for the main stream
                    hipLaunchKernelGGL((gj_rank_bw2_persistent_kernel<MI32_BW_BK>), dim3(ex.aux_workgroups, batch),
                                       dim3(256), lds_persistent, ex.aux, cur, oth, x, ws.mstride, ws.gk, ws.gkstride, ub, xs, np,
                                       p.ld, ws.mstride, C0, kb, rowsrc, copy, no_export, ws.tstride, next,
                                       next + kb_next, guard);
                }
                if ((e = hipEventRecord(ex.events[ev], ex.aux)) != hipSuccess) return e;
                pending_b = true;
            } else {
                if (strips_at_end) {  // no strip(t) tiles ran
                    ProfScope ps(prof, KC_TRANSPOSE, stream);
                    if ((e = launch_block_strip(w, batch, stream, cur, ws.mstride, np, p.ld, mf, ws.mfstride, p.bw, ub, xs,
                                                ws.xst, ws.gkstride, C0, kb, rowsrc, 0, 0, 0, 0, S, guard)) != hipSuccess)
                        return e;
                }
                launch_transpose(stream);
                ProfScope ps(prof, KC_UPDATE_OUT, stream);
                hipLaunchKernelGGL((gj_rank_bw2_kernel<MI32_BW_BK, MI32_BW_WPS, 128, (MI32_BW_PF != 0)>), dim3((np / 128) * (np / 128), batch),
                                   dim3(256), rank_bw2_lds_bytes<MI32_BW_BK>(kb), stream, cur, oth, x, ws.mstride, ws.gk,
                                   ws.gkstride, ub, xs, np, p.ld, ws.mstride, C0, kb, rowsrc, copy, exn, ws.tstride, 0, 0, guard);
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
                       np, np, guard);
    {
        int rpb = (64 * 1024) / (np * (int)sizeof(float));  // rows per workgroup: at most 64 KiB of LDS
        if (rpb < 1) rpb = 1;
        if (rpb > 8) rpb = 8;
        hipLaunchKernelGGL(unpermute_columns_ld_kernel, dim3((p.n + rpb - 1) / rpb, batch), dim3(256),
                           (size_t)rpb * np * sizeof(float), stream, cur, p.ld, np, ws.mstride, ws.invp, np, p.n, rpb,
                           d_inv, d_status);
    }
    return hipGetLastError();
}

#ifdef MI32_PANEL_STAMPS
// diagnostic builds only: where the panel kernels write their stamps (device buffer of 1024 x 64 u64, or NULL)
extern "C" int mi32_debug_panel_stamps(unsigned long long *dev_buf)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_panel_stamps), &dev_buf, sizeof(dev_buf)) == hipSuccess ? 0 : 3;
}
#endif

}  // namespace mi32
