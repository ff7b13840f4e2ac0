// mi32_internal.h -- shared declarations of libmat_inv_32.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mat_inv_32_c.h"

namespace mi32 {

// ---- pivot records ----------------------------------------------------------
// One 64-bit key per candidate: (bits of |a| << 32) | ~row.  |a| >= 0, so its
// IEEE bit pattern orders like the value; the inverted row index makes an
// unsigned max pick the LOWEST row among equal maxima (the reference's scan
// keeps the first maximum, mat_inv_32.cpp:121-127).  NaN candidates and "no
// candidate" are key 0, which loses against every real candidate.
__device__ __forceinline__ unsigned long long pivot_key(float a, int row)
{
    const float v = __builtin_fabsf(a);
    if (!(v == v)) return 0ull;
    return ((unsigned long long)__float_as_uint(v) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)row);
}
__device__ __forceinline__ int pivot_key_row(unsigned long long key, int fallback_row)
{
    return key == 0ull ? fallback_row : (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(k, off, 64);
        k = o > k ? o : k;
    }
    return k;
}

// ---- optional per-kernel-class timing (HIP events on the launch stream) -------------
// Off by default.  When a context enables it, every launch is bracketed by two
// events recorded on the stream the kernel is launched on; the classes mirror the
// reference's per-phase timing slots (FP32_bench.cpp:256-443: makeAug, pivot, row,
// column, getInverted).
enum KernelClass {
    KC_INIT = 0,       // makeAugmented counterpart
    KC_SWEEP_STEP = 1, // fused pivot step of the sweep path
    KC_PANEL = 2,      // register-resident panel steps of the blocked path
    KC_UPDATE_IN = 3,  // rank-w update inside a block
    KC_UPDATE_OUT = 4, // rank-bw update of the rest of the matrix (fp32 MFMA)
    KC_FINISH = 5,     // getInverted counterpart (column un-permutation)
    KC_TRANSPOSE = 6,  // multiplier transposition in front of a rank-bw update (A operand, k-major)
    KC_COUNT = 7
};
struct Profiler {
    virtual void begin(int kclass, hipStream_t s) = 0;
    virtual void end(int kclass, hipStream_t s) = 0;
    virtual ~Profiler() {}
};
struct ProfScope {
    Profiler *p; int k; hipStream_t s;
    ProfScope(Profiler *p_, int k_, hipStream_t s_) : p(p_), k(k_), s(s_) { if (p) p->begin(k, s); }
    ~ProfScope() { if (p) p->end(k, s); }
};

// ---- launch plumbing ----------------------------------------------------------
struct SweepPlan {
    int n;        // matrix order
    int ld;       // leading dimension of the working copies (n rounded up to 4)
    int tr;       // rows per workgroup of the step kernel
    int row_tiles;
    int col_tiles;
};

struct BlockedPlan {
    int n;     // matrix order
    int np;    // padded order (multiple of 128), identity padding
    int ld;    // row stride of the working copies in floats (np + 64)
    int w;     // widest sub-panel allowed (what the caller asked for; 16 by default)
    int bw;    // outer block width
    int nthreads_panel;
    int rpt;   // rows per thread in the panel kernel when it holds all np rows
    // Sub-panel width of every outer block.  The panel kernel keeps (rows at or below the block) x width
    // floats in registers, so the first blocks of a large matrix use narrow sub-panels and the width grows
    // as the elimination retires rows (16384 rows: 4, 8192: 8, 4096 and fewer: 16).
    int nblk;
    unsigned char wblk[128];
    // more than 4096 candidate rows: the panel is shared by up to 4 workgroups instead of narrowing the
    // sub-panels (small batches only: all of a panel's workgroups must be resident at the same time)
    int multi_panel;
};

SweepPlan make_sweep_plan(int n);
BlockedPlan make_blocked_plan(int n, int w, int bw, int batch);
bool blocked_supported(int n);  // the register-resident panel holds at most 16384 (padded) rows

size_t sweep_workspace_bytes(const SweepPlan &p, int batch, size_t elem_bytes);
size_t blocked_workspace_bytes(const BlockedPlan &p, int batch);

// Enqueue a whole inversion on `stream`.  ws: workspace of at least the size
// reported above, 256-byte aligned.
// pivoting = false: the reference's no-pivot variant (matrix_inversion_no_pivots.cpp:10), the diagonal entry is the pivot
hipError_t sweep_invert(const SweepPlan &p, const float *d_a, float *d_inv, int batch, int *d_status, void *ws,
                        hipStream_t stream, Profiler *prof, bool pivoting = true);
// the fp64 twin (matrix_inversion_FP64 of the reference): same launches on doubles
hipError_t sweep_invert_f64(const SweepPlan &p, const double *d_a, double *d_inv, int batch, int *d_status, void *ws,
                            hipStream_t stream, Profiler *prof, bool pivoting = true);
// fp64 blocked path (mi32_blocked64.hip): windowed sweep steps + rank-bw updates on the fp64 matrix cores
struct Blocked64Plan {
    int n, np, ld;  // matrix order, padded order (multiple of 64, identity padding), row stride in doubles
    int bw;         // outer block width (multiple of 64, <= 256)
    int tr;         // rows per workgroup of the step kernel (16 or 32)
    int row_tiles;  // workgroups per step launch = arg-max records per column
};
Blocked64Plan make_blocked64_plan(int n, int bw);
size_t blocked64_workspace_bytes(const Blocked64Plan &p, int batch);
hipError_t blocked64_invert(const Blocked64Plan &p, const double *d_a, double *d_inv, int batch, int *d_status, void *ws,
                            hipStream_t stream, Profiler *prof);

// streams/events a blocked inversion is enqueued with: `aux` (may be null) carries the look-ahead half
// of each rank-bw update; events[0 .. n/2) mark "second-stream work done", events[n/2 .. n) "panel phase done"
struct BlockedExec {
    hipStream_t stream = nullptr;
    hipStream_t aux = nullptr;
    hipEvent_t *events = nullptr;
    int n_events = 0;
    int aux_workgroups = 0;  // grid of the persistent look-ahead kernel: CUs minus the ones kept free
    bool aux_exclusive = false;  // its workgroups take a whole CU's LDS: nothing of the main stream shares their CUs
    Profiler *prof = nullptr;
    bool pivoting = true;  // false: the reference's no-pivot variant (the diagonal entry is every step's pivot)
};
hipError_t blocked_invert(const BlockedPlan &p, const float *d_a, float *d_inv, int batch, int *d_status, void *ws,
                          const BlockedExec &ex);
hipError_t residual_launch(const float *d_a, const float *d_x, int n, int batch, double *d_out, void *ws,
                           hipStream_t stream);
size_t residual_workspace_bytes(int n, int batch);
hipError_t frobenius_launch_f64(const double *d_a, const double *d_b, int n, double *d_out, void *ws, hipStream_t stream);

}  // namespace mi32
