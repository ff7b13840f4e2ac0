/*
 * mat_inv_32_c.h -- C ABI of libmat_inv_32.so (MI355X / gfx950).
 *
 * The reference exposes exactly one entry point for this path,
 *     std::vector<float> matrix_inv_32(std::vector<float>, int)
 *     (/root/reference/Matlab/mat_inv_32.h:4, body mat_inv_32.cpp:11-395),
 * a C++-ABI function that MATLAB binds through clibgen (README.md:31-52).
 * mi32_matrix_inv_32() below is its flat-pointer twin (what a ctypes / cgo /
 * JNI / MEX binding would call); include/mat_inv_32.h keeps the original C++
 * signature on top of it.  Everything else here is additive: a handle so that
 * the device context, stream and workspace outlive a call (the reference
 * rebuilds platform/context/queue/programs per call, mat_inv_32.cpp:238-290),
 * device-pointer entry points for device-resident batches, and a device-side
 * residual check (the reference's matrix_multiply.cpp verification helper).
 *
 * Plain pointers and sizes only; no C++ or torch types.  All functions
 * return an mi32_status unless stated otherwise.
 */
#ifndef MAT_INV_32_C_H
#define MAT_INV_32_C_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum mi32_status {
    MI32_OK = 0,
    MI32_BAD_SHAPE = 1,     /* reference: returns {} (mat_inv_32.cpp:206-215)                 */
    MI32_SINGULAR = 2,      /* invalid matrix: a zero / NaN / infinite pivot was met, or the input holds a
                             * non-finite entry (reference: inf/NaN out, unchecked; README.md:54 "empty vector") */
    MI32_RUNTIME_ERROR = 3  /* HIP error (reference: unreachable catch, mat_inv_32.cpp:391); as a per-matrix
                             * status: a shared panel lost a partner workgroup, that inverse is NaN-filled */
} mi32_status;

typedef enum mi32_algo {
    MI32_ALGO_AUTO = 0,
    /* One fused launch per pivot step over the whole working matrix: the
     * literal restatement of the reference's 5-kernel step (maxPivot,
     * finalMaxPivot, pivotElements, fixRow, fixColumn; mat_inv_32.cpp:317-362).
     * HBM-bound; bit-identical to the CPU oracle. */
    MI32_ALGO_SWEEP = 1,
    /* Same elimination with the column updates delayed and applied as rank-k
     * updates on the fp32 matrix cores (v_mfma_f32_32x32x2_f32); the pivot
     * search / swap / normalise / eliminate steps run on a register-resident
     * panel.  Equal to SWEEP up to fp32 rounding. */
    MI32_ALGO_BLOCKED = 2
} mi32_algo;

typedef struct mi32_context *mi32_handle_t;

/* ---- drop-in twin of matrix_inv_32 (host pointers) ----------------------- */
/* a_rowmajor: a_len floats, row-major n x n (the reference's integer-division
 * guard accepts a_len in [n*n, n*n+n); the tail is ignored).  inv_rowmajor: n*n
 * floats, written only on MI32_OK / MI32_SINGULAR.  Uses a process-wide default
 * context (device MI32_DEVICE or 0), created on first use, mutex-protected. */
int mi32_matrix_inv_32(const float *a_rowmajor, size_t a_len, int n, float *inv_rowmajor);

/* batch of independent n x n matrices, contiguous (batch x n x n); status may be
 * NULL, else receives one mi32_status per matrix.  Returns the worst status. */
int mi32_matrix_inv_32_batched(const float *a, int n, int batch, float *inv, int *status);

/* The same batch over ngpus GPUs of this node (ngpus <= 0: every visible device) -- what replaces the reference's
 * platforms[0] / devices[0] (Matlab/mat_inv_32/mat_inv_32/mat_inv_32.cpp:239-244) for the callers the reference has
 * (C++ and MATLAB: no launcher, no Python).  One context and one host thread per GPU; GPU g owns the matrices
 * [g * ceil(batch / ngpus), min(batch, (g + 1) * ceil(batch / ngpus))), copies ITS OWN shard host -> device, inverts it
 * and copies it back.  No data-path exchange between the GPUs (independent matrices); the return value is the worst
 * status, status[] (may be NULL) one word per matrix.  Every matrix's inverse is bit-identical to what
 * mi32_matrix_inv_32_batched gives for it alone.  ngpus > visible devices is MI32_BAD_SHAPE unless
 * MI32_MULTI_OVERSUBSCRIBE=1 maps logical GPU g onto device g % visible (tests, single-GPU hosts). */
int mi32_matrix_inv_32_batched_multi(const float *a, int n, int batch, float *inv, int *status, int ngpus);
/* the shard of GPU g of ngpus: [*lo, *hi) (empty for the GPUs a ragged batch leaves without work); needs no device */
int mi32_shard_range(int batch, int ngpus, int g, int *lo, int *hi);

/* ---- context ------------------------------------------------------------- */
int mi32_create(mi32_handle_t *out, int device /* HIP ordinal, <0 = current */);
int mi32_destroy(mi32_handle_t h);
/* hipStream_t on which every launch of this context is enqueued from now on; NULL is
 * HIP's default stream.  A new context starts on a non-blocking stream of its own.  The
 * caller keeps ownership of the stream it passes. */
int mi32_set_stream(mi32_handle_t h, void *hip_stream);
int mi32_set_algo(mi32_handle_t h, int algo);
/* tuning knobs of the blocked path: sub-panel width (4/8/16/32, capped by what fits in registers) and the outer
 * block width (multiple of the sub-panel width, <= 512); 0 keeps the default */
int mi32_set_blocking(mi32_handle_t h, int panel_width, int block_width);
/* look-ahead of the blocked path (second stream; on by default for single matrices of more than 4096 padded rows) */
int mi32_set_lookahead(mi32_handle_t h, int enable);
/* bytes of device workspace a call of this shape needs (excluding in/out) */
size_t mi32_workspace_bytes(int n, int batch, int algo);
/* allocate the workspace up front so that later calls never hipMalloc */
int mi32_reserve(mi32_handle_t h, int n, int batch);

/* ---- device-resident entry points ---------------------------------------- */
/* d_a, d_inv: device pointers, batch x n x n fp32 row-major, contiguous; d_a is
 * not modified, d_inv may not alias d_a.  d_status: device int[batch] (may be
 * NULL: the context then keeps the status words itself).  Asynchronous: everything is enqueued on the context's stream and the
 * call returns without synchronising.  The call shape, minus the host copies,
 * of mat_inv_32.cpp:292-376 (makeAugmented -> N pivot steps -> getInverted). */
int mi32_inv_device(mi32_handle_t h, const float *d_a, int n, int batch, float *d_inv, int *d_status);

/* ---- fp64 (the reference's matrix_inversion_FP64, matrix_inversion/headers.h:9) ---------------- */
/* Same Gauss-Jordan step sequence in double.  N < 256: the sweep path (one fused launch per pivot step over the whole
 * matrix, 16 N (N+1) bytes per step), bit-identical to the oracle's fp64 restatement.  N >= 256: blocked -- the same
 * fused steps on a window of bw columns and one rank-bw update per block on v_mfma_f64_16x16x4_f64 (mi32_blocked64.hip),
 * bit-identical to the oracle's fp64 blocked mirror.  Host-pointer twin of the C++
 * function in mat_inv_64.h, and the device-resident batched form (asynchronous on the context's stream). */
int mi32_matrix_inv_64(const double *a_rowmajor, size_t a_len, int n, double *inv_rowmajor);
/* The reference's no-pivot variant (matrix_inversion_no_pivots.cpp:10, headers.h:11): the same steps with the
 * diagonal entry as pivot, no search and no swap -- for diagonally dominant inputs.  Host-pointer twin in double
 * (as the reference ships it); mi32_set_pivoting(h, 0) selects it for the device-resident calls of a context, in
 * either precision.  fp64 runs on the sweep path; fp32 takes the blocked path from 512 rows on: without a search the
 * W pivot rows of a sub-panel are known in advance, so the "panel" is their W x W diagonal block and every other row
 * is taken through the W steps by the update tiles on the whole chip (N = 4096: 5.7 ms against 92 ms for the sweep
 * kernels), bit-identical to the step-by-step restatement.  A zero / non-finite diagonal entry -> MI32_SINGULAR. */
int mi32_matrix_inversion_no_pivots(const double *a_rowmajor, size_t a_len, int n, double *inv_rowmajor);
int mi32_set_pivoting(mi32_handle_t h, int enable);
int mi32_inv_device_f64(mi32_handle_t h, const double *d_a, int n, int batch, double *d_inv, int *d_status);
/* outer block width of the fp64 blocked path for this order (the step kernels run on a window of that many columns,
 * one rank-bw update on the fp64 matrix cores per block); 0 where the unblocked sweep is used (N < 256,
 * MI32_ALGO_SWEEP, pivoting off) */
int mi32_resolve_blocking_f64(mi32_handle_t h, int n, int *block_width);

/* Device-side verification (the reference's matrix_multiply.cpp:17-36,193-200 and
 * the residual BASELINE.json gates): per matrix, d_out[3*b+0] = ||A X - I||_inf,
 * d_out[3*b+1] = ||X A - I||_inf, d_out[3*b+2] = sqrt(N) - ||A X||_F, all
 * accumulated in fp64.  d_out: device double[3*batch].  Asynchronous. */
int mi32_residual_device(mi32_handle_t h, const float *d_a, const float *d_x, int n, int batch, double *d_out);

/* ---- per-phase timing (the reference's FP32_bench.cpp:256-443 timing slots) ---------- */
/* When enabled, every kernel launch of this context is bracketed by two HIP events
 * recorded on the launch stream.  mi32_get_profile synchronises those events and returns,
 * per kernel class, the summed milliseconds and the number of launches since the last
 * call.  Classes (MI32_KC_*): 0 init (makeAugmented), 1 sweep step, 2 panel steps,
 * 3 in-block rank-w update, 4 rank-bw update (fp32 MFMA), 5 finish (getInverted),
 * 6 multiplier transposition in front of each rank-bw update (its A operand). */
#define MI32_KC_COUNT 7
int mi32_set_profiling(mi32_handle_t h, int enable);
int mi32_get_profile(mi32_handle_t h, double *ms_per_class, long long *launches_per_class, int nclasses);

/* The reference's benchmark twin (Res FP32_bench(vector<float>, int), FP32_bench.cpp:11; C++ signature in
 * mat_inv_bench.h): one host-pointer inversion that also fills times10[10] with the reference's timing vector
 * (FP32_bench.cpp:256-443), seconds:
 *   [0] queue/context (the cached default context: ~0 after the first call)   [1] buffers: staging + workspace
 *   allocation and the H2D copy (the reference's CL_MEM_COPY_HOST_PTR)        [2] program build: 0 (one AOT code
 *   object)   [3] makeAugmented = init kernel   [4] pivot = the panel kernels (search + swap + normalise, and
 *   the elimination of the panel's own columns)   [5] fixRow: 0, it has no launch of its own   [6] fixColumn =
 *   in-block + rank-bw updates (+ panel transposes), or the fused step launches of the sweep path
 *   [7] compute (wall time of the device-resident inversion)   [8] getInverted = un-permutation kernels + D2H
 *   [9] total.  Slots 3-6 and the kernel part of 8 are HIP-event durations on the launch stream. */
int mi32_bench_32(const float *a_rowmajor, size_t a_len, int n, float *inv_rowmajor, double *times10);
/* the fp64 twins (Res FP64_bench / no_pivots_bench, headers.h:14,16): pivoting = 0 selects the no-pivot variant */
int mi32_bench_64(const double *a_rowmajor, size_t a_len, int n, double *inv_rowmajor, double *times10, int pivoting);
/* matrix_multiply of the reference (matrix_multiply.cpp:15): *errore = sqrt(N) - ||A * B||_F, N = sqrt(len), the product
 * accumulated in double on the fp64 matrix cores; MI32_BAD_SHAPE unless len is a perfect square */
int mi32_matrix_multiply_64(const double *a, const double *b, size_t len, double *errore);

/* ---- introspection -------------------------------------------------------- */
/* The two durations the reference prints per call ("Tempo Totale Impiegato",
 * "Tempo Computazione", mat_inv_32.cpp:385-386) for the last host-pointer call
 * on the default context: total (H2D + compute + D2H) and compute only. */
int mi32_last_timing(double *total_seconds, double *compute_seconds);
/* which algorithm a call of this shape would use after AUTO resolution */
int mi32_resolve_algo(mi32_handle_t h, int n, int batch);
/* widest sub-panel allowed and outer block width the blocked path would use for this shape */
int mi32_resolve_blocking(mi32_handle_t h, int n, int batch, int *panel_width, int *block_width);
/* The sub-panel width of every outer block (the panel kernel keeps rows x width floats in registers, so the
 * first blocks of a large matrix use narrower sub-panels): *nblocks receives the number of outer blocks,
 * widths[0 .. min(capacity, *nblocks)) their sub-panel widths. */
int mi32_resolve_panel_widths(mi32_handle_t h, int n, int batch, int *widths, int capacity, int *nblocks);
/* name of the dominant device kernel of that algorithm (for rocprof filtering) */
const char *mi32_dominant_kernel(int algo);
/* thread-local description of the last MI32_RUNTIME_ERROR */
const char *mi32_last_error(void);
int mi32_version(void);

#ifdef __cplusplus
}
#endif
#endif
