/*
 * gj_oracle.h -- CPU oracle for the fp32 Gauss-Jordan inversion hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker.  The product path
 * (gpu_matrix_inversion_amd/) never links, imports or calls it.
 *
 * What it restates: the reference's matrix_inv_32 algorithm, i.e. the host
 * loop and the seven OpenCL kernels of
 *   /root/reference/Matlab/mat_inv_32/mat_inv_32/mat_inv_32.cpp
 * (cited function by function in gj_oracle.c), as plain scalar C.
 *
 * Parity pinning: the reference's C++ path compiles here but cannot run (the
 * OpenCL platform exposes 0 devices) and ships no golden vectors, so this
 * restatement is pinned by (i) outputs of the reference's own NumPy script
 * (matrix_inv_numpy.py just_inv -> numpy.linalg.inv) captured in this
 * container into tests/golden/, and (ii) the reference's own acceptance
 * properties (exact identity in the left half, A*inv(A) ~= I).
 */
#ifndef GJ_ORACLE_H
#define GJ_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pivot search flavour */
enum {
    GJO_PIVOT_TRUE_PARTIAL = 0, /* arg-max |a[i][r]|, i >= r, first max wins: what the
                                   reference intends and what north_star names            */
    GJO_PIVOT_REFERENCE_DEFECT = 1 /* lock-step emulation of maxPivotKernel's work-group
                                   reduction exactly as written (mat_inv_32.cpp:61-106),
                                   including its shrinking-subset defect                  */
};

/* arithmetic flavour of the elimination a[i][j] - a[i][r]*a[r][j] (mat_inv_32.cpp:34-37) */
enum {
    GJO_ARITH_FMA = 0,     /* fmaf(-a[i][r], a[r][j], a[i][j]): one rounding (contracted) */
    GJO_ARITH_UNFUSED = 1  /* product rounded, then subtraction rounded                   */
};

/* status codes (shared with include/mat_inv_32_c.h) */
enum {
    GJO_OK = 0,
    GJO_BAD_SHAPE = 1,  /* reference returns an empty vector (mat_inv_32.cpp:206-215)  */
    GJO_SINGULAR = 2    /* a zero or NaN pivot was met; output holds inf/NaN garbage   */
};

/*
 * Restatement of matrix_inv_32 on the [A|I] augmented N x 2N panel with
 * out-of-place (ping-pong) elimination, exactly the reference's data flow.
 *   in      : in_len floats, row-major N x N (a tail of < N extra floats is
 *             tolerated and ignored, as the reference's integer-division guard does)
 *   out     : N*N floats, row-major inverse (untouched on GJO_BAD_SHAPE)
 *   pivots  : optional [N] chosen pivot row per step (may be NULL)
 *   aug_out : optional [N*2N] final augmented panel (left half must be exactly I)
 */
int gjo_matrix_inv_32(const float *in, size_t in_len, int n, float *out,
                      int pivot_mode, int arith_mode, int *pivots, float *aug_out);

/*
 * Same arithmetic in the in-place N x N formulation (the inverse column
 * overwrites the eliminated column; columns are un-permuted at the end).
 * With GJO_PIVOT_TRUE_PARTIAL it is bit-identical to gjo_matrix_inv_32 --
 * tests assert that -- and it is the data layout the HIP kernels use.
 */
int gjo_matrix_inv_32_inplace(const float *in, size_t in_len, int n, float *out,
                              int arith_mode, int *pivots);

/* fp64 twin of the in-place form (the reference's matrix_inversion_FP64, matrix_inversion_FP64.cpp:13), with
 * true partial pivoting and one fma per element. */
int gjo_matrix_inv_64_inplace(const double *in, size_t in_len, int n, double *out, int *pivots);

/* fp64 blocked restatement (outer blocks of bw pivot columns, one delayed rank-bw update per block whose fma
 * chains start from the old value): the CPU mirror of the HIP fp64 blocked path's operation order. */
int gjo_matrix_inv_64_blocked(const double *in, size_t in_len, int n, double *out, int bw, int *pivots);

/* The reference's no-pivot variant (matrix_inversion_no_pivots.cpp:10; headers.h:11): the pivot of step r is the
 * diagonal entry; fp64 as the reference ships it, and the same steps in fp32. */
int gjo_matrix_inv_64_nopivot(const double *in, size_t in_len, int n, double *out);
int gjo_matrix_inv_32_nopivot(const float *in, size_t in_len, int n, float *out, int arith_mode);

/* Blocked (rank-b delayed update) restatement of the same elimination: the
 * CPU mirror of the HIP blocked path's operation order (panel of width w
 * factored with partial pivoting, then one rank-w update of every other
 * column).  Equal to the unblocked result up to fp32 rounding only. */
int gjo_matrix_inv_32_blocked(const float *in, size_t in_len, int n, float *out,
                              int w, int *pivots);

/* Two-level variant with the exact block structure of the HIP blocked path
 * (sub-panels of w inside outer blocks of bw): bit-identical to it. */
int gjo_matrix_inv_32_blocked2(const float *in, size_t in_len, int n, float *out,
                               int w, int bw, int *pivots);
/* The same with one sub-panel width per outer block (w_of_block[b], the last entry repeated if there are more
 * blocks): the HIP path widens its sub-panels as the elimination retires rows (mi32_resolve_panel_widths). */
int gjo_matrix_inv_32_blocked2w(const float *in, size_t in_len, int n, float *out, const int *w_of_block,
                                int nblocks, int bw, int *pivots);

/* Blocked evaluation of the SEQUENTIAL arithmetic (multipliers kept, pivot-row strip, fmaf chains from the old
 * value): bit-identical to gjo_matrix_inv_32_inplace (GJO_ARITH_FMA) for every block width bw; the operation order
 * of the HIP blocked path from round 3 on. */
int gjo_matrix_inv_32_blocked_exact(const float *in, size_t in_len, int n, float *out, int bw, int *pivots);

/* ||A*X - I||_inf (max abs row sum), product accumulated in double. */
double gjo_residual_inf(const float *a, const float *x, int n);
/* ||X*A - I||_inf: the side the reference's Python scripts check (PY:341). */
double gjo_residual_inf_left(const float *a, const float *x, int n);
/* The reference's own metric: sqrt(N) - ||A*X||_F, double accumulate
 * (matrix_multiply.cpp:25-33,193-200). */
double gjo_frobenius_metric(const float *a, const float *x, int n);

/* The reference's exact-identity acceptance check on the final augmented
 * panel (matrix_inversion_FP32.cpp:814-835), restated without its
 * column-0 index slip: returns 1 when the left half is exactly I. */
int gjo_left_half_is_identity(const float *aug, int n);

/* MSVC rand() stream (seed 1 = the unseeded default) and the sweep driver's
 * hollow rand()%10 matrix (main_file.cpp:41-52).  state is the LCG state. */
int gjo_msvc_rand(unsigned int *state);
void gjo_fill_hollow_msvc(float *a, int k, unsigned int *state);

#ifdef __cplusplus
}
#endif
#endif
