/*
 * gj_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see gj_oracle.h).
 *
 * Plain scalar C restatement of the reference's fp32 Gauss-Jordan inversion,
 *   R = /root/reference/Matlab/mat_inv_32/mat_inv_32/mat_inv_32.cpp
 * Every function cites the R lines it follows.  No code is copied: the
 * reference is OpenCL C inside C++ raw strings driven by cl.hpp; this file
 * is sequential C over the same data layout ([A|I] row-major N x 2N panel,
 * two ping-pong copies), written from the step semantics.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no silent fusing, the
 * fused/unfused choice is explicit via fmaf()).
 */
#include "gj_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- helpers ---------------------------------------------------------- */

/* |x| with NaN demoted below every real candidate, so that a NaN can never
 * be chosen as the arg-max (the reference's `fabs(a) > fabs(b)` is false for
 * NaN too, R:90,123). */
static inline float cand_abs(float x)
{
    float v = fabsf(x);
    return (v == v) ? v : -1.0f;
}

/* Boundary rule (README.md:54 "in case of invalid matrix an empty vector is returned"; the experiment twin's
 * exact-identity check, matrix_inversion_FP32.cpp:814-835, rejects every result that went through a zero or
 * non-finite pivot): status GJO_SINGULAR when a pivot is zero, NaN or infinite, or when the INPUT holds a
 * non-finite entry (a NaN never wins the search, R:90,123, but it poisons the result). */
static int oracle_threads(void) __attribute__((unused));
static inline int bad_pivot(double piv) { return piv == 0.0 || piv != piv || piv - piv != 0.0; }
static int input_status_f32(const float *in, int n)
{
    for (size_t i = 0; i < (size_t)n * n; ++i)
        if (in[i] - in[i] != 0.0f) return GJO_SINGULAR; /* NaN or +-inf */
    return GJO_OK;
}
static int input_status_f64(const double *in, int n)
{
    for (size_t i = 0; i < (size_t)n * n; ++i)
        if (in[i] - in[i] != 0.0) return GJO_SINGULAR;
    return GJO_OK;
}

static inline float elim(float cij, float cir, float crj, int arith_mode)
{
    /* R:34-37  Cij = Cij - (Cir * Crj) */
    if (arith_mode == GJO_ARITH_FMA)
        return fmaf(-cir, crj, cij);
    {
        volatile float prod = cir * crj; /* volatile: forbid re-fusing */
        return cij - prod;
    }
}

/* ---- R:177-192 makeAugmentedMatrix ------------------------------------ */
static void make_augmented(float *aug, const float *in, int n)
{
    const size_t w = (size_t)2 * n;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < 2 * n; ++j)
            aug[i * w + j] = (j < n) ? in[(size_t)i * n + j] : ((j - n) == i ? 1.0f : 0.0f);
}

/* ---- R:195-203 getInvertedMatrix --------------------------------------- */
static void get_inverted(const float *aug, float *out, int n)
{
    const size_t w = (size_t)2 * n;
    for (int i = 0; i < n; ++i)
        for (int j = n; j < 2 * n; ++j)
            out[(size_t)i * n + (j - n)] = aug[i * w + j];
}

/* ---- pivot search, intended semantics --------------------------------- */
/* arg-max_{i>=r} |m[i][r]|, first maximum wins (strict '>' scanned in
 * ascending row order, as finalMaxPivotKernel scans its partials, R:121-127).
 * Deviation, documented: an all-zero/NaN column yields p = r (the reference
 * would yield "row 0, pivot 0" from its (0,0) initial record, R:120). */
static int max_pivot_true(const float *m, size_t ld, int n, int r)
{
    int p = r;
    float best = cand_abs(m[(size_t)r * ld + r]);
    for (int i = r + 1; i < n; ++i) {
        float v = cand_abs(m[(size_t)i * ld + r]);
        if (v > best) {
            best = v;
            p = i;
        }
    }
    return p;
}

/* ---- R:61-106 maxPivotKernel + R:112-132 finalMaxPivotKernel, as written -- */
/* Lock-step emulation of one 256-wide work-group (barriers separate the tree
 * levels, and inside a level writes go to [0,i) while reads come from
 * [i,2i) and the thread's own slot, so sequential evaluation is exact). */
static void ref_max_pivot_group(const float *m, int size, int n, int r, int wg, float *ox, float *oy)
{
    float lx[257], ly[257];
    memset(lx, 0, sizeof lx);
    memset(ly, 0, sizeof ly);
    int nthreads = n - wg * 256; /* non-uniform last group when N % 256 != 0 (R:328) */
    if (nthreads > 256) nthreads = 256;

    for (int lid = 0; lid < nthreads; ++lid) { /* R:70 */
        int gid = wg * 256 + lid;
        lx[lid] = m[(size_t)gid * size + r];
        ly[lid] = (float)gid;
    }
    if (r <= wg * 256 + 255) { /* R:73 */
        int loop_limit = 256, lim;
        if ((size / 2) < 256) /* R:74-77 */
            loop_limit = size / 2;
        else if (wg == (int)floorf((float)(size / 512)))
            loop_limit = (size / 2) % 256;
        if (r >= wg * 256) /* R:79-82 */
            lim = loop_limit - (r % 256);
        else
            lim = loop_limit;
        if (lim % 2 != 0) { /* R:84-87 */
            lx[loop_limit] = 0.0f;
            ly[loop_limit] = 0.0f;
            lim++;
        }
        for (int i = lim >> 1; i > 0; i >>= 1) { /* R:90-98 */
            for (int lid = 0; lid < nthreads; ++lid) {
                int gid = wg * 256 + lid;
                if (lid < i && fabsf(lx[lid + i]) > fabsf(lx[lid]) && gid >= r) {
                    lx[lid] = lx[lid + i];
                    ly[lid] = ly[lid + i];
                }
            }
            if (i % 2 != 0 && i != 1) i++;
        }
        if (r >= wg * 256) { /* R:99-102 */
            *ox = lx[r % 256];
            *oy = ly[r % 256];
        } else {
            *ox = lx[0];
            *oy = ly[0];
        }
    } else { /* R:103-105 */
        *ox = 0.0f;
        *oy = 0.0f;
    }
}

static int max_pivot_reference_defect(const float *m, int n, int r, float *pivot_value)
{
    const int size = 2 * n;
    const int groups = (n % 256 == 0) ? n / 256 : n / 256 + 1; /* R:257-261 */
    float mx = 0.0f, my = 0.0f;                                /* R:120 */
    for (int g = 0; g < groups; ++g) {
        float x, y;
        ref_max_pivot_group(m, size, n, r, g, &x, &y);
        if (fabsf(x) > fabsf(mx)) { /* R:122-125 */
            mx = x;
            my = y;
        }
    }
    *pivot_value = mx;
    return (int)my; /* R:163 maxRow = (int)(pivot[0].y) */
}

/* ---- R:206-395 matrix_inv_32: guards + host loop ----------------------- */
static int shape_ok(size_t in_len, int n)
{
    if (n <= 0) return 0;                     /* R:206-208 */
    if ((int)(in_len / (size_t)n) != n) return 0; /* R:211-214 (integer division) */
    return 1;
}

int gjo_matrix_inv_32(const float *in, size_t in_len, int n, float *out, int pivot_mode,
                      int arith_mode, int *pivots, float *aug_out)
{
    if (!shape_ok(in_len, n)) return GJO_BAD_SHAPE;
    const size_t w = (size_t)2 * n;
    float *buf0 = (float *)malloc(sizeof(float) * w * n);
    float *buf1 = (float *)malloc(sizeof(float) * w * n);
    if (!buf0 || !buf1) {
        free(buf0);
        free(buf1);
        return GJO_BAD_SHAPE;
    }
    int status = input_status_f32(in, n);
    make_augmented(buf0, in, n); /* R:292-297 */

    for (int r = 0; r < n; ++r) { /* R:317 */
        float *src = (r % 2 == 0) ? buf0 : buf1; /* R:318,353-360 ping-pong */
        float *dst = (r % 2 == 0) ? buf1 : buf0;

        /* maxPivot + finalMaxPivot, R:322-331.  pivot value is read BEFORE the swap. */
        int p;
        float piv;
        if (pivot_mode == GJO_PIVOT_REFERENCE_DEFECT) {
            p = max_pivot_reference_defect(src, n, r, &piv);
        } else {
            p = max_pivot_true(src, w, n, r);
            piv = src[(size_t)p * w + r];
        }
        if (pivots) pivots[r] = p;
        if (bad_pivot((double)piv)) status = GJO_SINGULAR;

        /* pivotElementsKernel, R:154-173: swap rows r <-> p over all 2N columns */
        if (p != r) {
            for (size_t j = 0; j < w; ++j) {
                float t = src[(size_t)r * w + j];
                src[(size_t)r * w + j] = src[(size_t)p * w + j];
                src[(size_t)p * w + j] = t;
            }
        }
        /* fixRowKernel, R:138-150: true IEEE division of row r by the pivot */
        for (size_t j = 0; j < w; ++j) src[(size_t)r * w + j] = src[(size_t)r * w + j] / piv;

        /* fixColumnKernel, R:13-57: out-of-place elimination of column r */
        const float *rowr = src + (size_t)r * w;
        for (int i = 0; i < n; ++i) {
            const float *si = src + (size_t)i * w;
            float *di = dst + (size_t)i * w;
            const float cir = si[r];
            if (cir != 0.0f && i != r) { /* R:28 */
                for (size_t j = 0; j < w; ++j) di[j] = elim(si[j], cir, rowr[j], arith_mode);
            } else {
                memcpy(di, si, sizeof(float) * w);
            }
        }
    }
    /* R:368-372: the last-written buffer holds [I | A^-1] */
    const float *fin = ((n - 1) % 2 == 0) ? buf1 : buf0;
    get_inverted(fin, out, n);
    if (aug_out) memcpy(aug_out, fin, sizeof(float) * w * n);
    free(buf0);
    free(buf1);
    return status;
}

/* ---- the same arithmetic, in-place N x N ------------------------------- */
/* Column r of the working matrix is, from step r on, the augmented panel's
 * right-half column that went dense at step r (column N + orig[r], orig[r] =
 * original index of the pivot row of step r).  Every skipped operation in
 * this form is one that the augmented form performs on an exact 0 or 1, so the
 * stored values are identical. */
static int inv32_inplace_impl(const float *in, size_t in_len, int n, float *out, int arith_mode, int *pivots, int pivoting)
{
    if (!shape_ok(in_len, n)) return GJO_BAD_SHAPE;
    const size_t ld = (size_t)n;
    float *m = (float *)malloc(sizeof(float) * ld * n);
    int *orig = (int *)malloc(sizeof(int) * n);
    float *rowr = (float *)malloc(sizeof(float) * n);
    if (!m || !orig || !rowr) {
        free(m);
        free(orig);
        free(rowr);
        return GJO_BAD_SHAPE;
    }
    memcpy(m, in, sizeof(float) * ld * n);
    for (int i = 0; i < n; ++i) orig[i] = i;
    int status = input_status_f32(in, n);

    for (int r = 0; r < n; ++r) {
        const int p = pivoting ? max_pivot_true(m, ld, n, r) : r; /* no-pivot variant: the diagonal entry, findCrr */
        const float piv = m[(size_t)p * ld + r];
        if (pivots) pivots[r] = p;
        if (bad_pivot((double)piv)) status = GJO_SINGULAR;
        if (p != r) {
            for (int j = 0; j < n; ++j) {
                float t = m[(size_t)r * ld + j];
                m[(size_t)r * ld + j] = m[(size_t)p * ld + j];
                m[(size_t)p * ld + j] = t;
            }
            int t = orig[r];
            orig[r] = orig[p];
            orig[p] = t;
        }
        /* normalise; the identity-column entry 1 becomes 1/piv */
        for (int j = 0; j < n; ++j) m[(size_t)r * ld + j] = m[(size_t)r * ld + j] / piv;
        m[(size_t)r * ld + r] = 1.0f / piv;
        memcpy(rowr, m + (size_t)r * ld, sizeof(float) * n);
        for (int i = 0; i < n; ++i) {
            if (i == r) continue;
            float *mi = m + (size_t)i * ld;
            const float cir = mi[r];
            mi[r] = 0.0f; /* the identity column's entry in this row */
            if (cir != 0.0f)
                for (int j = 0; j < n; ++j) mi[j] = elim(mi[j], cir, rowr[j], arith_mode);
        }
    }
    /* un-permute columns: inverse column orig[c] lives in working column c */
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < n; ++c) out[(size_t)i * n + orig[c]] = m[(size_t)i * ld + c];
    free(m);
    free(orig);
    free(rowr);
    return status;
}

/* ---- fp64 twin ----------------------------------------------------------- */
/* The reference's matrix_inversion_FP64 (matrix_inversion_FP64.cpp:13; kernels :18-206, host loop as in the
 * fp32 library) is the same five-kernel step in double; here the in-place N x N form of it with true partial
 * pivoting (largest |a|, lowest row among equals), IEEE division and one fused multiply-add per element
 * (exact-zero multipliers skipped, matrix_inversion_FP64.cpp:28). */
static int inv64_inplace_impl(const double *in, size_t in_len, int n, double *out, int *pivots, int pivoting)
{
    if (!shape_ok(in_len, n)) return GJO_BAD_SHAPE;
    const size_t ld = (size_t)n;
    double *m = (double *)malloc(sizeof(double) * ld * n);
    int *orig = (int *)malloc(sizeof(int) * n);
    double *rowr = (double *)malloc(sizeof(double) * n);
    if (!m || !orig || !rowr) {
        free(m); free(orig); free(rowr);
        return GJO_BAD_SHAPE;
    }
    memcpy(m, in, sizeof(double) * ld * n);
    for (int i = 0; i < n; ++i) orig[i] = i;
    int status = input_status_f64(in, n);
    for (int r = 0; r < n; ++r) {
        int p = r;
        double best = -1.0;
        for (int i = r; i < (pivoting ? n : r + 1); ++i) { /* no-pivot variant: the diagonal entry only */
            const double v = fabs(m[(size_t)i * ld + r]);
            if (v > best) { best = v; p = i; } /* NaN never wins, the first maximum is kept */
        }
        const double piv = m[(size_t)p * ld + r];
        if (pivots) pivots[r] = p;
        if (bad_pivot((double)piv)) status = GJO_SINGULAR;
        if (p != r) {
            for (int j = 0; j < n; ++j) {
                double t = m[(size_t)r * ld + j];
                m[(size_t)r * ld + j] = m[(size_t)p * ld + j];
                m[(size_t)p * ld + j] = t;
            }
            int t = orig[r]; orig[r] = orig[p]; orig[p] = t;
        }
        for (int j = 0; j < n; ++j) m[(size_t)r * ld + j] = m[(size_t)r * ld + j] / piv;
        m[(size_t)r * ld + r] = 1.0 / piv;
        memcpy(rowr, m + (size_t)r * ld, sizeof(double) * n);
        for (int i = 0; i < n; ++i) {
            if (i == r) continue;
            double *mi = m + (size_t)i * ld;
            const double cir = mi[r];
            mi[r] = 0.0;
            if (cir != 0.0)
                for (int j = 0; j < n; ++j) mi[j] = fma(-cir, rowr[j], mi[j]);
        }
    }
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < n; ++c) out[(size_t)i * n + orig[c]] = m[(size_t)i * ld + c];
    free(m); free(orig); free(rowr);
    return status;
}

int gjo_matrix_inv_32_inplace(const float *in, size_t in_len, int n, float *out, int arith_mode, int *pivots)
{
    return inv32_inplace_impl(in, in_len, n, out, arith_mode, pivots, 1);
}
int gjo_matrix_inv_64_inplace(const double *in, size_t in_len, int n, double *out, int *pivots)
{
    return inv64_inplace_impl(in, in_len, n, out, pivots, 1);
}
/* The reference's no-pivot variant, matrix_inversion_no_pivots.cpp:10 (kernels :13-70, loop :482-560): per step
 * findCrr (the diagonal entry, no search, no swap), fixRow (IEEE division), copyCirColumn + fixColumn (skip zero
 * multipliers, :29).  Same in-place form; a zero / non-finite diagonal entry -> GJO_SINGULAR (the reference
 * returns {} when the reduced left half is not exactly I, :670). */
int gjo_matrix_inv_64_nopivot(const double *in, size_t in_len, int n, double *out)
{
    return inv64_inplace_impl(in, in_len, n, out, NULL, 0);
}
int gjo_matrix_inv_32_nopivot(const float *in, size_t in_len, int n, float *out, int arith_mode)
{
    return inv32_inplace_impl(in, in_len, n, out, arith_mode, NULL, 0);
}

/* ---- blocked restatement (CPU mirror of the HIP blocked path) ---------- */
/* In-place Gauss-Jordan on column blocks of width w: the N x w panel is
 * reduced with the unblocked steps above (pivot search over the whole column
 * height below the diagonal), its row swaps are applied to every other
 * column, the block's pivot rows R = M[K, J] are snapshotted, and all other
 * columns J receive one delayed rank-w update
 *     M[i, J] = (i in K ? 0 : M[i, J]) + G[i, :] * R,
 * G = the transformed panel.  Accumulation is a k-ascending fmaf chain
 * starting from the old value, which is what v_mfma_f32_32x32x2_f32 computes. */
int gjo_matrix_inv_32_blocked(const float *in, size_t in_len, int n, float *out, int w, int *pivots)
{
    if (!shape_ok(in_len, n)) return GJO_BAD_SHAPE;
    if (w <= 0) w = 16;
    const size_t ld = (size_t)n;
    float *m = (float *)malloc(sizeof(float) * ld * n);
    int *orig = (int *)malloc(sizeof(int) * n);
    float *rs = (float *)malloc(sizeof(float) * (size_t)w * n);
    float *prn = (float *)malloc(sizeof(float) * w);
    if (!m || !orig || !rs || !prn) {
        free(m); free(orig); free(rs); free(prn);
        return GJO_BAD_SHAPE;
    }
    memcpy(m, in, sizeof(float) * ld * n);
    for (int i = 0; i < n; ++i) orig[i] = i;
    int status = input_status_f32(in, n);

    for (int c0 = 0; c0 < n; c0 += w) {
        const int kw = (c0 + w <= n) ? w : n - c0;
        /* panel: unblocked steps restricted to columns [c0, c0+kw) */
        for (int s = 0; s < kw; ++s) {
            const int r = c0 + s;
            const int p = max_pivot_true(m, ld, n, r);
            const float piv = m[(size_t)p * ld + r];
            if (pivots) pivots[r] = p;
            if (bad_pivot((double)piv)) status = GJO_SINGULAR;
            if (p != r) { /* swap across ALL columns right away */
                for (int j = 0; j < n; ++j) {
                    float t = m[(size_t)r * ld + j];
                    m[(size_t)r * ld + j] = m[(size_t)p * ld + j];
                    m[(size_t)p * ld + j] = t;
                }
                int t = orig[r]; orig[r] = orig[p]; orig[p] = t;
            }
            float *mr = m + (size_t)r * ld + c0;
            for (int c = 0; c < kw; ++c) prn[c] = mr[c] / piv;
            prn[s] = 1.0f / piv;
            for (int c = 0; c < kw; ++c) mr[c] = prn[c];
            for (int i = 0; i < n; ++i) {
                if (i == r) continue;
                float *mi = m + (size_t)i * ld + c0;
                const float f = mi[s];
                mi[s] = 0.0f;
                for (int c = 0; c < kw; ++c) mi[c] = fmaf(-f, prn[c], mi[c]);
            }
        }
        /* snapshot the block's pivot rows over the other columns */
        for (int k = 0; k < kw; ++k) memcpy(rs + (size_t)k * n, m + (size_t)(c0 + k) * ld, sizeof(float) * n);
        /* delayed rank-kw update of every column outside the panel */
        for (int i = 0; i < n; ++i) {
            float *mi = m + (size_t)i * ld;
            const float *g = mi + c0;
            const int in_block = (i >= c0 && i < c0 + kw);
            for (int j = 0; j < n; ++j) {
                if (j >= c0 && j < c0 + kw) continue;
                float acc = in_block ? 0.0f : mi[j];
                for (int k = 0; k < kw; ++k) acc = fmaf(g[k], rs[(size_t)k * n + j], acc);
                mi[j] = acc;
            }
        }
    }
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < n; ++c) out[(size_t)i * n + orig[c]] = m[(size_t)i * ld + c];
    free(m); free(orig); free(rs); free(prn);
    return status;
}

/* ---- fp64 blocked restatement: exact CPU mirror of mi32_blocked64.hip ---------- */
/* Outer blocks of bw pivot columns.  Inside a block the unblocked fp64 steps above (partial pivoting over the
 * whole column height, IEEE division, one fma per element, zero multipliers skipped) run on the block's columns
 * only; the row swaps are applied to every column right away (the HIP path applies them to the other columns
 * lazily, through a row map: same values); after the block every other column receives one delayed rank-bw update
 *     M[i][j] = (i in K ? 0 : M[i][j]) + sum_k G[i][k] * R[k][j],   R = the block's pivot rows as they stood,
 * one k-ascending fma chain per element STARTING FROM THE OLD VALUE -- what a chain of v_mfma_f64_16x16x4_f64 with
 * the old value as its C operand computes. */
static void rank_update_row_f64(double *mi, const double *g, const double *rs, size_t n, int kw, int ja, int jb, int in_block)
{
    int j = ja;
#if defined(__AVX2__) && defined(__FMA__)
    for (; j + 32 <= jb; j += 32) {
        __m256d acc[8];
        for (int v = 0; v < 8; ++v) acc[v] = in_block ? _mm256_setzero_pd() : _mm256_loadu_pd(mi + j + 4 * v);
        for (int k = 0; k < kw; ++k) {
            const __m256d b = _mm256_set1_pd(g[k]);
            const double *r = rs + (size_t)k * n + j;
            for (int v = 0; v < 8; ++v) acc[v] = _mm256_fmadd_pd(b, _mm256_loadu_pd(r + 4 * v), acc[v]);
        }
        for (int v = 0; v < 8; ++v) _mm256_storeu_pd(mi + j + 4 * v, acc[v]);
    }
#endif
    for (; j < jb; ++j) {
        double acc = in_block ? 0.0 : mi[j];
        for (int k = 0; k < kw; ++k) acc = fma(g[k], rs[(size_t)k * n + j], acc);
        mi[j] = acc;
    }
}

int gjo_matrix_inv_64_blocked(const double *in, size_t in_len, int n, double *out, int bw, int *pivots)
{
    if (!shape_ok(in_len, n)) return GJO_BAD_SHAPE;
    if (bw <= 0) return GJO_BAD_SHAPE;
    const size_t ld = (size_t)n;
    double *m = (double *)malloc(sizeof(double) * ld * n);
    int *orig = (int *)malloc(sizeof(int) * n);
    double *rs = (double *)malloc(sizeof(double) * (size_t)bw * n);
    double *rowr = (double *)malloc(sizeof(double) * bw);
    if (!m || !orig || !rs || !rowr) {
        free(m); free(orig); free(rs); free(rowr);
        return GJO_BAD_SHAPE;
    }
    memcpy(m, in, sizeof(double) * ld * n);
    for (int i = 0; i < n; ++i) orig[i] = i;
    int status = input_status_f64(in, n);
    for (int c0 = 0; c0 < n; c0 += bw) {
        const int kw = (c0 + bw <= n) ? bw : n - c0;
        for (int s = 0; s < kw; ++s) { /* the steps of inv64_inplace_impl on the columns [c0, c0 + kw) */
            const int r = c0 + s;
            int p = r;
            double best = -1.0;
            for (int i = r; i < n; ++i) {
                const double v = fabs(m[(size_t)i * ld + r]);
                if (v > best) { best = v; p = i; }
            }
            const double piv = m[(size_t)p * ld + r];
            if (pivots) pivots[r] = p;
            if (bad_pivot(piv)) status = GJO_SINGULAR;
            if (p != r) {
                for (int j = 0; j < n; ++j) {
                    double t = m[(size_t)r * ld + j];
                    m[(size_t)r * ld + j] = m[(size_t)p * ld + j];
                    m[(size_t)p * ld + j] = t;
                }
                int t = orig[r]; orig[r] = orig[p]; orig[p] = t;
            }
            double *mr = m + (size_t)r * ld + c0;
            for (int c = 0; c < kw; ++c) mr[c] = mr[c] / piv;
            mr[s] = 1.0 / piv;
            memcpy(rowr, mr, sizeof(double) * kw);
            for (int i = 0; i < n; ++i) {
                if (i == r) continue;
                double *mi = m + (size_t)i * ld + c0;
                const double cir = mi[s];
                mi[s] = 0.0;
                if (cir != 0.0)
                    for (int c = 0; c < kw; ++c) mi[c] = fma(-cir, rowr[c], mi[c]);
            }
        }
        if (kw < n) {
            for (int k = 0; k < kw; ++k) memcpy(rs + (size_t)k * n, m + (size_t)(c0 + k) * ld, sizeof(double) * n);
#ifdef _OPENMP
            const int nthreads = ((double)n * kw * n > 4e7) ? oracle_threads() : 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
            for (int i = 0; i < n; ++i) {
                double *mi = m + (size_t)i * ld;
                const int in_block = (i >= c0 && i < c0 + kw);
                if (c0 > 0) rank_update_row_f64(mi, mi + c0, rs, (size_t)n, kw, 0, c0, in_block);
                if (c0 + kw < n) rank_update_row_f64(mi, mi + c0, rs, (size_t)n, kw, c0 + kw, n, in_block);
            }
        }
    }
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < n; ++c) out[(size_t)i * n + orig[c]] = m[(size_t)i * ld + c];
    free(m); free(orig); free(rs); free(rowr);
    return status;
}

/* ---- two-level blocked restatement: exact CPU mirror of mi32_blocked.hip ----- */
/* Outer blocks of bw pivot columns; inside a block, sub-panels of w columns are
 * reduced with the unblocked steps and followed by a rank-w update of the block's
 * other columns; after the block, one rank-bw update of every column outside it.
 * Row swaps are applied to all columns immediately (the HIP path applies them
 * lazily through row maps, which moves the same values).  Every accumulation is
 * the k-ascending fmaf chain of v_mfma_f32_32x32x2_f32, so the HIP blocked path
 * with the same (w, bw) reproduces these values bit for bit. */
/* add_last = 0: the fmaf chain starts from the old value (in-block updates);
 * add_last = 1: the chain starts from zero and the old value is added at the end (rank-bw updates:
 *               mi32_blocked.hip keeps only the accumulators live across its k-loop that way). */
/* One row of a rank-kw update over the columns [ja, jb) (none of them in the block).  Every output element
 * is its own k-ascending fmaf chain, so the order in which elements are visited -- 64 at a time with AVX2
 * FMA lanes here, one at a time in the scalar tail and in the portable build -- does not change a bit
 * (tests/test_oracle.py compares the two builds). */
static void rank_update_row(float *mi, const float *g, const float *rs, size_t n, int kw, int ja, int jb,
                            int in_block, int add_last)
{
    int j = ja;
#if defined(__AVX2__) && defined(__FMA__)
    for (; j + 64 <= jb; j += 64) {
        __m256 old[8], acc[8];
        for (int v = 0; v < 8; ++v) {
            old[v] = in_block ? _mm256_setzero_ps() : _mm256_loadu_ps(mi + j + 8 * v);
            acc[v] = add_last ? _mm256_setzero_ps() : old[v];
        }
        for (int k = 0; k < kw; ++k) {
            const __m256 b = _mm256_set1_ps(g[k]);
            const float *r = rs + (size_t)k * n + j;
            for (int v = 0; v < 8; ++v) acc[v] = _mm256_fmadd_ps(b, _mm256_loadu_ps(r + 8 * v), acc[v]);
        }
        for (int v = 0; v < 8; ++v)
            _mm256_storeu_ps(mi + j + 8 * v, add_last ? _mm256_add_ps(acc[v], old[v]) : acc[v]);
    }
    for (; j + 8 <= jb; j += 8) {
        const __m256 old = in_block ? _mm256_setzero_ps() : _mm256_loadu_ps(mi + j);
        __m256 acc = add_last ? _mm256_setzero_ps() : old;
        for (int k = 0; k < kw; ++k)
            acc = _mm256_fmadd_ps(_mm256_set1_ps(g[k]), _mm256_loadu_ps(rs + (size_t)k * n + j), acc);
        _mm256_storeu_ps(mi + j, add_last ? _mm256_add_ps(acc, old) : acc);
    }
#endif
    for (; j < jb; ++j) {
        const float old = in_block ? 0.0f : mi[j];
        float acc = add_last ? 0.0f : old;
        for (int k = 0; k < kw; ++k) acc = fmaf(g[k], rs[(size_t)k * n + j], acc);
        mi[j] = add_last ? acc + old : acc;
    }
}

static int oracle_threads(void)
{
#ifdef _OPENMP
    const char *e = getenv("GJO_THREADS");
    int t = e ? atoi(e) : 0;
    if (t <= 0) {
        t = omp_get_num_procs();
        if (t > 16) t = 16; /* a GPU box hands a 1-GPU job 16 CPUs of a 256-CPU host */
    }
    return t;
#else
    return 1;
#endif
}

static void rank_update(float *m, size_t ld, int n, int r0, int kw, int j_lo, int j_hi, float *rs, int add_last)
{
    /* rs: kw x n snapshot of rows [r0, r0+kw) */
    for (int k = 0; k < kw; ++k) memcpy(rs + (size_t)k * n, m + (size_t)(r0 + k) * ld, sizeof(float) * n);
    const int a_hi = (r0 < j_hi) ? r0 : j_hi;           /* columns left of the block:  [j_lo, a_hi) */
    const int b_lo = (r0 + kw > j_lo) ? r0 + kw : j_lo; /* columns right of the block: [b_lo, j_hi) */
    /* rows are independent (each reads the snapshot and its own G entries, writes its own columns) */
#ifdef _OPENMP
    const int nthreads = ((double)n * kw * (j_hi - j_lo) > 4e7) ? oracle_threads() : 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int i = 0; i < n; ++i) {
        float *mi = m + (size_t)i * ld;
        const int in_block = (i >= r0 && i < r0 + kw);
        if (j_lo < a_hi) rank_update_row(mi, mi + r0, rs, (size_t)n, kw, j_lo, a_hi, in_block, add_last);
        if (b_lo < j_hi) rank_update_row(mi, mi + r0, rs, (size_t)n, kw, b_lo, j_hi, in_block, add_last);
    }
}

int gjo_matrix_inv_32_blocked2w(const float *in, size_t in_len, int n, float *out, const int *w_of_block, int nblocks,
                                int bw, int *pivots)
{
    if (!shape_ok(in_len, n)) return GJO_BAD_SHAPE;
    if (!w_of_block || nblocks <= 0 || bw <= 0) return GJO_BAD_SHAPE;
    const size_t ld = (size_t)n;
    float *m = (float *)malloc(sizeof(float) * ld * n);
    int *orig = (int *)malloc(sizeof(int) * n);
    float *rs = (float *)malloc(sizeof(float) * (size_t)bw * n);
    float *prn = (float *)malloc(sizeof(float) * 64);
    if (!m || !orig || !rs || !prn) {
        free(m); free(orig); free(rs); free(prn);
        return GJO_BAD_SHAPE;
    }
    memcpy(m, in, sizeof(float) * ld * n);
    for (int i = 0; i < n; ++i) orig[i] = i;
    int status = input_status_f32(in, n);

    int blk = 0;
    for (int C0 = 0; C0 < n; C0 += bw, ++blk) {
        const int kb = (C0 + bw <= n) ? bw : n - C0;
        int w = w_of_block[blk < nblocks ? blk : nblocks - 1]; /* sub-panel width of this outer block */
        if (w <= 0) w = 16;
        if (w > 64) w = 64;
        for (int c0 = C0; c0 < C0 + kb; c0 += w) {
            const int kw = (c0 + w <= C0 + kb) ? w : C0 + kb - c0;
            for (int s = 0; s < kw; ++s) {
                const int r = c0 + s;
                const int p = max_pivot_true(m, ld, n, r);
                const float piv = m[(size_t)p * ld + r];
                if (pivots) pivots[r] = p;
                if (bad_pivot((double)piv)) status = GJO_SINGULAR;
                if (p != r) {
                    for (int j = 0; j < n; ++j) {
                        float t = m[(size_t)r * ld + j];
                        m[(size_t)r * ld + j] = m[(size_t)p * ld + j];
                        m[(size_t)p * ld + j] = t;
                    }
                    int t = orig[r]; orig[r] = orig[p]; orig[p] = t;
                }
                float *mr = m + (size_t)r * ld + c0;
                for (int c = 0; c < kw; ++c) prn[c] = mr[c] / piv;
                prn[s] = 1.0f / piv;
                for (int c = 0; c < kw; ++c) mr[c] = prn[c];
                for (int i = 0; i < n; ++i) {
                    if (i == r) continue;
                    float *mi = m + (size_t)i * ld + c0;
                    const float f = mi[s];
                    mi[s] = 0.0f;
                    for (int c = 0; c < kw; ++c) mi[c] = fmaf(-f, prn[c], mi[c]);
                }
            }
            if (kb > kw) rank_update(m, ld, n, c0, kw, C0, C0 + kb, rs, 0); /* inside the block */
        }
        if (kb < n) rank_update(m, ld, n, C0, kb, 0, n, rs, 1); /* everything outside the block */
    }
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < n; ++c) out[(size_t)i * n + orig[c]] = m[(size_t)i * ld + c];
    free(m); free(orig); free(rs); free(prn);
    return status;
}

int gjo_matrix_inv_32_blocked2(const float *in, size_t in_len, int n, float *out, int w, int bw, int *pivots)
{
    if (w <= 0) w = 16;
    if (bw < w) bw = w;
    return gjo_matrix_inv_32_blocked2w(in, in_len, n, out, &w, 1, bw, pivots);
}

/* ---- blocked restatement with the SEQUENTIAL arithmetic (round 3) --------------------------------------------
 * The same elimination as inv32_inplace_impl -- one fmaf per element and pivot step (R:28-38), one IEEE division
 * per element of a pivot row (R:149), exact-zero multipliers skipped (R:28) -- evaluated block by block:
 *   panel   : the bw pivot steps on the block's own columns, all rows; every step's MULTIPLIER column
 *             f[i] = m[i][r] (R:30, the value fixColumn reads before it overwrites the column) is kept, and the
 *             pivot itself in the pivot row's slot: mult[i][s].  Row swaps move a row's multiplier history with it.
 *   strip   : for every other column j, the bw pivot rows alone run the bw steps in order: row s is divided by its
 *             pivot (-> u[s][j], the pivot row as fixColumn sees it at step s), every other pivot row takes
 *             fmaf(-mult[k][s], u[s][j], x[k]).
 *   update  : every other row i: x = fmaf(-mult[i][s], u[s][j], x) for s ascending.
 * Every element goes through exactly the operations the step-by-step loop applies to it, in the same order, so the
 * result is BIT-IDENTICAL to gjo_matrix_inv_32_inplace for every block width (tests/test_oracle.py) -- this is the
 * operation order of the HIP blocked path from round 3 on, and a cache-friendly way of computing the reference-order
 * result at the BASELINE sizes in seconds. */
int gjo_matrix_inv_32_blocked_exact(const float *in, size_t in_len, int n, float *out, int bw, int *pivots)
{
    if (!shape_ok(in_len, n)) return GJO_BAD_SHAPE;
    if (bw <= 0) bw = 64;
    if (bw > n) bw = n;
    const size_t ld = (size_t)n;
    float *m = (float *)malloc(sizeof(float) * ld * n);
    int *orig = (int *)malloc(sizeof(int) * n);
    float *mult = (float *)malloc(sizeof(float) * (size_t)n * bw);
    float *us = (float *)malloc(sizeof(float) * (size_t)bw * n);
    float *prn = (float *)malloc(sizeof(float) * bw);
    if (!m || !orig || !mult || !us || !prn) {
        free(m); free(orig); free(mult); free(us); free(prn);
        return GJO_BAD_SHAPE;
    }
    memcpy(m, in, sizeof(float) * ld * n);
    for (int i = 0; i < n; ++i) orig[i] = i;
    int status = input_status_f32(in, n);

    for (int c0 = 0; c0 < n; c0 += bw) {
        const int kw = (c0 + bw <= n) ? bw : n - c0;
        /* panel: the unblocked steps on the columns [c0, c0 + kw), multipliers kept */
        for (int s = 0; s < kw; ++s) {
            const int r = c0 + s;
            const int p = max_pivot_true(m, ld, n, r);
            const float piv = m[(size_t)p * ld + r];
            if (pivots) pivots[r] = p;
            if (bad_pivot((double)piv)) status = GJO_SINGULAR;
            if (p != r) { /* pivotElements, R:154-173: all columns, and the rows' multiplier histories */
                for (int j = 0; j < n; ++j) {
                    float t = m[(size_t)r * ld + j];
                    m[(size_t)r * ld + j] = m[(size_t)p * ld + j];
                    m[(size_t)p * ld + j] = t;
                }
                for (int k = 0; k < s; ++k) {
                    float t = mult[(size_t)r * bw + k];
                    mult[(size_t)r * bw + k] = mult[(size_t)p * bw + k];
                    mult[(size_t)p * bw + k] = t;
                }
                int t = orig[r]; orig[r] = orig[p]; orig[p] = t;
            }
            float *mr = m + (size_t)r * ld + c0;
            for (int c = 0; c < kw; ++c) prn[c] = mr[c] / piv;
            prn[s] = 1.0f / piv;
            for (int c = 0; c < kw; ++c) mr[c] = prn[c];
            mult[(size_t)r * bw + s] = piv;
            for (int i = 0; i < n; ++i) {
                if (i == r) continue;
                float *mi = m + (size_t)i * ld + c0;
                const float f = mi[s];
                mult[(size_t)i * bw + s] = f;
                mi[s] = 0.0f;
                if (f != 0.0f)
                    for (int c = 0; c < kw; ++c) mi[c] = fmaf(-f, prn[c], mi[c]);
            }
        }
        if (kw == n) break;
        /* strip: the kw pivot rows (positions c0 .. c0+kw-1) over every other column, 64 columns at a time */
#ifdef _OPENMP
        const int nthreads = ((double)n * kw * n > 4e7) ? oracle_threads() : 1;
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int jn = (j0 + 64 <= n) ? 64 : n - j0;
            for (int s = 0; s < kw; ++s) {
                float *xs = m + (size_t)(c0 + s) * ld + j0;
                float *u = us + (size_t)s * n + j0;
                const float piv = mult[(size_t)(c0 + s) * bw + s];
                for (int j = 0; j < jn; ++j) {
                    const int jj = j0 + j;
                    if (jj >= c0 && jj < c0 + kw) continue;
                    xs[j] = xs[j] / piv;
                    u[j] = xs[j];
                }
                for (int k = 0; k < kw; ++k) {
                    if (k == s) continue;
                    const float f = mult[(size_t)(c0 + k) * bw + s];
                    if (f == 0.0f) continue;
                    float *xk = m + (size_t)(c0 + k) * ld + j0;
                    for (int j = 0; j < jn; ++j) {
                        const int jj = j0 + j;
                        if (jj >= c0 && jj < c0 + kw) continue;
                        xk[j] = fmaf(-f, u[j], xk[j]);
                    }
                }
            }
        }
        /* update: every row outside the block's pivot rows */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
        for (int i = 0; i < n; ++i) {
            if (i >= c0 && i < c0 + kw) continue;
            float *mi = m + (size_t)i * ld;
            const float *f = mult + (size_t)i * bw;
            for (int part = 0; part < 2; ++part) {
                const int ja = part ? c0 + kw : 0, jb = part ? n : c0;
                int j = ja;
#if defined(__AVX2__) && defined(__FMA__)
                for (; j + 32 <= jb; j += 32) {
                    __m256 acc[4];
                    for (int v = 0; v < 4; ++v) acc[v] = _mm256_loadu_ps(mi + j + 8 * v);
                    for (int k = 0; k < kw; ++k) {
                        if (f[k] == 0.0f) continue;
                        const __m256 fb = _mm256_set1_ps(f[k]);
                        const float *r = us + (size_t)k * n + j;
                        for (int v = 0; v < 4; ++v) acc[v] = _mm256_fnmadd_ps(fb, _mm256_loadu_ps(r + 8 * v), acc[v]);
                    }
                    for (int v = 0; v < 4; ++v) _mm256_storeu_ps(mi + j + 8 * v, acc[v]);
                }
#endif
                for (; j < jb; ++j) {
                    float acc = mi[j];
                    for (int k = 0; k < kw; ++k)
                        if (f[k] != 0.0f) acc = fmaf(-f[k], us[(size_t)k * n + j], acc);
                    mi[j] = acc;
                }
            }
        }
    }
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < n; ++c) out[(size_t)i * n + orig[c]] = m[(size_t)i * ld + c];
    free(m); free(orig); free(mult); free(us); free(prn);
    return status;
}

/* ---- metrics ----------------------------------------------------------- */
static double residual_generic(const float *l, const float *r, int n)
{
    /* || L*R - I ||_inf with double accumulation, row by row */
    double *acc = (double *)malloc(sizeof(double) * n);
    double worst = 0.0;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) acc[j] = 0.0;
        for (int k = 0; k < n; ++k) {
            const double lik = (double)l[(size_t)i * n + k];
            const float *rk = r + (size_t)k * n;
            for (int j = 0; j < n; ++j) acc[j] += lik * (double)rk[j];
        }
        acc[i] -= 1.0;
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += fabs(acc[j]);
        if (s > worst || s != s) worst = s;
    }
    free(acc);
    return worst;
}

double gjo_residual_inf(const float *a, const float *x, int n) { return residual_generic(a, x, n); }
double gjo_residual_inf_left(const float *a, const float *x, int n) { return residual_generic(x, a, n); }

/* matrix_multiply.cpp:25-33 (C = A*B in double) and :193-200 (sqrt(N) - ||C||_F) */
double gjo_frobenius_metric(const float *a, const float *x, int n)
{
    double *acc = (double *)malloc(sizeof(double) * n);
    double somma = 0.0;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) acc[j] = 0.0;
        for (int k = 0; k < n; ++k) {
            const double aik = (double)a[(size_t)i * n + k];
            const float *xk = x + (size_t)k * n;
            for (int j = 0; j < n; ++j) acc[j] += aik * (double)xk[j];
        }
        for (int j = 0; j < n; ++j) somma += acc[j] * acc[j];
    }
    free(acc);
    return sqrt((double)n) - sqrt(somma);
}

/* matrix_inversion_FP32.cpp:814-835 (without its column-0 slip) */
int gjo_left_half_is_identity(const float *aug, int n)
{
    const size_t w = (size_t)2 * n;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const float v = aug[i * w + j];
            if (i == j ? (v != 1.0f) : (v != 0.0f)) return 0;
        }
    return 1;
}

/* MSVC CRT rand(): s = s*214013 + 2531011; return (s >> 16) & 0x7fff.
 * The sweep driver never seeds it (RAND false, main_file.cpp:18,22-25). */
int gjo_msvc_rand(unsigned int *state)
{
    *state = *state * 214013u + 2531011u;
    return (int)((*state >> 16) & 0x7fffu);
}

/* main_file.cpp:41-52: row-major fill, zero diagonal, rand()%10 elsewhere
 * (rand() is NOT drawn for diagonal entries). */
void gjo_fill_hollow_msvc(float *a, int k, unsigned int *state)
{
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j)
            a[(size_t)i * k + j] = (i == j) ? 0.0f : (float)(gjo_msvc_rand(state) % 10);
}
