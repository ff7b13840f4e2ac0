// mi32_residual.hip -- device-side verification of an inverse, fp64 accumulate.
//
// Counterpart of the reference's verification helper
// /root/reference/matrix_inv_solution/matrix_inversion_solution/matrix_inversion/
// matrix_multiply.cpp (naive fp64 GEMM kernel :17-36, metric sqrt(N) - ||C||_F
// :193-200) and of the residual BASELINE.json gates: ||A X - I||_inf.
// Per matrix b:  out[3b+0] = ||A X - I||_inf,  out[3b+1] = ||X A - I||_inf,
//                out[3b+2] = sqrt(N) - ||A X||_F.
// fp32 operands are widened to fp64 in registers; products and sums are fp64, on v_mfma_f64_16x16x4_f64.
#include "mi32_internal.h"

namespace mi32 {

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
// workspace: per matrix rowsum_right[n], rowsum_left[n], sumsq (doubles)
size_t residual_workspace_bytes(int n, int batch) { return align256(((size_t)2 * n + 2) * sizeof(double) * batch); }

// C = L * R (n x n) on the fp64 matrix cores: 64 x 64 tile per workgroup, 4 waves (2 x 2), each 32 x 32 = 2 x 2 tiles
// of v_mfma_f64_16x16x4_f64 (A[i = lane & 15][k = lane >> 4], B[k][j = lane & 15], D[row = (lane >> 4) + 4 reg][col =
// lane & 15]); the fp32 operands are widened to fp64 on their way from LDS, so every product is exact and every
// element is one k-ascending fp64 fma chain.
// which = 0: right residual (L=A, R=X) also accumulates sum of squares of C.
// TIN = float: the inverse check of the fp32 path.  TIN = double: matrix_multiply of the reference (fp64 operands).
typedef double res_d4v __attribute__((ext_vector_type(4)));
template <typename TIN>
__global__ __launch_bounds__(256) void residual_tile_kernel(const TIN *__restrict__ l_all,
                                                             const TIN *__restrict__ r_all, int n,
                                                             double *__restrict__ ws, int which)
{
    __shared__ TIN s_l[16][65];  // [k][i]
    __shared__ TIN s_r[16][65];  // [k][j]
    __shared__ double s_sq[4];
    const int b = blockIdx.z;
    const TIN *L = l_all + (size_t)b * n * n;
    const TIN *R = r_all + (size_t)b * n * n;
    double *rowsum = ws + (size_t)b * (2 * (size_t)n + 2) + (which ? n : 0);
    double *sumsq = ws + (size_t)b * (2 * (size_t)n + 2) + 2 * (size_t)n;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    res_d4v acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[u][v][q] = 0.0;

    for (int k0 = 0; k0 < n; k0 += 16) {
        if ((n & 3) == 0) {  // rows are 16-byte aligned: one 16- (32-) byte load per thread and operand
            typedef TIN res_f4v __attribute__((ext_vector_type(4)));
            const int ii = tid >> 2, k4 = (tid & 3) * 4;   // L tile: 64 rows x 16 k
            res_f4v v = (res_f4v)(TIN(0));
            if (i0 + ii < n && k0 + k4 < n) v = *reinterpret_cast<const res_f4v *>(L + (size_t)(i0 + ii) * n + k0 + k4);
            s_l[k4 + 0][ii] = v[0]; s_l[k4 + 1][ii] = v[1]; s_l[k4 + 2][ii] = v[2]; s_l[k4 + 3][ii] = v[3];
            const int kk = tid >> 4, j4 = (tid & 15) * 4;   // R tile: 16 k x 64 j
            res_f4v w = (res_f4v)(TIN(0));
            if (k0 + kk < n && j0 + j4 < n) w = *reinterpret_cast<const res_f4v *>(R + (size_t)(k0 + kk) * n + j0 + j4);
            s_r[kk][j4 + 0] = w[0]; s_r[kk][j4 + 1] = w[1]; s_r[kk][j4 + 2] = w[2]; s_r[kk][j4 + 3] = w[3];
        } else {
            for (int idx = tid; idx < 64 * 16; idx += 256) {   // L tile: 64 rows x 16 k
                const int ii = idx >> 4, kk = idx & 15;
                const int gi = i0 + ii, gk = k0 + kk;
                s_l[kk][ii] = (gi < n && gk < n) ? L[(size_t)gi * n + gk] : TIN(0);
            }
            for (int idx = tid; idx < 16 * 64; idx += 256) {
                const int kk = idx >> 6, jj = idx & 63;
                const int gk = k0 + kk, gj = j0 + jj;
                s_r[kk][jj] = (gk < n && gj < n) ? R[(size_t)gk * n + gj] : TIN(0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
            double lv[2], rv[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                lv[q] = (double)s_l[kk + l4][wr * 32 + q * 16 + l15];
                rv[q] = (double)s_r[kk + l4][wc * 32 + q * 16 + l15];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v)
                    acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(lv[u], rv[v], acc[u][v], 0, 0, 0);
        }
        __syncthreads();
    }
    double sq = 0.0;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gi = i0 + wr * 32 + u * 16 + l4 + 4 * q;
            double rs = 0.0;
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                const int gj = j0 + wc * 32 + v * 16 + l15;
                if (gi < n && gj < n) {
                    const double c = acc[u][v][q];
                    sq += c * c;
                    rs += fabs(c - (gi == gj ? 1.0 : 0.0));
                }
            }
            // the 16 lanes that share l4 hold the 16 columns of this row: consecutive lanes
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
            if (l15 == 0 && gi < n) atomicAdd(&rowsum[gi], rs);
        }
    if (which == 0) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_xor(sq, off, 64);
        if ((tid & 63) == 0) s_sq[tid >> 6] = sq;
        __syncthreads();
        if (tid == 0) atomicAdd(sumsq, s_sq[0] + s_sq[1] + s_sq[2] + s_sq[3]);
    }
}

__global__ __launch_bounds__(256) void residual_finalize_kernel(const double *__restrict__ ws, int n,
                                                                 double *__restrict__ out)
{
    __shared__ double s_m[2][4];
    const int b = blockIdx.x;
    const double *base = ws + (size_t)b * (2 * (size_t)n + 2);
    const int tid = threadIdx.x;
    double m0 = 0.0, m1 = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double a = base[i], c = base[n + i];
        m0 = (a > m0 || a != a) ? a : m0;  // NaN propagates
        m1 = (c > m1 || c != c) ? c : m1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o0 = __shfl_xor(m0, off, 64), o1 = __shfl_xor(m1, off, 64);
        m0 = (o0 > m0 || o0 != o0) ? o0 : m0;
        m1 = (o1 > m1 || o1 != o1) ? o1 : m1;
    }
    if ((tid & 63) == 0) { s_m[0][tid >> 6] = m0; s_m[1][tid >> 6] = m1; }
    __syncthreads();
    if (tid == 0) {
        for (int q = 1; q < 4; ++q) {
            const double o0 = s_m[0][q], o1 = s_m[1][q];
            m0 = (o0 > m0 || o0 != o0) ? o0 : m0;
            m1 = (o1 > m1 || o1 != o1) ? o1 : m1;
        }
        out[3 * b + 0] = m0;
        out[3 * b + 1] = m1;
        out[3 * b + 2] = sqrt((double)n) - sqrt(base[2 * (size_t)n]);
    }
}

hipError_t residual_launch(const float *d_a, const float *d_x, int n, int batch, double *d_out, void *ws,
                           hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(ws, 0, ((size_t)2 * n + 2) * sizeof(double) * batch, stream);
    if (e != hipSuccess) return e;
    const dim3 grid((n + 63) / 64, (n + 63) / 64, batch);
    hipLaunchKernelGGL(residual_tile_kernel<float>, grid, dim3(256), 0, stream, d_a, d_x, n, (double *)ws, 0);
    hipLaunchKernelGGL(residual_tile_kernel<float>, grid, dim3(256), 0, stream, d_x, d_a, n, (double *)ws, 1);
    hipLaunchKernelGGL(residual_finalize_kernel, dim3(batch), dim3(256), 0, stream, (const double *)ws, n, d_out);
    return hipGetLastError();
}

// sqrt(N) - ||A B||_F for fp64 operands (matrix_multiply.cpp:17-36 the product, :193-200 the metric): one product,
// d_out[0] = the metric.  ws: residual_workspace_bytes(n, 1).
__global__ void frobenius_finalize_kernel(const double *__restrict__ ws, int n, double *__restrict__ out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = sqrt((double)n) - sqrt(ws[2 * (size_t)n]);
}
hipError_t frobenius_launch_f64(const double *d_a, const double *d_b, int n, double *d_out, void *ws, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(ws, 0, ((size_t)2 * n + 2) * sizeof(double), stream);
    if (e != hipSuccess) return e;
    const dim3 grid((n + 63) / 64, (n + 63) / 64, 1);
    hipLaunchKernelGGL(residual_tile_kernel<double>, grid, dim3(256), 0, stream, d_a, d_b, n, (double *)ws, 0);
    hipLaunchKernelGGL(frobenius_finalize_kernel, dim3(1), dim3(64), 0, stream, (const double *)ws, n, d_out);
    return hipGetLastError();
}

}  // namespace mi32
