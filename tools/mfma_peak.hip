// Diagnostic (not part of the product): what the fp32 matrix pipe sustains on this device -- bare
// v_mfma_f32_32x32x2_f32 loops (registers only) and with LDS operand reads in several shapes, 1..4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2v __attribute__((ext_vector_type(2)));

// MODE 0: registers only.  1: 4 ds_read_b32 per 4 MFMAs, used at once.  2: the same, prefetched one iteration ahead.
// 3: 2 ds_read_b64 per 4 MFMAs... i.e. 4 per 8 MFMAs (two k-pairs per read), prefetched.
// 4: 2 x 4 accumulator tiles (64 x 128 per wave): 6 ds_read_b32 per 8 MFMAs, prefetched.
template <int MODE>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) float s[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) s[i] = seed * (float)(i % 97) * 1e-3f;
    __syncthreads();
    f16v a0 = {}, a1 = {}, a2 = {}, a3 = {}, a4 = {}, a5 = {}, a6 = {}, a7 = {};
    float x = seed + threadIdx.x * 1e-3f, y = seed - threadIdx.x * 1e-3f;
    const float *p = s + (threadIdx.x & 63);
    const f2v *p2 = reinterpret_cast<const f2v *>(s) + (threadIdx.x & 63);
    float xa = x, ya = y, xb = y, yb = x, zb = x, wb = y;
    f2v qa = {x, y}, qb = {y, x}, qc = {x, x}, qd = {y, y};
    if (MODE == 2 || MODE == 4) { xa = p[0]; ya = p[64]; xb = p[2048]; yb = p[2112]; zb = p[4096]; wb = p[4160]; }
    if (MODE == 3) { qa = p2[0]; qb = p2[64]; qc = p2[1024]; qd = p2[1088]; }
    for (int i = 0; i < iters; ++i) {
        const int o = ((i + 1) & 15) * 128;
        if (MODE == 1) { xa = p[o]; ya = p[o + 64]; xb = p[o + 2048]; yb = p[o + 2112]; }
        float nxa = xa, nya = ya, nxb = xb, nyb = yb, nzb = zb, nwb = wb;
        f2v nqa = qa, nqb = qb, nqc = qc, nqd = qd;
        if (MODE == 2 || MODE == 4) { nxa = p[o]; nya = p[o + 64]; nxb = p[o + 2048]; nyb = p[o + 2112]; }
        if (MODE == 4) { nzb = p[o + 4096]; nwb = p[o + 4160]; }
        if (MODE == 3) { nqa = p2[o]; nqb = p2[o + 64]; nqc = p2[o + 1024]; nqd = p2[o + 1088]; }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 3) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[0], qc[0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[0], qd[0], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(qb[0], qc[0], a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(qb[0], qd[0], a3, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[1], qc[1], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[1], qd[1], a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(qb[1], qc[1], a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(qb[1], qd[1], a3, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, xb, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, yb, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(ya, xb, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(ya, yb, a3, 0, 0, 0);
            if (MODE == 4) {
                a4 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, zb, a4, 0, 0, 0);
                a5 = __builtin_amdgcn_mfma_f32_32x32x2f32(xa, wb, a5, 0, 0, 0);
                a6 = __builtin_amdgcn_mfma_f32_32x32x2f32(ya, zb, a6, 0, 0, 0);
                a7 = __builtin_amdgcn_mfma_f32_32x32x2f32(ya, wb, a7, 0, 0, 0);
            }
        }
        xa = nxa; ya = nya; xb = nxb; yb = nyb; zb = nzb; wb = nwb;
        qa = nqa; qb = nqb; qc = nqc; qd = nqd;
    }
    float r = 0.f;
    for (int k = 0; k < 16; ++k) r += a0[k] + a1[k] + a2[k] + a3[k] + a4[k] + a5[k] + a6[k] + a7[k];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
static void run(float *out, const char *name)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 10000;
    const int mf = (MODE == 3 || MODE == 4) ? 8 : 4;
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; ++wgs_per_cu) {
        if (MODE == 4 && wgs_per_cu > 3) continue;
        const int grid = 256 * wgs_per_cu;
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(mfma_loop<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 0.5f);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double flops = (double)grid * 4 * iters * mf * 4096.0;
        printf("%-44s %d waves/SIMD: %7.2f ms  %6.1f TFLOP/s\n", name, wgs_per_cu, ms, flops / ms * 1e-9);
    }
}

int main()
{
    float *out;
    hipMalloc(&out, 4096 * 256 * 4 * 8);
    run<0>(out, "registers only");
    run<1>(out, "4 ds_read_b32 / 4 MFMA, used at once");
    run<2>(out, "4 ds_read_b32 / 4 MFMA, prefetched");
    run<3>(out, "4 ds_read_b64 / 8 MFMA, prefetched");
    run<4>(out, "6 ds_read_b32 / 8 MFMA (64x128 per wave)");
    return 0;
}
