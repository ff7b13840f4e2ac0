// Diagnostic (not part of the product): fp32 VALU issue rates on this device -- v_fma_f32 against v_pk_fma_f32
// (two fp32 lanes per instruction) in the shape the panel kernel uses them: 1024-thread workgroups, one per CU,
// 64 independent accumulators per lane, a broadcast multiplier.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_peak.hip -o tools/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2v __attribute__((ext_vector_type(2)));

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void valu_loop(float *out, int iters, float seed)
{
    float a[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) a[i] = seed * (float)(i + threadIdx.x);
    float p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i] = seed + (float)i * 1e-3f;
    float f[4] = {seed, -seed, seed * 0.5f, seed * 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (MODE == 0) {
#pragma unroll
                for (int c = 0; c < 16; ++c) a[k * 16 + c] = __builtin_fmaf(-f[k], p[c], a[k * 16 + c]);
            } else {
                const f2v nf = {-f[k], -f[k]};
#pragma unroll
                for (int c = 0; c < 16; c += 2) {
                    f2v acc = {a[k * 16 + c], a[k * 16 + c + 1]};
                    const f2v pp = {p[c], p[c + 1]};
                    acc = __builtin_elementwise_fma(nf, pp, acc);
                    a[k * 16 + c] = acc[0];
                    a[k * 16 + c + 1] = acc[1];
                }
            }
        }
        asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) r += a[i];
    out[blockIdx.x * NT + threadIdx.x] = r;
}

template <int MODE, int NT>
static void run(const char *name, int blocks)
{
    float *out;
    hipMalloc(&out, sizeof(float) * blocks * NT);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((valu_loop<MODE, NT>), dim3(blocks), dim3(NT), 0, 0, out, 100, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((valu_loop<MODE, NT>), dim3(blocks), dim3(NT), 0, 0, out, iters, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double fma = (double)blocks * NT * 64.0 * iters;
    printf("%-28s NT=%4d blocks=%4d: %.3f ms  %.1f TFLOP/s  (%.2f fma/clk/CU at 2.4 GHz)\n", name, NT, blocks, ms,
           2.0 * fma / ms * 1e-9, fma / (ms * 1e-3) / 2.4e9 / (blocks < 256 ? blocks : 256));
    hipFree(out);
}

int main()
{
    run<0, 1024>("v_fma_f32", 256);
    run<1, 1024>("v_pk_fma_f32", 256);
    run<0, 256>("v_fma_f32", 256);
    run<1, 256>("v_pk_fma_f32", 256);
    run<0, 256>("v_fma_f32", 1024);
    run<1, 256>("v_pk_fma_f32", 1024);
    return 0;
}
