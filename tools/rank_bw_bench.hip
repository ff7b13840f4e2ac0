// rank_bw_bench.hip -- diagnostic only (not part of the product): runs the rank-bw update kernels alone
// on synthetic operands, checks generation 2 against generation 1 bit for bit, and prints TFLOP/s.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -ffp-contract=off -I include \
//        -I gpu_matrix_inversion_amd/csrc tools/rank_bw_bench.hip -o tools/rank_bw_bench
// Run:   tools/rank_bw_bench [np] [kdim] [batch] [reps]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <numeric>
#include "../gpu_matrix_inversion_amd/csrc/mi32_blocked.hip"
#include "rank_bw_gen1.h"

using namespace mi32;

// experiment: persistent grid walking the tiles, WPC workgroups per CU, optional one-time start stagger
template <int BK, int WPC>
__global__ __launch_bounds__(256, WPC) void rb_persistent_exp_kernel(
    const float *__restrict__ src_all, float *__restrict__ dst_all, const float *__restrict__ g_all, size_t gstride,
    const float *__restrict__ gk_all, size_t gkstride, int np, int ld, size_t mstride, int c0, int kdim,
    const int *__restrict__ map_all, int copy_panel, PanelExport ex, size_t tstride, int stagger_cycles)
{
    extern __shared__ __attribute__((aligned(16))) float rb_smem[];
    const int T = np / 128;
    if (stagger_cycles > 0) {
        const int slot = (int)(blockIdx.x / 256) % WPC;   // workgroups b and b + 256 tend to share a CU
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)slot * stagger_cycles) __builtin_amdgcn_s_sleep(8);
    }
    for (int id = blockIdx.x; id < T * T; id += gridDim.x) {
        int rt, ct;
        rb_tile_of(id, T, T, rt, ct);
        rank_bw2_tile<BK>(src_all, dst_all, g_all, gstride, gk_all, gkstride, np, ld, mstride, c0, kdim, map_all,
                          copy_panel, ex, tstride, 0, 0, blockIdx.y, rt, ct, rb_smem);
        __syncthreads();
    }
}

#ifdef MI32_RB_STAMPS
__global__ void set_stamp_buffer(unsigned long long *p) { mi32::g_rb_stamps = p; }
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#ifndef RB_BK
#define RB_BK 16
#endif
#ifndef RB_WPS
#define RB_WPS 3
#endif
#ifndef RB_BN
#define RB_BN 128
#endif

int main(int argc, char **argv)
{
    const int np = argc > 1 ? atoi(argv[1]) : 4096;
    const int kdim = argc > 2 ? atoi(argv[2]) : 256;
    const int batch = argc > 3 ? atoi(argv[3]) : 1;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const int ld = np + 64;
    const int c0 = (np >= 4 * kdim) ? kdim : 0;  // a block in the middle
    const size_t mstride = (size_t)np * ld;
    const size_t gkstride = (size_t)kdim * np;
    const size_t tstride = (size_t)32 * np;
    printf("rank-bw bench: np=%d kdim=%d batch=%d c0=%d BK=%d WPS=%d BN=%d\n", np, kdim, batch, c0, RB_BK, RB_WPS, RB_BN);

    std::vector<float> h(mstride * batch);
    srand(3);
    for (auto &v : h) v = (float)(rand() & 0xffff) / 65536.f * 2.f - 1.f;
    std::vector<int> hmap((size_t)np * batch);
    for (int b = 0; b < batch; ++b) {
        int *m = hmap.data() + (size_t)b * np;
        std::iota(m, m + np, 0);
        for (int i = np - 1; i > 0; --i) std::swap(m[i], m[rand() % (i + 1)]);
    }
    float *src, *g, *d1, *d2, *gk, *pt1, *pt2;
    int *map;
    CK(hipMalloc(&src, mstride * batch * 4)); CK(hipMalloc(&g, mstride * batch * 4));
    CK(hipMalloc(&d1, mstride * batch * 4)); CK(hipMalloc(&d2, mstride * batch * 4));
    CK(hipMalloc(&gk, gkstride * batch * 4));
    CK(hipMalloc(&pt1, tstride * batch * 4)); CK(hipMalloc(&pt2, tstride * batch * 4));
    CK(hipMalloc(&map, (size_t)np * batch * 4));
    CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (auto &v : h) v = (float)(rand() & 0xffff) / 65536.f * 0.5f - 0.25f;
    CK(hipMemcpy(g, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(map, hmap.data(), hmap.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d1, 0xff, mstride * batch * 4)); CK(hipMemset(d2, 0xff, mstride * batch * 4));
    CK(hipMemset(pt1, 0, tstride * batch * 4)); CK(hipMemset(pt2, 0, tstride * batch * 4));

    const int T = np / 128;
    const int pt_col = c0 + kdim, pt_w = 16;
    const size_t lds2 = rank_bw2_lds_bytes<RB_BK, RB_BN>(kdim);
    CK(hipFuncSetAttribute((const void *)gj_rank_bw2_kernel<RB_BK, RB_WPS, RB_BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    auto run1 = [&]() {
        hipLaunchKernelGGL((gj_rank_bw_update_kernel<16, 3>), dim3(T * T, batch), dim3(256), 0, 0, src, d1, g, mstride, np, ld,
                           mstride, c0, kdim, map, 1, pt1, tstride, pt_col, pt_w, 0, 0);
    };
    auto runT = [&]() {
        hipLaunchKernelGGL(gj_panel_transpose_kernel, dim3(np / 64, kdim / 64, batch), dim3(256), 0, 0, g, mstride, np, ld, c0, gk,
                           gkstride);
    };
    CK(hipFuncSetAttribute((const void *)gj_rank_bw2_kernel<RB_BK, RB_WPS, RB_BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    auto run2 = [&]() {
        hipLaunchKernelGGL((gj_rank_bw2_kernel<RB_BK, RB_WPS, RB_BN>), dim3(T * (np / RB_BN), batch), dim3(256), lds2, 0, src, d2, g, mstride, gk,
                           gkstride, np, ld, mstride, c0, kdim, map, 1, PanelExport{pt2, 0, pt_col, pt_w, 1}, tstride, 0, 0, (const int *)nullptr);
    };
    auto run2old = [&]() {  // lane = column layout (first version of generation 2), for A/B
        hipLaunchKernelGGL((gj_rank_bw2_kernel<RB_BK, RB_WPS, RB_BN>), dim3(T * (np / RB_BN), batch), dim3(256), lds2, 0, src, d2, g, mstride, gk,
                           gkstride, np, ld, mstride, c0, kdim, map, 1, PanelExport{pt2, 0, pt_col, pt_w, 1}, tstride, 0, 0, (const int *)nullptr);
    };
    CK(hipFuncSetAttribute((const void *)rb_persistent_exp_kernel<RB_BK, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    CK(hipFuncSetAttribute((const void *)rb_persistent_exp_kernel<RB_BK, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    auto runP = [&](int wpc, int stagger) {
        if (wpc == 4)
            hipLaunchKernelGGL((rb_persistent_exp_kernel<RB_BK, 4>), dim3(256 * 4, batch), dim3(256), lds2, 0, src, d2, g, mstride, gk,
                               gkstride, np, ld, mstride, c0, kdim, map, 1, PanelExport{pt2, 0, pt_col, pt_w, 1}, tstride, stagger);
        else
            hipLaunchKernelGGL((rb_persistent_exp_kernel<RB_BK, 3>), dim3(256 * 3, batch), dim3(256), lds2, 0, src, d2, g, mstride, gk,
                               gkstride, np, ld, mstride, c0, kdim, map, 1, PanelExport{pt2, 0, pt_col, pt_w, 1}, tstride, stagger);
    };
    run1(); runT(); run2();
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    std::vector<float> o1(mstride * batch), o2(mstride * batch);
    CK(hipMemcpy(o1.data(), d1, o1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o2.data(), d2, o2.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < np; ++i)
            if (memcmp(&o1[(size_t)b * mstride + (size_t)i * ld], &o2[(size_t)b * mstride + (size_t)i * ld], (size_t)np * 4) != 0) ++bad;
    std::vector<float> p1(tstride * batch), p2(tstride * batch);
    CK(hipMemcpy(p1.data(), pt1, p1.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(p2.data(), pt2, p2.size() * 4, hipMemcpyDeviceToHost));
    const bool pt_same = memcmp(p1.data(), p2.data(), p1.size() * 4) == 0;
    printf("gen2 vs gen1: %zu differing rows of %d, panel export %s\n", bad, np * batch, pt_same ? "identical" : "DIFFERENT");

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 2.0 * np * (double)(np - kdim) * kdim * batch;
    auto time_it = [&](const char *name, auto fn) {
        for (int i = 0; i < 3; ++i) fn();
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) fn();
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1000.0 / reps;
        printf("%-28s %9.2f us  %7.1f TFLOP/s\n", name, us, flops / us * 1e-6);
    };
#ifdef MI32_RB_STAMPS
    {   // per-workgroup phase stamps of ONE gen2 launch -> per-CU timeline statistics
        const size_t nwg = (size_t)T * T * batch;
        unsigned long long *dst_stamps;
        CK(hipMalloc(&dst_stamps, nwg * 8 * 8));
        CK(hipMemset(dst_stamps, 0, nwg * 8 * 8));
        hipLaunchKernelGGL(set_stamp_buffer, dim3(1), dim3(1), 0, 0, dst_stamps);
        CK(hipDeviceSynchronize());
        run2(); run2();
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> st(nwg * 8);
        CK(hipMemcpy(st.data(), dst_stamps, st.size() * 8, hipMemcpyDeviceToHost));
        hipLaunchKernelGGL(set_stamp_buffer, dim3(1), dim3(1), 0, 0, (unsigned long long *)nullptr);
        CK(hipDeviceSynchronize());
        FILE *f = fopen("gpurun_out/rb_stamps.csv", "w");
        if (f) {
            fprintf(f, "wg,t0,t1,t2,t3,hwid,xcc\n");
            for (size_t i = 0; i < nwg; ++i)
                if (st[i * 8 + 3])
                    fprintf(f, "%zu,%llu,%llu,%llu,%llu,%llu,%llu\n", i, st[i * 8], st[i * 8 + 1], st[i * 8 + 2], st[i * 8 + 3],
                            st[i * 8 + 4] & 0xffffffffull, st[i * 8 + 4] >> 32);
            fclose(f);
        }
        printf("stamps written for %zu workgroups\n", nwg);
    }
#endif
    time_it("gen1 update", run1);
    time_it("gen2 transpose", runT);
    for (int round = 0; round < 3; ++round) {  // interleaved A/B rounds in one process
        time_it("gen2 update", run2old);
        time_it("gen2 update (again)", run2);
    }
    for (int round = 0; round < 1; ++round) {
        time_it("persistent 4/CU", [&]() { runP(4, 0); });
        time_it("persistent 4/CU stagger 10k", [&]() { runP(4, 10000); });
        time_it("persistent 4/CU stagger 30k", [&]() { runP(4, 30000); });
        time_it("persistent 3/CU", [&]() { runP(3, 0); });
        time_it("persistent 3/CU stagger 40k", [&]() { runP(3, 40000); });
        time_it("gen2 update", run2);
    }
    time_it("gen2 transpose+update", [&]() { runT(); run2(); });
    return bad != 0 || !pt_same;
}
