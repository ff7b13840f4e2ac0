// panel_probe.hip -- diagnostic only (not part of the product): runs gj_panel_kernel alone on a
// 4096-row compact panel with s_memtime stamps at the phase boundaries of every pivot step.
// Build: hipcc -O3 --offload-arch=gfx950 -DMI32_STAMPS -fno-slp-vectorize -ffp-contract=off \
//        -I include -I gpu_matrix_inversion_amd/csrc tools/panel_probe.hip -o tools/panel_probe
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../gpu_matrix_inversion_amd/csrc/mi32_blocked.hip"

using namespace mi32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#ifndef PROBE_NT
#define PROBE_NT 1024
#endif
#ifndef PROBE_RPT
#define PROBE_RPT 4
#endif

__global__ void clock_probe(unsigned long long *out)
{
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < 200000; ++i) __builtin_amdgcn_s_sleep(1);
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[0] = t1 - t0; out[1] = r1 - r0;
}

int main()
{
    const int np = 4096, n = 4096, W = 16;
    printf("probe: NT=%d RPT=%d W=%d\n", PROBE_NT, PROBE_RPT, W);
    std::vector<float> h((size_t)W * np);
    srand(1);
    for (auto &v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *pt, *gt; int *maps; unsigned long long *stamps, *clk; int *status;
    CK(hipMalloc(&pt, h.size() * 4)); CK(hipMalloc(&gt, h.size() * 4)); CK(hipMalloc(&maps, 3 * np * 4));
    CK(hipMalloc(&stamps, 16 * 8 * 8)); CK(hipMalloc(&clk, 16)); CK(hipMalloc(&status, 4));
    CK(hipMemcpy(pt, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(maps, 0, 3 * np * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; ++variant) {
        unsigned long long *sb = variant ? stamps : nullptr;   // variant 0: stamps disabled (null buffer)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int it = 0; it < 20; ++it)
                hipLaunchKernelGGL((gj_panel_kernel<PROBE_NT, PROBE_RPT, 16>), dim3(1), dim3(PROBE_NT), 2 * PROBE_RPT * PROBE_NT * sizeof(int), 0, pt, gt, np, n,
                                   (size_t)W * np, 256, maps, maps + np, maps + 2 * np, 1, status, sb);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s rep %d: %.2f us per panel launch (20 back-to-back)\n", variant ? "stamped " : "unstamped", rep, ms * 1000 / 20);
        }
    }
    hipLaunchKernelGGL(clock_probe, dim3(1), dim3(64), 0, 0, clk);
    CK(hipDeviceSynchronize());
    unsigned long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
    printf("shader clock ~ %.0f MHz (memtime/memrealtime*100)\n", 100.0 * hc[0] / hc[1]);
    std::vector<unsigned long long> st(16 * 8);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    const char *names[8] = {"extract+local-max", "dpp-max+index", "cand-row+divide+publish", "barrier", "key+prn-read", "eliminate", "fixup+labels", "->next"};
    double sum[8] = {0};
    for (int r = 0; r < 16; ++r)
        for (int s = 0; s < 8; ++s) {
            unsigned long long a = st[r * 8 + s];
            unsigned long long b = (s < 7) ? st[r * 8 + s + 1] : (r < 15 ? st[(r + 1) * 8] : a);
            sum[s] += (double)(b - a);
        }
    double tot = 0;
    for (int s = 0; s < 8; ++s) { printf("%-22s %8.0f cycles/step\n", names[s], sum[s] / (s == 7 ? 15 : 16)); tot += sum[s] / (s == 7 ? 15 : 16); }
    printf("sum %.0f cycles/step; steps span %.0f cycles total\n", tot, (double)(st[15 * 8 + 7] - st[0]));
    return 0;
}
