// Diagnostic (not part of the product): round-trip latency of an 8-byte {payload, tag} granule between two workgroups --
// what one pivot step of a shared panel pays for its exchange -- for partners on different XCDs (agent-scope accesses,
// what the product does) and on the same XCD (agent scope, and L2-coherent accesses that only bypass the CU's L1).
// Workgroup i of a grid goes to XCD i mod 8; the two partners are workgroups 0 and `other` of a grid of `other + 1`.
// Build: hipcc -O3 --offload-arch=gfx950 tools/xchg_latency.hip -o tools/xchg_latency
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>  // 0: agent-scope atomics (sc1), 1: L1-bypassing accesses coherent in the XCD's L2 (sc0)
__device__ __forceinline__ unsigned long long g_load(const unsigned long long *p)
{
    if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE>
__device__ __forceinline__ void g_store(unsigned long long *p, unsigned long long v)
{
    if (MODE == 0) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(64) void pingpong(unsigned long long *buf, int other, int rounds, unsigned long long *out)
{
    const int me = (blockIdx.x == 0) ? 0 : ((int)blockIdx.x == other ? 1 : -1);
    if (me < 0) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x != 0) return;
    unsigned long long *mine = buf + me * 32, *theirs = buf + (1 - me) * 32;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long spins = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (me == 0) {
            g_store<MODE>(mine, (unsigned long long)r);
            while (g_load<MODE>(theirs) != (unsigned long long)r)
                if (++spins > 2000000ull) { out[4] = 1; return; }
        } else {
            while (g_load<MODE>(theirs) != (unsigned long long)r)
                if (++spins > 2000000ull) { out[4] = 1; return; }
            g_store<MODE>(mine, (unsigned long long)r);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[me * 2] = t1 - t0;
    out[me * 2 + 1] = xcc & 7;
}

template <int MODE>
static void run(const char *name, int other)
{
    unsigned long long *buf, *out, h[8] = {};
    hipMalloc(&buf, 64 * 8);
    hipMalloc(&out, 8 * 8);
    const int rounds = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(buf, 0, 64 * 8);
        hipMemset(out, 0, 8 * 8);
        hipLaunchKernelGGL((pingpong<MODE>), dim3(other + 1), dim3(64), 0, 0, buf, other, rounds, out);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-44s workgroups 0 and %2d (XCD %llu and %llu): %.3f us per round trip%s\n", name, other, h[1], h[3],
           (double)h[0] / 100.0 / rounds, h[4] ? "  [TIMED OUT]" : "");
    hipFree(buf);
    hipFree(out);
}

int main()
{
    run<0>("agent scope (sc1), different XCDs", 1);
    run<0>("agent scope (sc1), different XCDs", 3);
    run<0>("agent scope (sc1), same XCD", 8);
    run<1>("L1 bypass, L2 coherent (sc0), same XCD", 8);
    run<1>("L1 bypass, L2 coherent (sc0), same XCD", 16);
    return 0;
}
