// Diagnostic: host <-> device copy rates for a 64 MiB pageable buffer (the matrix_inv_32 boundary at N = 4096):
// plain hipMemcpy, hipHostRegister + copy + unregister, and a pinned staging buffer filled by memcpy.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t bytes = 64ull << 20;
    std::vector<float> h(bytes / 4, 1.0f), h2(bytes / 4, 0.0f);
    float *d; hipMalloc(&d, bytes);
    float *pin; hipHostMalloc(&pin, bytes, hipHostMallocDefault);
    hipStream_t s; hipStreamCreate(&s);
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        hipMemcpyAsync(d, h.data(), bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
        double t1 = now();
        hipMemcpyAsync(h2.data(), d, bytes, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
        double t2 = now();
        hipHostRegister(h.data(), bytes, hipHostRegisterDefault);
        double t3 = now();
        hipMemcpyAsync(d, h.data(), bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
        double t4 = now();
        hipHostUnregister(h.data());
        double t5 = now();
        hipHostRegister(h2.data(), bytes, hipHostRegisterDefault);
        double t6 = now();
        hipMemcpyAsync(h2.data(), d, bytes, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
        double t7 = now();
        hipHostUnregister(h2.data());
        double t8 = now();
        memcpy(pin, h.data(), bytes);
        double t9 = now();
        hipMemcpyAsync(d, pin, bytes, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
        double t10 = now();
        printf("pageable H2D %.2f ms  D2H %.2f ms | register %.2f + H2D %.2f + unregister %.2f | register %.2f + D2H %.2f + unregister %.2f | memcpy->pinned %.2f + H2D %.2f\n",
               1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t5 - t4), 1e3 * (t6 - t5), 1e3 * (t7 - t6),
               1e3 * (t8 - t7), 1e3 * (t9 - t8), 1e3 * (t10 - t9));
    }
    return 0;
}
