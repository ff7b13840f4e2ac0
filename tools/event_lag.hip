// Diagnostic (not part of the product): when does a stream that waits for an event recorded EARLY in another, long
// stream actually start?   hipcc --offload-arch=gfx950 -O2 -o tools/event_lag tools/event_lag.hip
//   stream 1: K0, record(ev), K1 .. Kn (each `us` microseconds of spinning on a few workgroups)
//   stream 2: wait(ev), B     (B stamps its start time)
// Prints the start of B relative to the end of K0, for a few ways of creating the event and the streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(long long cycles, long long *stamp)
{
    const long long t0 = wall_clock64();
    if (stamp && threadIdx.x == 0 && blockIdx.x == 0) stamp[0] = t0;
    while (wall_clock64() - t0 < cycles) { }
    if (stamp && threadIdx.x == 0 && blockIdx.x == 0) stamp[1] = wall_clock64();
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static int run(const char *label, bool null_stream, unsigned ev_flags, int n_after, bool record_twice)
{
    hipStream_t s1 = nullptr, s2 = nullptr;
    if (!null_stream) CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, ev_flags));
    long long *st;
    CK(hipMalloc(&st, sizeof(long long) * 2 * (n_after + 2)));
    CK(hipMemset(st, 0, sizeof(long long) * 2 * (n_after + 2)));
    int rate_khz = 0;
    CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const long long cyc = (long long)rate_khz * 100 / 1000;  // 100 us per kernel
    for (int rep = 0; rep < 2; ++rep) {  // the second repetition is the one reported
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(spin_kernel, dim3(8), dim3(64), 0, s1, cyc, st);
        CK(hipEventRecord(ev, s1));
        if (record_twice) CK(hipStreamWaitEvent(s2, ev, 0));
        for (int i = 0; i < n_after; ++i) hipLaunchKernelGGL(spin_kernel, dim3(8), dim3(64), 0, s1, cyc, st + 2 * (i + 2));
        if (!record_twice) CK(hipStreamWaitEvent(s2, ev, 0));
        hipLaunchKernelGGL(spin_kernel, dim3(8), dim3(64), 0, s2, cyc, st + 2);
        CK(hipDeviceSynchronize());
    }
    std::vector<long long> h(2 * (n_after + 2));
    CK(hipMemcpy(h.data(), st, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
    const double us = 1e3 / rate_khz;
    std::printf("%-58s B starts %8.1f us after K0 ends (stream 1 runs on for %8.1f us)\n", label, (h[2] - h[1]) * us,
                (h[2 * (n_after + 1) + 1] - h[1]) * us);
    CK(hipFree(st));
    CK(hipEventDestroy(ev));
    if (s1) CK(hipStreamDestroy(s1));
    CK(hipStreamDestroy(s2));
    return 0;
}

int main()
{
    run("own stream, DisableTiming, wait enqueued last", false, hipEventDisableTiming, 50, false);
    run("own stream, DisableTiming, wait enqueued right after record", false, hipEventDisableTiming, 50, true);
    run("own stream, default flags, wait enqueued last", false, hipEventDefault, 50, false);
    run("own stream, default flags, wait right after record", false, hipEventDefault, 50, true);
    run("null stream, DisableTiming, wait enqueued last", true, hipEventDisableTiming, 50, false);
    run("null stream, DisableTiming, wait right after record", true, hipEventDisableTiming, 50, true);
    run("null stream, default flags, wait right after record", true, hipEventDefault, 50, true);
    return 0;
}
