// Diagnostic: wall-clock of the reference's call shape, std::vector in / std::vector out, N = 4096.
// Build: g++ -O2 -std=c++17 -I include tools/cxx_e2e.cpp -o tools/cxx_e2e -L gpu_matrix_inversion_amd/lib -lmat_inv_32 -Wl,-rpath,$PWD/gpu_matrix_inversion_amd/lib
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "mat_inv_32.h"
#include "mat_inv_32_c.h"
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 4096;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-1.f, 1.f);
    std::vector<float> a((size_t)n * n);
    for (auto &v : a) v = u(rng);
    for (int i = 0; i < n; ++i) a[(size_t)i * n + i] += std::sqrt((float)n);
    for (int rep = 0; rep < 6; ++rep) {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<float> x = matrix_inv_32(a, n);
        const auto t1 = std::chrono::steady_clock::now();
        double s = 0;
        for (size_t i = 0; i < x.size(); i += 4097) s += x[i];
        double tot = 0, comp = 0;
        mi32_last_timing(&tot, &comp);
        const auto c0 = std::chrono::steady_clock::now();
        std::vector<float> copy = a;   // what pass-by-value costs the caller
        const auto c1 = std::chrono::steady_clock::now();
        std::vector<float> zero(a.size(), 0.0f);
        const auto c2 = std::chrono::steady_clock::now();
        s += copy[5] + zero[7];
        std::printf("call %d: %.2f ms  (library total %.2f, compute %.2f; a 64 MiB vector copy here %.2f ms, a zero-filled one %.2f ms; checksum %.6f)\n",
                    rep, std::chrono::duration<double, std::milli>(t1 - t0).count(), tot * 1e3, comp * 1e3,
                    std::chrono::duration<double, std::milli>(c1 - c0).count(), std::chrono::duration<double, std::milli>(c2 - c1).count(), s);
    }
    return 0;
}
