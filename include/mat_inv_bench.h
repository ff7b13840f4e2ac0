// mat_inv_bench.h -- the reference's experiment-project header behind its own declarations.
//
// /root/reference/matrix_inv_solution/matrix_inversion_solution/matrix_inversion/headers.h:5-16 declares
//     double matrix_multiply(std::vector<double> matriceA, std::vector<double> matriceB);
//     std::vector<float>  matrix_inversion_FP32(std::vector<float> matrix_vector, int matrix_order);
//     std::vector<double> matrix_inversion_FP64(std::vector<double> matrix_vector, int matrix_order);        (mat_inv_64.h)
//     std::vector<double> matrix_inversion_no_pivots(std::vector<double> matrix_vector, int matrix_order);   (mat_inv_64.h)
//     Res no_pivots_bench(...);  Res FP32_bench(...);  Res FP64_bench(...);
// with (res_struct.h:4-6)
//     struct Res { std::vector<double> inversa64; std::vector<double> times; std::vector<float> inversa32; };
// and FP32_bench.cpp:256-443 fills `times` with ten durations in seconds:
//     [0] queue/context creation  [1] buffer creation  [2] program build  [3] makeAugmented
//     [4] pivot (maxPivot + finalMaxPivot + pivotElements)  [5] fixRow  [6] fixColumn
//     [7] compute (the whole step loop)  [8] getInverted  [9] total.
// libmat_inv_32.so exports the same functions on the HIP path:
//   FP32_bench                    -- `inversa32` = the inverse, `times` = the same ten slots (how the fused HIP kernels map
//                                    onto them: mi32_bench_32 in mat_inv_32_c.h);
//   FP64_bench / no_pivots_bench  -- `inversa64` = the inverse in double (blocked fp64 path / no-pivot variant), `times` alike;
//   matrix_inversion_FP32         -- the experiment twin of matrix_inv_32: {} for an invalid matrix, like its exact-identity
//                                    check (matrix_inversion_FP32.cpp:814-835);
//   matrix_multiply               -- the verification helper (matrix_multiply.cpp:15-212): sqrt(N) - ||A * B||_F with the
//                                    product in double on the device (fp64 matrix cores), N = sqrt(size) as the reference
//                                    takes it; NaN for operands that are not two N x N matrices.
// An empty Res for a bad shape or an invalid matrix, like the reference's error paths (FP32_bench.cpp:212,217,456).
#pragma once
#include <vector>

#include "mat_inv_64.h"

struct Res {
    std::vector<double> inversa64;
    std::vector<double> times;
    std::vector<float> inversa32;
};

double matrix_multiply(std::vector<double> matriceA, std::vector<double> matriceB);
std::vector<float> matrix_inversion_FP32(std::vector<float> matrix_vector, int matrix_order);

Res no_pivots_bench(std::vector<double> matrix_vector, int matrix_order);
Res FP32_bench(std::vector<float> matrix_vector, int matrix_order);
Res FP64_bench(std::vector<double> matrix_vector, int matrix_order);
