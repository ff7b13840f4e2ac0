// mat_inv_bench.h -- the reference's benchmark twin behind its own declaration.
//
// /root/reference/matrix_inv_solution/matrix_inversion_solution/matrix_inversion/headers.h:15 declares
//     Res FP32_bench(std::vector<float> matrix_vector, int matrix_order);
// with (res_struct.h:4-6)
//     struct Res { std::vector<double> inversa64; std::vector<double> times; std::vector<float> inversa32; };
// and FP32_bench.cpp:256-443 fills `times` with ten durations in seconds:
//     [0] queue/context creation  [1] buffer creation  [2] program build  [3] makeAugmented
//     [4] pivot (maxPivot + finalMaxPivot + pivotElements)  [5] fixRow  [6] fixColumn
//     [7] compute (the whole step loop)  [8] getInverted  [9] total.
// libmat_inv_32.so exports the same function on the HIP path: `inversa32` = the inverse, `times` = the same ten
// slots (how the fused HIP kernels map onto them: mi32_bench_32 in mat_inv_32_c.h); an empty Res for a bad shape
// or an invalid matrix, like the reference's error paths (FP32_bench.cpp:212,217,456).
#pragma once
#include <vector>

struct Res {
    std::vector<double> inversa64;
    std::vector<double> times;
    std::vector<float> inversa32;
};

Res FP32_bench(std::vector<float> matrix_vector, int matrix_order);
