// mat_inv_64.h -- the reference's second precision behind its own declaration.
//
// /root/reference/matrix_inv_solution/matrix_inversion_solution/matrix_inversion/headers.h:9 declares
//     std::vector<double> matrix_inversion_FP64(std::vector<double> matrix_vector, int matrix_order);
// (body matrix_inversion_FP64.cpp:13: the five-kernel Gauss-Jordan step of the fp32 library in double).
// libmat_inv_32.so exports the same function: row-major N x N in, row-major inverse out, empty vector for a
// bad shape (N <= 0, size / N != N) or a singular / NaN matrix (the reference returns {} when the reduced left
// half is not exactly I, matrix_inversion_FP64.cpp:846-867).  Flat-pointer twins: mi32_matrix_inv_64 and
// mi32_inv_device_f64 in mat_inv_32_c.h.
#pragma once
#include <vector>
std::vector<double> matrix_inversion_FP64(std::vector<double> matrix_vector, int matrix_order);
inline std::vector<double> matrix_inv_64(std::vector<double> matrix_vector, int matrix_order)
{
    return matrix_inversion_FP64(static_cast<std::vector<double> &&>(matrix_vector), matrix_order);
}

// The reference's no-pivot variant (headers.h:11, body matrix_inversion_no_pivots.cpp:10): Gauss-Jordan in double
// with the diagonal entry as the pivot of every step (findCrr / fixRow / copyCirColumn / fixColumn, kernels :13-70)
// -- for diagonally dominant inputs.  Same call shape; an empty vector for a bad shape or when a zero / non-finite
// diagonal entry is met (the reference returns {} when the reduced left half is not exactly I, :670).
std::vector<double> matrix_inversion_no_pivots(std::vector<double> matrix_vector, int matrix_order);
