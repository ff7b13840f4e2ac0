#pragma once
#include <vector>

// Drop-in replacement for the reference library header
// (/root/reference/Matlab/mat_inv_32.h:1-4, identical copy at
// Matlab/mat_inv_32/mat_inv_32/mat_inv_32.h): same name, same C++ signature,
// same semantics -- row-major flattened N x N fp32 in, row-major flattened
// inverse out, EMPTY vector for an invalid matrix (README.md:54,
// mat_inv_32.cpp:206-215).  Implemented in libmat_inv_32.so on top of the
// C ABI of mat_inv_32_c.h (HIP kernels for gfx950, no OpenCL, no per-call JIT).
std::vector<float> matrix_inv_32(std::vector<float> matrix_vector, int matrix_order);
