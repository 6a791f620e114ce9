// Per-row bodies of the sparse kernels (shared between the HIP kernels and the
// test-only host harness, like tet_ops.h).
#pragma once
#include <cmath>

#include "backend.h"
#include "tet_ops.h"

namespace sanm_hip {

// dst[i] = sum coef*src[idx]   (SparseLinearDesc::apply, libsanm/anm.cpp:55-75)
SANM_HD double gather_row(const SparseRowsDev& R, const double* src, int64_t i) {
    double s = 0;
    for (uint32_t p = R.ptr[i], e = R.ptr[i + 1]; p < e; ++p) s += R.coef[p] * src[R.idx[p]];
    return s;
}

// One CSR value: sum of its contributions, each dropped if |c| < 1e-9
// (SparseMatBuilder::add_constraint, libsanm/sparse_solver.cpp:286-305)
SANM_HD double assemble_slot(const AssemblyDev& A, const double* jac, int64_t s) {
    double v = 0;
    for (uint32_t p = A.ptr[s], e = A.ptr[s + 1]; p < e; ++p) {
        double c = A.coef[p] * jac[A.jidx[p]];
        if (fabs(c) >= 1e-9) v += c;
    }
    return v;
}

SANM_HD double spmv_row(const CsrDev& A, const double* x, int64_t i) {
    double s = 0;
    for (uint32_t p = A.rowptr[i], e = A.rowptr[i + 1]; p < e; ++p) s += A.val[p] * x[A.col[p]];
    return s;
}

// one entry of A'A + lambda I: sparse dot product of two columns of A (rows ascending in both), merge order
SANM_HD double ata_entry(const CsrDev& At, uint32_t i, uint32_t j, double lambda) {
    uint32_t p = At.rowptr[i], pe = At.rowptr[i + 1], q = At.rowptr[j], qe = At.rowptr[j + 1];
    double s = i == j ? lambda : 0.0;
    while (p < pe && q < qe) {
        const uint32_t a = At.col[p], b = At.col[q];
        if (a == b) s += At.val[p] * At.val[q];
        p += a <= b;
        q += b <= a;
    }
    return s;
}

SANM_HD double csr_diag(const CsrDev& A, int64_t i) {
    for (uint32_t p = A.rowptr[i], e = A.rowptr[i + 1]; p < e; ++p)
        if ((int64_t)A.col[p] == i) return A.val[p];
    return 0.0;
}

// |a-b| - eps*max(1, min(|a|,|b|))   (TensorND::assert_allclose, tensor.cpp:670-684)
SANM_HD double allclose_excess1(double a, double b, double eps) {
    double m = fmin(fabs(a), fabs(b));
    if (m < 1.0) m = 1.0;
    double d = fabs(a - b);
    // non-finite values must fail the check
    if (!(d == d)) return 1e300;
    return d - eps * m;
}

}  // namespace sanm_hip
