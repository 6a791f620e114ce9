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
    if (R.bptr) {  // rows in triples: the list of row 3u with its indices shifted by 3 (i - 3u)
        const uint32_t sh = 3u * (uint32_t)(i % 3);
        for (uint32_t p = R.bptr[i / 3], e = R.bptr[i / 3 + 1]; p < e; ++p) s += R.bcoef[p] * src[R.bidx[p] + sh];
        return s;
    }
    for (uint32_t p = R.ptr[i], e = R.ptr[i + 1]; p < e; ++p) s += R.coef[p] * src[R.idx[p]];
    return s;
}

// One row of the Jacobian's values (backend.h: AssemblyDev): the triples (p, m, q) in order, each product dropped if
// |c| < 1e-9 (SparseMatBuilder::add_constraint, libsanm/sparse_solver.cpp:286-305), added to the non-zero of its column.
// pos: scratch of n + 1 entries, all -1 on entry and on exit.  Host harness only (the device kernel stages the triples
// in LDS: backend_hip.hip, assemble_kernel); both add a non-zero's contributions in the same order.
inline void assemble_row(const AssemblyDev& A, const double* jac, int64_t i, double* val, double* grad_t, int32_t* pos) {
    const uint32_t c0 = A.rowptr[i], c1 = A.rowptr[i + 1];
    for (uint32_t p = c0; p < c1; ++p) {
        pos[A.col[p]] = (int32_t)(p - c0);
        val[p] = 0.0;
    }
    double gt = 0.0;
    for (uint32_t p = A.ro_ptr[i]; p < A.ro_ptr[i + 1]; ++p) {
        const uint32_t e = A.ro_idx[p];
        const int64_t b = e / A.odim;
        if (b < A.tet_begin || b >= A.tet_end) continue;
        const int o = (int)(e % A.odim);
        const double c_out = A.ro_coef[p];
        for (int m = 0; m < A.idim; ++m) {
            const double J = jac[((b - A.tet_begin) * A.odim + o) * A.idim + m];
            const int64_t irow = b * A.idim + m;
            for (uint32_t q = A.ri_ptr[irow]; q < A.ri_ptr[irow + 1]; ++q) {
                const double t = (c_out * A.ri_coef[q]) * J;
                if (!(fabs(t) >= 1e-9)) continue;
                const uint32_t c = A.ri_idx[q];
                if ((int64_t)c == A.n) gt += t;
                else val[c0 + pos[c]] += t;
            }
        }
    }
    for (uint32_t p = c0; p < c1; ++p) pos[A.col[p]] = -1;
    if (A.has_t && grad_t) grad_t[i] = gt;
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
