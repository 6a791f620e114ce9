// ORACLE -- test infrastructure only (see oracle/__init__.py).
//
// extern "C" entry points around the part of the reference that compiles in this image without any of its
// missing dependencies: libsanm/unary_polynomial.cpp (+ libsanm/utils.cpp, third_party/BRENT/brent.cpp).
// This file is OURS; the reference sources are compiled from where they lie under /root/reference by
// oracle/build_ref.py and never copied.  Outputs go to oracle/_ref/ (git-ignored).
//
// Used to (a) pin oracle/unary_polynomial.py (ACM algorithm 30 restatement, Brent zero) against the reference
// itself and (b) generate tests/golden/ref_poly.json (tests/golden/make_ref_poly.py).
#include "libsanm/unary_polynomial.h"

#include <cstring>

#define REF_API extern "C" __attribute__((visibility("default")))

using namespace sanm;

//! unary_polynomial::roots (unary_polynomial.cpp:154-334).  Returns the number of roots written to re/im
//! (capacity cap), or -1 when the reference returns None, or -2 on a reference assertion.
REF_API int ref_poly_roots(const double* f, int n, int only_real, int max_iter, double tol, double* re, double* im,
                           int cap) {
    try {
        auto r = unary_polynomial::roots({f, f + n}, only_real != 0, max_iter, tol);
        if (!r.valid()) return -1;
        int k = 0;
        for (auto z : r.val()) {
            if (k < cap) {
                re[k] = z.real();
                im[k] = z.imag();
            }
            ++k;
        }
        return k;
    } catch (std::exception&) {
        return -2;
    }
}

//! unary_polynomial::eval (unary_polynomial.cpp:71-77)
REF_API double ref_poly_eval(const double* f, int n, double x) { return unary_polynomial::eval({f, f + n}, x); }

//! unary_polynomial::solve_eqn (unary_polynomial.cpp:88-95); *ok = 0 on a reference assertion
REF_API double ref_poly_solve_eqn(const double* f, int n, double xmin, double xmax, double b, double eps, int* ok) {
    try {
        *ok = 1;
        return unary_polynomial::solve_eqn({f, f + n}, xmin, xmax, b, eps);
    } catch (std::exception&) {
        *ok = 0;
        return 0;
    }
}

//! unary_polynomial::stable_x_range (unary_polynomial.cpp:97-103)
REF_API double ref_poly_stable_x_range(int order) { return unary_polynomial::stable_x_range(order); }

//! unary_polynomial::solve_quad (unary_polynomial.cpp:79-86)
REF_API double ref_poly_solve_quad(double a, double b, double c) { return unary_polynomial::solve_quad(a, b, c); }

//! unary_polynomial::minimize / maximize (unary_polynomial.cpp:105-113)
REF_API void ref_poly_minimize(const double* f, int n, double xmin, double xmax, double eps, int is_max, double* out) {
    auto r = is_max ? unary_polynomial::maximize({f, f + n}, xmin, xmax, eps)
                    : unary_polynomial::minimize({f, f + n}, xmin, xmax, eps);
    out[0] = r.first;
    out[1] = r.second;
}
