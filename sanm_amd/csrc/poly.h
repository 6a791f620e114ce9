// Scalar polynomial helpers of the ANM driver (host side).
// Mirrors libsanm/unary_polynomial.{h,cpp}.
#pragma once
#include <complex>
#include <functional>
#include <vector>

namespace sanm_hip {
namespace poly {

//! Horner evaluation, coefficients low order first (unary_polynomial.cpp:71-77)
double eval(const double* f, int n, double x);
inline double eval(const std::vector<double>& f, double x) { return eval(f.data(), (int)f.size(), x); }

//! 1e15^(1/order) (unary_polynomial.cpp:97-103)
double stable_x_range(int order);

//! Brent's zero finder on a change-of-sign interval, tolerance
//! 2*macheps*|x| + t (R. Brent, "Algorithms for Minimization Without
//! Derivatives", procedure zero; the reference links third_party/BRENT)
double brent_zero(double a, double b, double t, const std::function<double(double)>& f);

//! x in [xmin,xmax] with f(x) = b (unary_polynomial.cpp:88-95)
double solve_eqn(const std::vector<double>& f, double xmin, double xmax, double b = 0,
                 double eps = 1e-6);

//! all roots of sum f[i] x^i by ACM algorithm 30 (Bairstow + Newton), complex pairs omitted when only_real;
//! returns false where the reference returns None (unary_polynomial::roots, unary_polynomial.cpp:154-334,
//! defaults unary_polynomial.h:50-52).  Roots come in the order the algorithm finds them.
bool roots(const std::vector<double>& f, bool only_real, std::vector<std::complex<double>>& out,
           int max_iter = 300, double tol = 1e-8);

//! roots(f, only_real = true) as pade.cpp:113 calls it: real parts only
bool real_roots(const std::vector<double>& f, std::vector<double>& roots);

}  // namespace poly
}  // namespace sanm_hip
