// Scalar polynomial helpers of the ANM driver (host side).
// Mirrors libsanm/unary_polynomial.{h,cpp}.
#pragma once
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

//! real roots of sum f[i] x^i; returns false if the iteration fails
//! (role of unary_polynomial::roots(only_real=true), unary_polynomial.cpp:154-334)
bool real_roots(const std::vector<double>& f, std::vector<double>& roots);

}  // namespace poly
}  // namespace sanm_hip
