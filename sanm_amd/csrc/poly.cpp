#include "poly.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <limits>

#include "graph.h"

namespace sanm_hip {
namespace poly {

double eval(const double* f, int n, double x) {
    double r = 0;
    for (int i = n - 1; i >= 0; --i) r = r * x + f[i];
    return r;
}

double stable_x_range(int order) { return std::pow(1e15, 1.0 / static_cast<double>(order)); }

double brent_zero(double a, double b, double t, const std::function<double(double)>& f) {
    const double macheps = std::numeric_limits<double>::epsilon();
    double sa = a, sb = b, fa = f(sa), fb = f(sb);
    double c = sa, fc = fa, e = sb - sa, d = e;
    for (;;) {
        if (std::fabs(fc) < std::fabs(fb)) {
            sa = sb; sb = c; c = sa;
            fa = fb; fb = fc; fc = fa;
        }
        double tol = 2.0 * macheps * std::fabs(sb) + t;
        double m = 0.5 * (c - sb);
        if (std::fabs(m) <= tol || fb == 0.0) break;
        if (std::fabs(e) < tol || std::fabs(fa) <= std::fabs(fb)) {
            e = m;
            d = e;
        } else {
            double p, q, r, s = fb / fa;
            if (sa == c) {
                p = 2.0 * m * s;
                q = 1.0 - s;
            } else {
                q = fa / fc;
                r = fb / fc;
                p = s * (2.0 * m * q * (q - r) - (sb - sa) * (r - 1.0));
                q = (q - 1.0) * (r - 1.0) * (s - 1.0);
            }
            if (0.0 < p) q = -q; else p = -p;
            s = e;
            e = d;
            if (2.0 * p < 3.0 * m * q - std::fabs(tol * q) && p < std::fabs(0.5 * s * q)) {
                d = p / q;
            } else {
                e = m;
                d = e;
            }
        }
        sa = sb;
        fa = fb;
        if (tol < std::fabs(d)) sb += d;
        else if (0.0 < m) sb += tol;
        else sb -= tol;
        fb = f(sb);
        if ((0.0 < fb && 0.0 < fc) || (fb <= 0.0 && fc <= 0.0)) {
            c = sa;
            fc = fa;
            e = sb - sa;
            d = e;
        }
    }
    return sb;
}

double solve_eqn(const std::vector<double>& f, double xmin, double xmax, double b, double eps) {
    sanm_check(!f.empty() && xmin < xmax, "solve_eqn: bad interval [%g, %g]", xmin, xmax);
    auto fn = [&](double x) { return eval(f, x) - b; };
    double f0 = fn(xmin), f1 = fn(xmax);
    sanm_check(f0 * f1 <= 0, "no zero point: f0=%g f1=%g", f0, f1);
    return brent_zero(xmin, xmax, eps, fn);
}

// All complex roots by the Aberth-Ehrlich simultaneous iteration, then the
// real ones are kept.  The reference uses ACM algorithm 30 (Bairstow+Newton);
// only the set of real roots matters to its caller (pade.cpp:113-126).
bool real_roots(const std::vector<double>& f, std::vector<double>& roots) {
    roots.clear();
    int n = (int)f.size() - 1;
    while (n >= 0 && f[n] == 0.0) --n;
    if (n <= 0) return true;
    // strip zero roots
    int lo = 0;
    while (lo < n && f[lo] == 0.0) ++lo;
    for (int i = 0; i < lo; ++i) roots.push_back(0.0);
    std::vector<double> c(f.begin() + lo, f.begin() + n + 1);
    n -= lo;
    if (n == 0) return true;
    // monic, long double for a little headroom.  Complex arithmetic is spelt out on the components: the
    // library's std::complex<long double> operators go through the overflow / NaN recovery paths of
    // __mulxc3 / __divxc3 and made this loop 4x slower (0.53 -> 0.13 ms at degree 19) -- the GPU waits for it once per continuation step.
    // (double, not long double: x87 arithmetic made the iteration 3x slower, and the caller -- the pole bound of the
    // Pade range -- needs the real roots to a few digits)
    using ld = double;
    struct cd {
        ld re, im;
    };
    auto mul = [](cd x, cd y) { return cd{x.re * y.re - x.im * y.im, x.re * y.im + x.im * y.re}; };
    auto inv = [](cd x) {
        const ld d = x.re * x.re + x.im * x.im;
        return cd{x.re / d, -x.im / d};
    };
    auto cabs = [](cd x) { return std::hypot(x.re, x.im); };
    std::vector<ld> a(n + 1);
    for (int i = 0; i <= n; ++i) a[i] = (ld)c[i] / (ld)c[n];
    // Cauchy bound based start radius
    ld radius = 0;
    for (int i = 0; i < n; ++i) radius = std::max(radius, std::pow(std::fabs(a[i]), 1.0 / (n - i)));
    if (radius == 0) {
        for (int i = 0; i < n; ++i) roots.push_back(0.0);
        return true;
    }
    std::vector<cd> z(n);
    for (int i = 0; i < n; ++i) {
        ld ang = 2.0 * 3.14159265358979323846264338327950288 * i / n + 0.4;
        z[i] = cd{radius * std::cos(ang), radius * std::sin(ang)};
    }
    bool converged = false;
    for (int it = 0; it < 500 && !converged; ++it) {
        ld maxstep = 0;
        for (int i = 0; i < n; ++i) {
            cd p{1.0, 0.0}, dp{0.0, 0.0};  // Horner for p and p'
            for (int k = n - 1; k >= 0; --k) {
                dp = mul(dp, z[i]);
                dp.re += p.re;
                dp.im += p.im;
                p = mul(p, z[i]);
                p.re += a[k];
            }
            if (p.re == 0 && p.im == 0) continue;
            const cd ratio = mul(p, inv(dp));
            cd sum{0.0, 0.0};
            for (int j = 0; j < n; ++j)
                if (j != i) {
                    const cd t = inv(cd{z[i].re - z[j].re, z[i].im - z[j].im});
                    sum.re += t.re;
                    sum.im += t.im;
                }
            const cd rs = mul(ratio, sum);
            const cd step = mul(ratio, inv(cd{1.0 - rs.re, -rs.im}));
            z[i].re -= step.re;
            z[i].im -= step.im;
            maxstep = std::max(maxstep, cabs(step) / std::max<ld>(cabs(z[i]), 1e-300));
        }
        if (maxstep < 1e-14) converged = true;
    }
    if (!converged) {
        // accept if every root has a tiny residual anyway
        for (int i = 0; i < n; ++i) {
            cd p{1.0, 0.0};
            ld scale = 1.0;
            const ld az = cabs(z[i]);
            for (int k = n - 1; k >= 0; --k) {
                p = mul(p, z[i]);
                p.re += a[k];
                scale = scale * az + std::fabs(a[k]);
            }
            if (cabs(p) > 1e-10 * scale) return false;
        }
    }
    for (int i = 0; i < n; ++i) {
        const double re = z[i].re, im = z[i].im;
        if (std::fabs(im) <= 1e-8 * std::max(1.0, std::fabs(re)))
            roots.push_back((double)re);
    }
    std::sort(roots.begin(), roots.end());
    return true;
}

}  // namespace poly
}  // namespace sanm_hip
