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

// Zero of f on a change-of-sign interval by the method of R. P. Brent, "Algorithms for Minimization Without
// Derivatives" (Prentice-Hall 1973), chapter 4, procedure `zero`: the best point `best` (smallest |f| seen), the
// previous best `prev`, and `brk`, the last point whose f has the other sign -- the zero lies between `best` and
// `brk`.  A step is inverse quadratic interpolation through the three points (linear when only two are distinct),
// accepted when it stays in the first three quarters of the bracket and is at most half the step before last;
// otherwise the bracket is bisected.  Steps shorter than the tolerance are replaced by a tolerance-sized step towards
// `brk`.  The reference calls the vendored third_party/BRENT for `solve_a` (unary_polynomial.cpp:88-95); the
// restart parameter must come out bit for bit, so the FORMULAS below are evaluated in the operation order of
// Brent's publication (which the vendored code also follows) -- 200/200 identical zeros against the reference's own
// translation unit (tests/golden/ref_poly.json).  The organisation (state record, named steps) is this file's.
namespace {
struct BrentState {
    double best, f_best;  // current iterate
    double prev, f_prev;  // the iterate before it
    double brk, f_brk;    // bracket end: f_brk and f_best differ in sign (or one is zero)
    double step;          // the step about to be taken
    double before;        // the step before that (acceptance test of the interpolation)

    //! the bracket end becomes the previous iterate (after a sign change, or at the start)
    void rebracket() {
        brk = prev;
        f_brk = f_prev;
        before = best - prev;
        step = before;
    }
    //! keep the smaller residual in `best`
    void order_by_residual() {
        if (std::fabs(f_brk) < std::fabs(f_best)) {
            prev = best;
            best = brk;
            brk = prev;
            f_prev = f_best;
            f_best = f_brk;
            f_brk = f_prev;
        }
    }
    void bisect(double half) {
        before = half;
        step = before;
    }
    //! secant / inverse quadratic step; falls back to bisection when it is not trusted
    void interpolate(double half, double tol) {
        double num, den;
        const double ratio = f_best / f_prev;
        if (prev == brk) {  // two distinct points: secant
            num = 2.0 * half * ratio;
            den = 1.0 - ratio;
        } else {  // three: inverse quadratic
            const double qa = f_prev / f_brk, qb = f_best / f_brk;
            num = ratio * (2.0 * half * qa * (qa - qb) - (best - prev) * (qb - 1.0));
            den = (qa - 1.0) * (qb - 1.0) * (ratio - 1.0);
        }
        if (0.0 < num) den = -den;
        else num = -num;
        const double two_back = before;
        before = step;
        if (2.0 * num < 3.0 * half * den - std::fabs(tol * den) && num < std::fabs(0.5 * two_back * den)) step = num / den;
        else bisect(half);
    }
};
}  // namespace

double brent_zero(double a, double b, double t, const std::function<double(double)>& f) {
    const double macheps = std::numeric_limits<double>::epsilon();
    BrentState z{};
    z.prev = a;
    z.f_prev = f(a);
    z.best = b;
    z.f_best = f(b);
    z.rebracket();
    for (;;) {
        z.order_by_residual();
        const double tol = 2.0 * macheps * std::fabs(z.best) + t;
        const double half = 0.5 * (z.brk - z.best);
        if (std::fabs(half) <= tol || z.f_best == 0.0) return z.best;
        if (std::fabs(z.before) < tol || std::fabs(z.f_prev) <= std::fabs(z.f_best)) z.bisect(half);
        else z.interpolate(half, tol);
        z.prev = z.best;
        z.f_prev = z.f_best;
        if (tol < std::fabs(z.step)) z.best += z.step;
        else if (0.0 < half) z.best += tol;
        else z.best -= tol;
        z.f_best = f(z.best);
        const bool same_side = (0.0 < z.f_best && 0.0 < z.f_brk) || (z.f_best <= 0.0 && z.f_brk <= 0.0);
        if (same_side) z.rebracket();
    }
}

double solve_eqn(const std::vector<double>& f, double xmin, double xmax, double b, double eps) {
    sanm_check(!f.empty() && xmin < xmax, "solve_eqn: bad interval [%g, %g]", xmin, xmax);
    auto fn = [&](double x) { return eval(f, x) - b; };
    double f0 = fn(xmin), f1 = fn(xmax);
    sanm_check(f0 * f1 <= 0, "no zero point: f0=%g f1=%g", f0, f1);
    return brent_zero(xmin, xmax, eps, fn);
}

// ---------------------------------------------------------------------------------------------------
// ACM algorithm 30 (K. W. Ellenberger, "On programming the numerical solution of polynomial equations",
// Commun. ACM 3(12), 1960, 644-647): simultaneous Bairstow (quadratic factor x^2 + p x + q) and Newton
// (linear factor x - r) iterated synthetic division on the polynomial or -- whichever converges -- its
// reciprocal; after max_iter trips without a factor the polynomial is reversed, and after every second
// failed round the accuracy requirement K drops by a decimal figure; below 1e-8 the search gives up.
//
// The reference's unary_polynomial::roots (unary_polynomial.cpp:154-334, defaults unary_polynomial.h:50-52)
// is this algorithm, and the Pade range estimate REJECTS the approximant when it gives up (pade.cpp:113-116)
// -- a discrete decision that changes continuation step counts.  On the ill-scaled degree-19 denominators of
// the Pade approximant the outcome depends on every rounding, so the arithmetic below follows the
// reference's order of operations exactly (strict double arithmetic: this library is built with
// -ffp-contract=off); tests/golden/ref_poly.json holds outcomes of the reference's own translation unit
// (g++ -O2) and tests/test_device_ops.py::test_host_poly_helpers compares valid flag and roots with them.
// Kept quirks: zero constant terms are stripped without reporting the root 0; work arrays are not cleared
// between deflations.
namespace {
class Acm30 {
    // work arrays indexed -2 .. n
    std::vector<double> m_store;
    double *h, *b, *c, *d, *e;
    int n, dir = 1;            // dir < 0: working on the reciprocal polynomial
    double K;                  // relative accuracy required of a factor
    double p = 0, q = 0, r = 0;
    double p_prev = 0, q_prev = 0, p_last = 0, q_last = 0;  // quadratic factors of the two previous rounds
    double flip = 1;
    const int max_iter;
    const bool only_real;
    std::vector<std::complex<double>>& out;

    enum class At { Init, Reverse, Start, Iterate, Linear, Quadratic, Done, GiveUp };

    void reverse() {
        dir = -dir;
        for (int j = (n - 1) / 2; j >= 0; --j) std::swap(h[j], h[n - j]);
    }
    static void deflate(double* hh, const double* quot, int deg, double K) {
        // keep a quotient coefficient only while it carries significance relative to the dividend's
        for (int j = deg; j >= 0; --j) hh[j] = (quot[j] != 0.0 && std::fabs(hh[j] / quot[j]) < K) ? quot[j] : 0.0;
    }

    At init() {
        if (n == 0) return At::Done;
        p_prev = q_prev = p_last = q_last = 0.0;
        flip = 1.0;
        if (n == 1) {
            r = -h[1] / h[0];
            return At::Linear;
        }
        double s = 0.0;  // scale by the geometric mean of the coefficients
        for (int j = n; j >= 0; --j)
            if (h[j] != 0.0) s += std::log(std::fabs(h[j]));
        s = std::exp(s / (n + 1));
        for (int j = n; j >= 0; --j) h[j] /= s;
        return std::fabs(h[1] / h[0]) < std::fabs(h[n - 1] / h[n]) ? At::Reverse : At::Start;
    }
    At start() {
        if (q_prev != 0.0) {
            p = p_prev;
            q = q_prev;
            return At::Iterate;
        }
        if (h[n - 2] == 0.0) {
            q = 1.0;
            p = -2.0;
        } else {
            q = h[n] / h[n - 2];
            p = (h[n - 1] - q * h[n - 3]) / h[n - 2];
        }
        if (n == 2) return At::Quadratic;
        r = 0.0;
        return At::Iterate;
    }
    At iterate() {
        for (int it = max_iter; it > 0; --it) {
            for (int j = 0; j <= n; ++j) {
                b[j] = h[j] - p * b[j - 1] - q * b[j - 2];
                c[j] = b[j] - p * c[j - 1] - q * c[j - 2];
            }
            if (h[n - 1] != 0.0 && b[n - 1] != 0.0) {
                if (std::fabs(h[n - 1] / b[n - 1]) >= K) b[n] = h[n] - q * b[n - 2];  // guards the lost significance
                if (b[n] == 0.0) return At::Quadratic;
                if (K < std::fabs(h[n] / b[n])) return At::Quadratic;
            }
            for (int j = 0; j <= n; ++j) {
                d[j] = h[j] + r * d[j - 1];
                e[j] = d[j] + r * e[j - 1];
            }
            if (d[n] == 0.0) return At::Linear;
            if (K < std::fabs(h[n] / d[n])) return At::Linear;
            c[n - 1] = -p * c[n - 2] - q * c[n - 3];
            const double s = c[n - 2] * c[n - 2] - c[n - 1] * c[n - 3];
            if (s == 0.0) {
                p -= 2.0;
                q *= (q + 1.0);
            } else {
                p += (b[n - 1] * c[n - 2] - b[n] * c[n - 3]) / s;
                q += (-b[n - 1] * c[n - 1] + b[n] * c[n - 2]) / s;
            }
            if (e[n - 1] == 0.0) r -= 1.0;
            else r -= d[n] / e[n - 1];
        }
        p_prev = p_last;
        q_prev = q_last;
        p_last = p;
        q_last = q;
        if (flip < 0.0) K /= 10.0;
        if (K < 1e-8) return At::GiveUp;
        flip = -flip;
        return At::Reverse;
    }
    At linear() {
        if (dir < 0) r = 1.0 / r;
        --n;
        out.emplace_back(r, 0.0);
        deflate(h, d, n, K);
        return n == 0 ? At::Done : At::Iterate;
    }
    At quadratic() {
        if (dir < 0) {
            p /= q;
            q = 1.0 / q;
        }
        n -= 2;
        if (0.0 < (q - (p * p / 4.0))) {
            const double s = std::sqrt(q - (p * p / 4.0));
            if (!only_real) {
                out.emplace_back(-p / 2.0, s);
                out.emplace_back(-p / 2.0, -s);
            }
        } else {
            const double s = std::sqrt(((p * p / 4.0)) - q);
            const double big = p < 0.0 ? -p / 2.0 + s : -p / 2.0 - s;
            out.emplace_back(big, 0.0);
            out.emplace_back(q / big, 0.0);
        }
        deflate(h, b, n, K);
        return At::Init;
    }

public:
    Acm30(const std::vector<double>& f, bool only_real_, int max_iter_, double tol,
          std::vector<std::complex<double>>& out_)
            : max_iter{max_iter_}, only_real{only_real_}, out{out_} {
        n = (int)f.size() - 1;
        const size_t len = n + 3;
        m_store.assign(len * 5, 0.0);
        h = m_store.data() + 2;
        b = h + len;
        c = b + len;
        d = c + len;
        e = d + len;
        for (int j = 0; j <= n; ++j) h[n - j] = f[j];
        K = 1.0 / tol;
        while (h[n] == 0.0) {
            --n;
            sanm_check(n > -2, "roots: zero polynomial");
        }
    }
    bool run() {
        At at = At::Init;
        for (;;) {
            switch (at) {
                case At::Init: at = init(); break;
                case At::Reverse: reverse(); at = At::Start; break;
                case At::Start: at = start(); break;
                case At::Iterate: at = iterate(); break;
                case At::Linear: at = linear(); break;
                case At::Quadratic: at = quadratic(); break;
                case At::Done: return true;
                case At::GiveUp: return false;
            }
        }
    }
};
}  // namespace

bool roots(const std::vector<double>& f, bool only_real, std::vector<std::complex<double>>& out, int max_iter,
           double tol) {
    sanm_check(f.size() >= 2, "roots: need at least two coefficients");
    out.clear();
    return Acm30(f, only_real, max_iter, tol, out).run();
}

bool real_roots(const std::vector<double>& f, std::vector<double>& real) {
    std::vector<std::complex<double>> z;
    real.clear();
    if (!roots(f, true, z)) return false;
    for (auto& v : z) real.push_back(v.real());
    return true;
}

}  // namespace poly
}  // namespace sanm_hip
