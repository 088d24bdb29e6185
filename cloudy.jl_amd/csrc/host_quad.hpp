// host_quad.hpp -- plan-time (host) construction of the start-value table of the per-parcel Gauss rules (quad.hpp).
//
// For a NumericalCoalStyle plan the library stages, once, polynomials in t = 2k/k_hi - 1 that give every node u_a(k)
// of the nq-point generalised Gauss-Laguerre rule for the weight u^(k-1) e^-u (row 0: u_1 / k) to ~1e-5 of the node
// spacing over k in (0, k_hi], plus the Gauss-Hermite rule for Lognormal modes.  The exact nodes needed for the fit
// come from the Jacobi matrix of the Laguerre recurrence (diag 2i + k, squared off-diagonal (i+1)(i+k)): eigenvalue i
// by Sturm-sequence bisection, then Newton on L_nq^(k-1).
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "quad.hpp"

namespace cloudy {
namespace quad_host {

inline int sturm_below(int n, const double *d, const double *e2, double x) {
    int c = 0;
    double q = d[0] - x;
    c += q < 0;
    for (int i = 1; i < n; ++i) {
        q = d[i] - x - e2[i - 1] / (q == 0.0 ? 1e-300 : q);
        c += q < 0;
    }
    return c;
}

inline void laguerre_nodes(int nq, double k, double *u) {
    std::vector<double> d(nq), e2(nq);
    for (int i = 0; i < nq; ++i) {
        d[i] = 2.0 * i + k;
        e2[i] = (i + 1.0) * (i + k);
    }
    const double top = 4.0 * nq + 2.0 * k + 8.0;
    for (int i = 0; i < nq; ++i) {
        double lo = 0.0, hi = top;
        for (int it = 0; it < 120 && hi - lo > 1e-12 * hi; ++it) {
            const double mid = 0.5 * (lo + hi);
            (sturm_below(nq, d.data(), e2.data(), mid) <= i ? lo : hi) = mid;
        }
        double x = 0.5 * (lo + hi);
        for (int it = 0; it < 6; ++it) {
            double Ln, Lm;
            laguerre_pair<0>(nq, k, x, Ln, Lm);
            const double D = nq * Ln - (nq - 1.0 + k) * Lm;
            const double xn = x - x * Ln / D;
            if (!(xn > 0.0)) break;
            x = xn;
        }
        u[i] = x;
    }
}

inline void hermite_rule(int nq, double *t, double *W) {
    std::vector<double> d(nq, 0.0), e2(nq);
    for (int i = 0; i < nq; ++i) e2[i] = 0.5 * (i + 1.0);
    const double bound = std::sqrt(2.0 * nq + 1.0) + 1.0;
    auto eval = [&](double x, double &hn, double &hm) {  // orthonormal Hermite h_nq, h_{nq-1}
        double p0 = 0.0, p1 = 0.7511255444649425;
        for (int j = 0; j < nq; ++j) {
            const double p2 = x * std::sqrt(2.0 / (j + 1.0)) * p1 - std::sqrt(double(j) / (j + 1.0)) * p0;
            p0 = p1;
            p1 = p2;
        }
        hn = p1;
        hm = p0;
    };
    for (int i = 0; i < nq; ++i) {
        double lo = -bound, hi = bound;
        for (int it = 0; it < 120 && hi - lo > 1e-13; ++it) {
            const double mid = 0.5 * (lo + hi);
            (sturm_below(nq, d.data(), e2.data(), mid) <= i ? lo : hi) = mid;
        }
        double x = 0.5 * (lo + hi), hn, hm;
        for (int it = 0; it < 6; ++it) {
            eval(x, hn, hm);
            x -= hn / (std::sqrt(2.0 * nq) * hm);
        }
        eval(x, hn, hm);
        t[i] = x;
        W[i] = 1.0 / (nq * hm * hm) / std::sqrt(M_PI);
    }
}

// q-point Gauss-Legendre rule on [-1, 1] (converged mode, quad_conv.hpp): Newton on P_q from the Chebyshev-like start values
inline void legendre_rule(int q, double *x, double *w) {
    for (int i = 0; i < q; ++i) {
        double t = std::cos(M_PI * (i + 0.75) / (q + 0.5)), dp = 1.0;
        for (int it = 0; it < 100; ++it) {
            double p0 = 1.0, p1 = t;
            for (int j = 2; j <= q; ++j) {
                const double p2 = ((2.0 * j - 1.0) * t * p1 - (j - 1.0) * p0) / j;
                p0 = p1;
                p1 = p2;
            }
            dp = q * (t * p1 - p0) / (t * t - 1.0);
            const double dt = p1 / dp;
            t -= dt;
            if (std::fabs(dt) < 1e-16) break;
        }
        x[q - 1 - i] = t;
        w[q - 1 - i] = 2.0 / ((1.0 - t * t) * dp * dp);
    }
}

// Builds the table for (nq, k_hi).  Raises the degree until the start values are within `tol` of the node spacing on
// a test grid; false (with a message) if kQuadDegMax does not reach it.
inline bool build_table(int nq, double k_hi, QArgs &Q, std::vector<double> &tab, std::string &msg) {
    const double tol = 1e-4;  // two Newton steps then leave < 1e-15 of the spacing (error_{n+1} ~ 3 error_n^2 / spacing)
    std::vector<double> u(nq), ut(nq);
    std::vector<double> ktest;
    for (int i = 0; i < 24; ++i) ktest.push_back(k_hi * std::pow(10.0, -16.0 + i * (16.0 / 24.0)));
    for (int i = 1; i <= 96; ++i) ktest.push_back(k_hi * i / 96.0);
    for (int deg = 8; deg <= kQuadDegMax; deg += 2) {
        const int np = deg + 1;
        // Chebyshev coefficients of f_a(t) from samples at the Chebyshev points
        std::vector<long double> cheb((size_t)nq * np, 0.0L);
        std::vector<double> tj(np);
        std::vector<std::vector<double>> f(np, std::vector<double>(nq));
        for (int j = 0; j < np; ++j) {
            tj[j] = std::cos(M_PI * (j + 0.5) / np);
            const double k = 0.5 * (tj[j] + 1.0) * k_hi;
            laguerre_nodes(nq, k, f[j].data());
            f[j][0] /= k;
        }
        for (int a = 0; a < nq; ++a)
            for (int m = 0; m < np; ++m) {
                long double s = 0.0L;
                for (int j = 0; j < np; ++j) s += (long double)f[j][a] * cosl((long double)M_PI * m * (j + 0.5L) / np);
                cheb[(size_t)a * np + m] = s * (m == 0 ? 1.0L : 2.0L) / np;
            }
        // to the monomial basis (extended precision: the conversion amplifies rounding by ~2^deg)
        std::vector<std::vector<long double>> T(np, std::vector<long double>(np, 0.0L));  // T[m][p]: coefficient of t^p in T_m
        T[0][0] = 1.0L;
        if (np > 1) T[1][1] = 1.0L;
        for (int m = 2; m < np; ++m)
            for (int p = 0; p < np; ++p) T[m][p] = (p > 0 ? 2.0L * T[m - 1][p - 1] : 0.0L) - T[m - 2][p];
        tab.assign((size_t)quad_tab_size(nq, deg), 0.0);
        for (int a = 0; a < nq; ++a)
            for (int p = 0; p < np; ++p) {
                long double s = 0.0L;
                for (int m = 0; m < np; ++m) s += cheb[(size_t)a * np + m] * T[m][p];
                tab[(size_t)a * np + (deg - p)] = (double)s;  // highest power first
            }
        // test: the Horner start values (in double, as the device evaluates them) against the nodes
        double worst = 0.0;
        for (double k : ktest) {
            laguerre_nodes(nq, k, u.data());
            const double t = k * (2.0 / k_hi) - 1.0;
            for (int a = 0; a < nq; ++a) {
                const double *c = &tab[(size_t)a * np];
                double x = c[0];
                for (int d = 1; d <= deg; ++d) x = std::fma(x, t, c[d]);
                if (a == 0) x *= k;
                const double left = a ? u[a] - u[a - 1] : u[0], right = a + 1 < nq ? u[a + 1] - u[a] : left;
                worst = std::fmax(worst, std::fabs(x - u[a]) / std::fmin(left, right));
            }
        }
        if (worst <= tol) {
            Q.nq = nq;
            Q.deg = deg;
            Q.t_scale = 2.0 / k_hi;
            hermite_rule(nq, &tab[(size_t)nq * np], &tab[(size_t)nq * np + nq]);
            return true;
        }
        msg = "start-value table for nq = " + std::to_string(nq) + ", k <= " + std::to_string(k_hi) + ": error " +
              std::to_string(worst) + " of the node spacing at degree " + std::to_string(deg);
    }
    return false;
}

}  // namespace quad_host
}  // namespace cloudy
