// host_gamma.hpp -- plan-time (host) helpers for MovingThreshold plans: the regularised incomplete gamma function, its
// inverse in x, and the fit of ln P^-1(k; p) in ln k that the kernels use as the start value of their per-parcel
// inversion (kernels.hpp, moving_threshold).  Host only; the device routines are in device_math.hpp.
#pragma once
#include <cmath>
#include <vector>

namespace cloudy {
namespace gamma_host {

// P(a, x): series for x <= a + 1, Legendre's continued fraction (modified Lentz) above
inline double inc_gamma_p(double a, double x) {
    if (!(x > 0.0)) return 0.0;
    const double lnpre = a * std::log(x) - x - std::lgamma(a + 1.0);  // ln(x^a e^-x / Gamma(a+1))
    if (x <= a + 1.0) {
        double term = 1.0, sum = 1.0;
        for (int n = 1; n < 100000; ++n) {
            term *= x / (a + n);
            sum += term;
            if (term < 1e-17 * sum) break;
        }
        return std::exp(lnpre) * sum;
    }
    double b = x + 1.0 - a, c = 1e300, d = 1.0 / b, h = d;
    for (int i = 1; i < 100000; ++i) {
        const double an = -i * (i - a);
        b += 2.0;
        d = an * d + b;
        if (std::fabs(d) < 1e-300) d = 1e-300;
        c = b + an / c;
        if (std::fabs(c) < 1e-300) c = 1e-300;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (std::fabs(del - 1.0) < 1e-16) break;
    }
    return 1.0 - std::exp(lnpre + std::log(a)) * h;  // Q = x^a e^-x / Gamma(a) * h
}

// ln x with P(a, x) = p, 0 < p < 1: bisection in ln x (P is increasing in x), then secant polish
inline double ln_inc_gamma_inv(double a, double p) {
    double lo = -745.0, hi = std::log(a + 40.0 * std::sqrt(a) + 200.0);
    for (int it = 0; it < 200 && hi - lo > 1e-13; ++it) {
        const double mid = 0.5 * (lo + hi);
        (inc_gamma_p(a, std::exp(mid)) < p ? lo : hi) = mid;
    }
    return 0.5 * (lo + hi);
}

// Coefficients (highest power first, kTerms of them) of the polynomial in t = map0 + map1 ln k that fits
// ln P^-1(k; p) over k in [k_lo, k_hi]; returns the largest error of exp(fit) relative to P^-1 on a test grid.
inline double fit_inverse(double p, double k_lo, double k_hi, int n_terms, double &map0, double &map1, double *coef) {
    const int deg = n_terms - 1, np = n_terms;
    const double l0 = std::log(k_lo), l1 = std::log(k_hi);
    map1 = 2.0 / (l1 - l0);
    map0 = -1.0 - l0 * map1;
    std::vector<long double> cheb(np), f(np);
    for (int j = 0; j < np; ++j) {
        const double t = std::cos(M_PI * (j + 0.5) / np);
        f[j] = ln_inc_gamma_inv(std::exp((t - map0) / map1), p);
    }
    for (int m = 0; m < np; ++m) {
        long double s = 0.0L;
        for (int j = 0; j < np; ++j) s += f[j] * cosl((long double)M_PI * m * (j + 0.5L) / np);
        cheb[m] = s * (m == 0 ? 1.0L : 2.0L) / np;
    }
    std::vector<std::vector<long double>> T(np, std::vector<long double>(np, 0.0L));
    T[0][0] = 1.0L;
    if (np > 1) T[1][1] = 1.0L;
    for (int m = 2; m < np; ++m)
        for (int q = 0; q < np; ++q) T[m][q] = (q > 0 ? 2.0L * T[m - 1][q - 1] : 0.0L) - T[m - 2][q];
    for (int q = 0; q < np; ++q) {
        long double s = 0.0L;
        for (int m = 0; m < np; ++m) s += cheb[m] * T[m][q];
        coef[deg - q] = (double)s;
    }
    double worst = 0.0;
    for (int i = 0; i <= 200; ++i) {
        const double lk = l0 + (l1 - l0) * i / 200.0, t = map0 + map1 * lk;
        double y = coef[0];
        for (int d = 1; d < n_terms; ++d) y = std::fma(y, t, coef[d]);
        worst = std::fmax(worst, std::fabs(std::expm1(y - ln_inc_gamma_inv(std::exp(lk), p))));
    }
    return worst;
}

}  // namespace gamma_host
}  // namespace cloudy
