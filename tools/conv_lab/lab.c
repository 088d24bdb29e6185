/* conv_lab: CPU experiments with the T_m rule of converged mode (Gamma-family mode j, homogeneous kernel; the other modes
 * Gamma-family or Lognormal).  Not product code, not the oracle: a sandbox to count integrand evaluations of candidate rules
 * before they are written into csrc/quad_conv.hpp and oracle/cloudy_oracle_quad.c.
 * Build: gcc -O2 -shared -fPIC -o liblab.so lab.c -lm */
#include <math.h>
#include <string.h>
#define MAXN 4
typedef struct {
    long nodes, edges, evals, rejects, skipped, init_panels;
} lab_stats;

static const double GKX[15] = {-0.991455371120812639206854697526329, -0.949107912342758524526189684047851,
                               -0.864864423359769072789712788640926, -0.741531185599394439863864773280788,
                               -0.586087235467691130294144838258730, -0.405845151377397166906606412076961,
                               -0.207784955007898467600689403773245, 0.0,
                               0.207784955007898467600689403773245,  0.405845151377397166906606412076961,
                               0.586087235467691130294144838258730,  0.741531185599394439863864773280788,
                               0.864864423359769072789712788640926,  0.949107912342758524526189684047851,
                               0.991455371120812639206854697526329};
static const double GKWK[15] = {0.022935322010529224963732008058970, 0.063092092629978553290700663189204,
                                0.104790010322250183839876322541518, 0.140653259715525918745189590510238,
                                0.169004726639267902826583426598550, 0.190350578064785409913256402421014,
                                0.204432940075298892414161999234649, 0.209482141084727828012999174891714,
                                0.204432940075298892414161999234649, 0.190350578064785409913256402421014,
                                0.169004726639267902826583426598550, 0.140653259715525918745189590510238,
                                0.104790010322250183839876322541518, 0.063092092629978553290700663189204,
                                0.022935322010529224963732008058970};
static const double GKWG[15] = {0.0, 0.129484966168869693270611432679082, 0.0, 0.279705391489276667901467771423780,
                                0.0, 0.381830050505118944950369775488975, 0.0, 0.417959183673469387755102040816327,
                                0.0, 0.381830050505118944950369775488975, 0.0, 0.279705391489276667901467771423780,
                                0.0, 0.129484966168869693270611432679082, 0.0};

/* ln rho_m(t) = ln f_m(s) - ln f_j(s) at s = theta_j e^t:  (q2 t + q1) t + q0 + cb e^t */
typedef struct {
    int N, j;
    double A, lgA, thj, lthj;
    double q2[MAXN], q1[MAXN], q0[MAXN], cb[MAXN];
    double lnmean[MAXN], cwidth[MAXN];
} rule;
/* type[m]: 0 Gamma family (th = theta, k = shape), 1 Lognormal (th = mu, k = sigma) */
static void rule_init(rule *r, int N, int j, const int *type, const double *th, const double *k, double gam) {
    r->N = N;
    r->j = j;
    r->A = 2.0 * k[j] + gam;
    r->lgA = lgamma(r->A);
    r->thj = th[j];
    r->lthj = log(th[j]);
    const double cj = lgamma(k[j]) + k[j] * r->lthj;
    for (int m = 0; m < N; ++m) {
        if (type && type[m] == 1) {
            const double b = 0.5 / (k[m] * k[m]), dl = r->lthj - th[m], cL = log(k[m] * 2.5066282746310002);
            r->q2[m] = -b;
            r->q1[m] = -2.0 * b * dl - k[j];
            r->q0[m] = -b * dl * dl - k[j] * r->lthj - cL + cj;
            r->cb[m] = 1.0;
            r->lnmean[m] = th[m] + 0.5 * k[m] * k[m];
            r->cwidth[m] = k[m];
        } else {
            const double da = k[m] - k[j], dc = lgamma(k[m]) + k[m] * log(th[m]) - cj;
            r->q2[m] = 0.0;
            r->q1[m] = da;
            r->q0[m] = da * r->lthj - dc;
            r->cb[m] = (1.0 / th[j] - 1.0 / th[m]) * th[j];
            r->lnmean[m] = log(k[m] * th[m]);
            r->cwidth[m] = 1.0 / sqrt(fmax(k[m], 1.0));
        }
    }
}
static double lrho(const rule *r, int m, double t, double u) { return (r->q2[m] * t + r->q1[m]) * t + r->q0[m] + r->cb[m] * u; }
static void node(const rule *r, double t, double *vals) {
    const double u = exp(t), wt = exp(r->A * t - u - r->lgA);
    double up = 0.0, den = 1.0;
    for (int m = 0; m < r->N; ++m) {
        if (m == r->j) continue;
        const double rho = exp(fmin(lrho(r, m, t, u), 700.0));
        den += rho;
        if (m > r->j) up += rho;
    }
    const double s = u * r->thj, h = wt * up / den;
    vals[0] = h;
    vals[1] = h * s;
    vals[2] = h * s * s;
}
static void range_of(double A, double range_eps, double *tlo, double *thi) {
    *tlo = fmax(-690.0, fmin(-1.0, (range_eps + lgamma(A + 1.0)) / A));
    const double Am = A + 2.0;
    *thi = log(Am + sqrt(60.0 * Am) + 30.0);
}
static double below_tlo(const rule *r, double tlo) {
    const double u_lo = exp(tlo);
    double up = 0.0, den = 1.0;
    for (int m = 0; m < r->N; ++m) {
        if (m == r->j) continue;
        const double rho = exp(fmin(lrho(r, m, tlo, u_lo), 700.0));
        den += rho;
        if (m > r->j) up += rho;
    }
    return up / den * exp(r->A * tlo - lgamma(r->A + 1.0));
}

/* ---------------- the current rule: equal pieces + graded marks, GK(7,15) bisection ---------------- */
typedef struct {
    int ninit;
    double tol;
    int marks, lmax;
    double floor_, range_eps;
    int imax;
    int desc;        /* 1: walk the initial panels from thi down to tlo */
    double tol_skip; /* > 0: skip an initial panel whose rigorous bound is below tol_skip * max(|acc|, floor * scale) */
    int est;         /* 0: |K15 - G7|; 1: QUADPACK qk15's estimate |K - G| min(1, (200 |K - G| / resasc)^1.5) */
    double est_pow, est_fac;
    int xmarks;      /* round 5: 1 = ALSO graded marks around the points where rho_m = 1 (the transitions of weighting_fn); 2 = those INSTEAD of the core marks where a crossing exists */
} lab_params;

int lab_T_rule(int N, int j, const int *type, const double *th, const double *k, double gam, const lab_params *P, double *T,
               lab_stats *st) {
    rule r;
    rule_init(&r, N, j, type, th, k, gam);
    const double A = r.A;
    double tlo, thi;
    range_of(A, P->range_eps, &tlo, &thi);
    const double scaleS[3] = {1.0, A * r.thj, A * (A + 1.0) * r.thj * r.thj};
    const double h0 = (thi - tlo) / P->ninit, gap = 1e-7 * (thi - tlo);
    double marks[256];
    int nm = 0;
    if (P->marks)
        for (int m = 0; m < N; ++m) {
            if (m == j) continue;
            /* crossings lrho_m(t) = 0 in [tlo, thi]: scan + bisection (sandbox: robust, not cheap) */
            int nx = 0;
            double xr[4], xw[4];
            if (P->xmarks) {
                const int NS = 400;
                double tp = tlo, fp = lrho(&r, m, tp, exp(tp));
                for (int i = 1; i <= NS && nx < 4; ++i) {
                    const double tc = tlo + (thi - tlo) * i / NS, fc = lrho(&r, m, tc, exp(tc));
                    if ((fp > 0) != (fc > 0)) {
                        double a = tp, b = tc, fa = fp;
                        for (int it = 0; it < 60; ++it) {
                            const double mid = 0.5 * (a + b), fm = lrho(&r, m, mid, exp(mid));
                            if ((fm > 0) == (fa > 0)) { a = mid; fa = fm; } else b = mid;
                        }
                        const double t0 = 0.5 * (a + b), sl = fabs(2.0 * r.q2[m] * t0 + r.q1[m] + r.cb[m] * exp(t0));
                        xr[nx] = t0;
                        xw[nx] = sl > 0 ? 1.0 / sl : 1.0;
                        ++nx;
                    }
                    tp = tc; fp = fc;
                }
                for (int x = 0; x < nx; ++x) {
                    const double ratio = h0 / xw[x];
                    int I = 0;
                    if (ratio > 1.0) I = ratio < 4096.0 ? (int)ceil(log2(ratio)) : P->imax;
                    if (I > P->imax) I = P->imax;
                    for (int q = -(I + 1); q <= I + 1 && nm < 120; ++q) {
                        const double off = q == 0 ? 0.0 : (q < 0 ? -1.0 : 1.0) * ldexp(xw[x], (q < 0 ? -q : q) - 1);
                        marks[nm++] = xr[x] + off;
                    }
                }
                if (P->xmarks == 2 && nx > 0) continue;
            }
            const double c = r.lnmean[m], w = r.cwidth[m];
            const double ratio = h0 / w;
            int I = 0;
            if (ratio > 1.0) I = ratio < 4096.0 ? (int)ceil(log2(ratio)) : P->imax;
            if (I > P->imax) I = P->imax;
            for (int q = -(I + 1); q <= I + 1; ++q) {
                const double off = q == 0 ? 0.0 : (q < 0 ? -1.0 : 1.0) * ldexp(w, (q < 0 ? -q : q) - 1);
                marks[nm++] = c + off - r.lthj;
            }
        }
    double out[3] = {0, 0, 0};
    /* the sorted list of initial edges (same set in either direction) */
    double edges[256];
    int ne = 0;
    {
        double cur = tlo;
        int io = 1;
        edges[ne++] = tlo;
        while (cur < thi) {
            double nxt = thi, own = tlo + h0 * io;
            while (own <= cur + gap) {
                ++io;
                own = tlo + h0 * io;
            }
            if (own < nxt) nxt = own;
            for (int m = 0; m < nm; ++m)
                if (marks[m] > cur + gap && marks[m] < nxt) nxt = marks[m];
            if (nxt > thi - gap) nxt = thi;
            cur = nxt;
            edges[ne++] = nxt;
        }
    }
    /* stationary points of concave Gamma-family log ratios */
    double tst[MAXN];
    for (int m = 0; m < N; ++m) tst[m] = (r.q2[m] == 0.0 && r.cb[m] < 0.0 && r.q1[m] > 0.0) ? log(-r.q1[m] / r.cb[m]) : -INFINITY;
    const double tmode = log(A);
    for (int pi = 0; pi < ne - 1; ++pi) {
        const int p = P->desc ? ne - 2 - pi : pi;
        const double a0 = edges[p], b0 = edges[p + 1], h = b0 - a0;
        st->init_panels++;
        if (P->tol_skip > 0.0) {
            const double ua = exp(a0), ub = exp(b0);
            st->edges++;
            double lw = fmax(A * a0 - ua, A * b0 - ub);
            if (tmode > a0 && tmode < b0) lw = A * tmode - A;
            double sup = 0.0;
            for (int m = j + 1; m < N; ++m) {
                double l;
                if (r.q2[m] != 0.0) { /* Lognormal: sup of the quadratic + u(b) */
                    double tv = -r.q1[m] / (2.0 * r.q2[m]);
                    tv = fmin(fmax(tv, a0), b0);
                    l = (r.q2[m] * tv + r.q1[m]) * tv + r.q0[m] + r.cb[m] * ub;
                } else if (r.cb[m] >= 0.0) {
                    l = fmax(lrho(&r, m, a0, ua), lrho(&r, m, b0, ub));
                } else {
                    const double tc = fmin(fmax(tst[m], a0), b0);
                    l = lrho(&r, m, tc, exp(tc));
                }
                sup += exp(fmin(l, 0.0));
            }
            sup = fmin(sup, 1.0);
            double B = exp(lw - r.lgA) * sup * h;
            const double sb = ub * r.thj;
            int skip = 1;
            for (int o = 0; o < 3; ++o) {
                if (!(B <= P->tol_skip * fmax(fabs(out[o]), P->floor_ * scaleS[o]))) skip = 0;
                B *= sb;
            }
            if (skip) {
                st->skipped++;
                continue;
            }
        }
        if (P->tol_skip < 0.0 && P->desc) {
            /* termination only: the bound over the WHOLE remaining range [tlo, b0] */
            const double ts = -P->tol_skip, ua = exp(tlo), ub = exp(b0);
            st->edges++;
            double lw = fmax(A * tlo - ua, A * b0 - ub);
            if (tmode > tlo && tmode < b0) lw = A * tmode - A;
            double sup = 0.0;
            for (int m = j + 1; m < N; ++m) {
                double l;
                if (r.q2[m] != 0.0) {
                    double tv = -r.q1[m] / (2.0 * r.q2[m]);
                    tv = fmin(fmax(tv, tlo), b0);
                    l = (r.q2[m] * tv + r.q1[m]) * tv + r.q0[m] + r.cb[m] * ub;
                } else if (r.cb[m] >= 0.0) {
                    l = fmax(lrho(&r, m, tlo, ua), lrho(&r, m, b0, ub));
                } else {
                    const double tc = fmin(fmax(tst[m], tlo), b0);
                    l = lrho(&r, m, tc, exp(tc));
                }
                sup += exp(fmin(l, 0.0));
            }
            sup = fmin(sup, 1.0);
            double B = exp(lw - r.lgA) * sup * (b0 - tlo);
            const double sb = ub * r.thj;
            int stop = 1;
            for (int o = 0; o < 3; ++o) {
                if (!(B <= ts * fmax(fabs(out[o]), P->floor_ * scaleS[o]))) stop = 0;
                B *= sb;
            }
            if (stop) {
                st->skipped += ne - 1 - pi;
                break;
            }
        }
        int L = 0;
        unsigned i = 0;
        for (;;) {
            const double w = ldexp(h, -L), hw = 0.5 * w, c = (a0 + w * i) + hw;
            double K[3] = {0, 0, 0}, G[3] = {0, 0, 0}, vals[3], fv[15][3];
            for (int g = 0; g < 15; ++g) {
                node(&r, c + hw * GKX[g], vals);
                st->nodes++;
                for (int o = 0; o < 3; ++o) {
                    fv[g][o] = vals[o];
                    K[o] += GKWK[g] * vals[o];
                    if (g & 1) G[o] += GKWG[g] * vals[o];
                }
            }
            st->evals++;
            int ok = 1;
            for (int o = 0; o < 3; ++o) {
                double e = fabs(K[o] - G[o]);
                if (P->est) {
                    double ra = 0.0;
                    for (int g = 0; g < 15; ++g) ra += GKWK[g] * fabs(fv[g][o] - 0.5 * K[o]);
                    if (ra > 0.0 && e > 0.0) e *= fmin(1.0, pow(P->est_fac * e / ra, P->est_pow));
                }
                if (e * hw > P->tol * fmax(fabs(out[o] + K[o] * hw), P->floor_ * scaleS[o])) ok = 0;
            }
            if (ok || L == P->lmax) {
                for (int o = 0; o < 3; ++o) out[o] += K[o] * hw;
                ++i;
                while (L > 0 && !(i & 1u)) {
                    i >>= 1;
                    --L;
                }
                if (L == 0) break;
            } else {
                st->rejects++;
                ++L;
                i <<= 1;
            }
        }
    }
    out[0] += below_tlo(&r, tlo);
    for (int o = 0; o < 3; ++o) T[o] = out[o];
    return 0;
}

/* ---------------- graded marching layout ---------------- */
typedef struct {
    double c_step;   /* w <= c_step * d (pole-distance model) */
    double c_exp;    /* w <= 2 c_exp / S (log-slope of the integrand) */
    double wmax;
    double win;      /* log-density window of the terms that count */
    int rule;        /* 15: K15 with |K-G| net; else Gauss-Legendre with that many points and the null-rule net */
    double tol_net;  /* bisect when the estimate exceeds tol_net * max(|acc|, floor scale) */
    double floor_;
    double range_eps;
    double skip_tol; /* skip a panel whose a-priori bound is below skip_tol * max(|acc|, floor * scale) (0: off) */
    int lmax;
    int topdown;     /* 1: march from thi down to tlo */
} grad_params;

static void gl_rule(int q, double *x, double *w) {
    for (int i = 0; i < q; ++i) {
        double t = cos(M_PI * (i + 0.75) / (q + 0.5)), dp = 1.0;
        for (int it = 0; it < 100; ++it) {
            double p0 = 1.0, p1 = t;
            for (int jj = 2; jj <= q; ++jj) {
                const double p2 = ((2.0 * jj - 1.0) * t * p1 - (jj - 1.0) * p0) / jj;
                p0 = p1;
                p1 = p2;
            }
            dp = q * (t * p1 - p0) / (t * t - 1.0);
            const double dt = p1 / dp;
            t -= dt;
            if (fabs(dt) < 1e-16) break;
        }
        x[q - 1 - i] = t;
        w[q - 1 - i] = 2.0 / ((1.0 - t * t) * dp * dp);
    }
}
static double legendre(int n, double x) {
    double p0 = 1.0, p1 = x;
    if (n == 0) return 1.0;
    for (int jj = 2; jj <= n; ++jj) {
        const double p2 = ((2.0 * jj - 1.0) * x * p1 - (jj - 1.0) * p0) / jj;
        p0 = p1;
        p1 = p2;
    }
    return p1;
}

typedef struct {
    double t, u, L[MAXN], lam[MAXN], Lmax;
} edge;
static void edge_eval(const rule *r, double t, edge *e) {
    e->t = t;
    e->u = exp(t);
    e->Lmax = 0.0;
    for (int m = 0; m < r->N; ++m) {
        if (m == r->j) {
            e->L[m] = 0.0;
            e->lam[m] = 0.0;
        } else {
            e->L[m] = lrho(r, m, t, e->u);
            e->lam[m] = fabs(2.0 * r->q2[m] * t + r->q1[m]) + 1.72 * fabs(r->cb[m]) * e->u + sqrt(M_PI * fabs(r->q2[m]));
        }
        if (e->L[m] > e->Lmax) e->Lmax = e->L[m];
    }
}
/* model over a panel from its two edges: pole distance and log-slope */
static void panel_model(const rule *r, const edge *a, const edge *b, double win, double *d, double *S) {
    double dd = 1e300, lamq = 0.0;
    const int N = r->N;
    for (int i = 0; i < N; ++i) {
        const int qi = a->L[i] >= a->Lmax - win || b->L[i] >= b->Lmax - win;
        if (!qi) continue;
        lamq = fmax(lamq, fmax(a->lam[i], b->lam[i]));
        for (int k = i + 1; k < N; ++k) {
            const int qk = a->L[k] >= a->Lmax - win || b->L[k] >= b->Lmax - win;
            if (!qk) continue;
            const double da_ = a->L[i] - a->L[k], db_ = b->L[i] - b->L[k];
            const double Lmin = (da_ > 0.0) != (db_ > 0.0) ? 0.0 : fmin(fabs(da_), fabs(db_));
            const double lam = fmax(a->lam[i], b->lam[i]) + fmax(a->lam[k], b->lam[k]);
            if (lam > 0.0) dd = fmin(dd, sqrt(Lmin * Lmin + M_PI * M_PI) / lam);
        }
    }
    *d = dd;
    const double sa = fmax(fabs(r->A - a->u), fabs(r->A + 2.0 - a->u)), sb = fmax(fabs(r->A - b->u), fabs(r->A + 2.0 - b->u));
    *S = fmax(sa, sb) + lamq;
}

int lab_T_graded(int N, int j, const int *type, const double *th, const double *k, double gam, const grad_params *P, double *T,
                 lab_stats *st) {
    rule r;
    rule_init(&r, N, j, type, th, k, gam);
    const double A = r.A;
    double tlo, thi;
    range_of(A, P->range_eps, &tlo, &thi);
    const double scaleS[3] = {1.0, A * r.thj, A * (A + 1.0) * r.thj * r.thj};
    double glx[64], glw[64], n1[64], n2[64];
    if (P->rule != 15) {
        gl_rule(P->rule, glx, glw);
        for (int g = 0; g < P->rule; ++g) {
            n1[g] = glw[g] * legendre(P->rule - 1, glx[g]) * (2.0 * (P->rule - 1) + 1.0) * 0.5;
            n2[g] = glw[g] * legendre(P->rule - 2, glx[g]) * (2.0 * (P->rule - 2) + 1.0) * 0.5;
        }
    }
    double out[3] = {0, 0, 0};
    const double dir = P->topdown ? -1.0 : 1.0, t_start = P->topdown ? thi : tlo, t_end = P->topdown ? tlo : thi;
    edge ea, eb;
    edge_eval(&r, t_start, &ea);
    st->edges++;
    double cur = t_start;
    while (dir * (t_end - cur) > 0.0) {
        double d, S;
        panel_model(&r, &ea, &ea, P->win, &d, &S);
        double w = fmin(fmin(P->c_step * d, 2.0 * P->c_exp / S), P->wmax);
        for (int tries = 0;; ++tries) {
            double b = cur + dir * w;
            if (dir * (t_end - b) < 0.0) {
                b = t_end;
                w = fabs(b - cur);
            }
            edge_eval(&r, b, &eb);
            st->edges++;
            panel_model(&r, &ea, &eb, P->win, &d, &S);
            const double wl = fmin(fmin(P->c_step * d, 2.0 * P->c_exp / S), P->wmax);
            if (w <= wl * (1.0 + 1e-12) || tries >= 40) break;
            w = fmax(0.5 * w, wl);
        }
        const double a0 = P->topdown ? cur - w : cur, h = w;
        cur = (fabs(t_end - (cur + dir * w)) <= 1e-12 * (thi - tlo)) ? t_end : cur + dir * w;
        st->init_panels++;
        int skip = 0;
        if (P->skip_tol > 0.0) {
            double lup = -1e300;
            for (int m = j + 1; m < N; ++m) lup = fmax(lup, fmax(ea.L[m] - ea.Lmax, eb.L[m] - eb.Lmax));
            const double tm0 = log(A), t1 = fmin(ea.t, eb.t), t2 = fmax(ea.t, eb.t);
            double lw = fmax(A * ea.t - ea.u, A * eb.t - eb.u);
            if (tm0 > t1 && tm0 < t2) lw = A * tm0 - A;
            double B = exp(lw - r.lgA + fmin(lup, 0.0)) * (N - 1 - j) * h;
            const double sb = fmax(ea.u, eb.u) * r.thj;
            skip = 1;
            for (int o = 0; o < 3; ++o) {
                if (!(B <= P->skip_tol * fmax(fabs(out[o]), P->floor_ * scaleS[o]))) skip = 0;
                B *= sb;
            }
        }
        ea = eb;
        if (skip) {
            st->skipped++;
            continue;
        }
        int L = 0;
        unsigned i = 0;
        for (;;) {
            const double ww = ldexp(h, -L), hw = 0.5 * ww, c = (a0 + ww * i) + hw;
            double K[3] = {0, 0, 0}, G[3] = {0, 0, 0}, E1[3] = {0, 0, 0}, E2[3] = {0, 0, 0}, vals[3];
            int ok = 1;
            if (P->rule == 15) {
                for (int g = 0; g < 15; ++g) {
                    node(&r, c + hw * GKX[g], vals);
                    st->nodes++;
                    for (int o = 0; o < 3; ++o) {
                        K[o] += GKWK[g] * vals[o];
                        if (g & 1) G[o] += GKWG[g] * vals[o];
                    }
                }
                for (int o = 0; o < 3; ++o)
                    if (fabs(K[o] - G[o]) * hw > P->tol_net * fmax(fabs(out[o] + K[o] * hw), P->floor_ * scaleS[o])) ok = 0;
            } else {
                for (int g = 0; g < P->rule; ++g) {
                    node(&r, c + hw * glx[g], vals);
                    st->nodes++;
                    for (int o = 0; o < 3; ++o) {
                        K[o] += glw[g] * vals[o];
                        E1[o] += n1[g] * vals[o];
                        E2[o] += n2[g] * vals[o];
                    }
                }
                for (int o = 0; o < 3; ++o)
                    if ((fabs(E1[o]) + fabs(E2[o])) * hw > P->tol_net * fmax(fabs(out[o] + K[o] * hw), P->floor_ * scaleS[o])) ok = 0;
            }
            st->evals++;
            if (ok || L == P->lmax) {
                for (int o = 0; o < 3; ++o) out[o] += K[o] * hw;
                ++i;
                while (L > 0 && !(i & 1u)) {
                    i >>= 1;
                    --L;
                }
                if (L == 0) break;
            } else {
                st->rejects++;
                ++L;
                i <<= 1;
            }
        }
    }
    out[0] += below_tlo(&r, tlo);
    for (int o = 0; o < 3; ++o) T[o] = out[o];
    return 0;
}

/* batch: ntk[3N][n] planes of (n, theta, k) per mode; type[N]; T[(N-1)*3][n]; cost[(N-1)][n] = nodes + edge_cost * edges.
 * which: 0 = lab_T_rule (P = lab_params), 1 = lab_T_graded (P = grad_params) */
int lab_batch(int which, int N, const int *type, long n, const double *ntk, double gam, const void *P, double *T, double *cost,
              double edge_cost, lab_stats *tot) {
    memset(tot, 0, sizeof *tot);
    for (long p = 0; p < n; ++p) {
        double th[MAXN], k[MAXN], nn[MAXN];
        for (int m = 0; m < N; ++m) {
            nn[m] = ntk[(3 * m + 0) * n + p];
            th[m] = ntk[(3 * m + 1) * n + p];
            k[m] = ntk[(3 * m + 2) * n + p];
        }
        for (int j = 0; j < N - 1; ++j) {
            lab_stats st;
            memset(&st, 0, sizeof st);
            double t3[3] = {0, 0, 0};
            if (nn[j] > 0.0 && !(type && type[j] == 1)) {
                if (which == 0)
                    lab_T_rule(N, j, type, th, k, gam, (const lab_params *)P, t3, &st);
                else
                    lab_T_graded(N, j, type, th, k, gam, (const grad_params *)P, t3, &st);
            }
            for (int o = 0; o < 3; ++o) T[(3 * j + o) * n + p] = t3[o];
            cost[j * n + p] = edge_cost < 0.0 ? (double)st.init_panels : st.nodes + edge_cost * st.edges;
            tot->nodes += st.nodes;
            tot->edges += st.edges;
            tot->evals += st.evals;
            tot->rejects += st.rejects;
            tot->skipped += st.skipped;
            tot->init_panels += st.init_panels;
        }
    }
    return 0;
}
