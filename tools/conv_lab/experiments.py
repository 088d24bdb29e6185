"""The CPU experiments behind DESIGN.md (converged mode, rounds 4-5) in ONE parameterised script (the
20 one-off scan drivers grad*/qp*/trap*/walk*/wild*.py folded in).  Sandbox of the T_m rule: lab.c / lab.py.

    python tools/conv_lab/experiments.py --list
    python tools/conv_lab/experiments.py qp3 walk4 ...
"""
import math
import sys

import numpy as np

from lab import *  # noqa: F401,F403 -- params, gparams, run, report, batch_ntk, wild_ntk

EXPERIMENTS = {}


def experiment(name, what):
    def deco(f):
        EXPERIMENTS[name] = (f, what)
        return f
    return deco


@experiment("scan2", "initial pieces x tolerance of the upward walk on the cfg4q batch")
def scan2():
    ntk = batch_ntk(6400)
    ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
    for ni in (6, 8, 10, 12, 16):
        for tol in (1e-8, 1e-7, 1e-6):
            report(f"ninit {ni} tol {tol:g}", ntk, params(ninit=ni, tol=tol), ref)
    for ni in (8, 12):
        for tol in (1e-9, 1e-10):
          for ep in (1.5, 2.0):
            report(f"ninit {ni} tol {tol:g} est_pow {ep}", ntk, params(ninit=ni, tol=tol, est_pow=ep), ref)


@experiment("walk2", "ascending vs descending walk, per-panel skip")
def walk2():
    ntk = batch_ntk(6400)
    ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
    base = dict(ninit=12, tol=1e-8)
    report("now (12, 1e-8) ascending", ntk, params(**base), ref)
    report("descending, no skip", ntk, params(desc=1, **base), ref)
    for ts in (1e-12, 1e-11, 1e-10):
        report(f"ascending skip {ts:g}", ntk, params(tol_skip=ts, **base), ref)
        report(f"descending skip {ts:g}", ntk, params(desc=1, tol_skip=ts, **base), ref)


@experiment("walk3", "descending walk: pieces x tolerance at skip 1e-10")
def walk3():
    ntk = batch_ntk(6400)
    ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
    for ni in (8, 10, 12):
        for tol in (1e-8, 1e-7, 1e-6):
            report(f"desc ninit {ni} tol {tol:g} skip 1e-10", ntk, params(ninit=ni, tol=tol, desc=1, tol_skip=1e-10), ref)


@experiment("walk4", "descending walk: per-panel skip vs terminate-only bound")
def walk4():
    ntk = batch_ntk(6400)
    ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
    base = dict(ninit=12, tol=1e-8, desc=1)
    report("desc no skip", ntk, params(**base), ref)
    report("desc per-panel skip 1e-10", ntk, params(tol_skip=1e-10, **base), ref)
    for ts in (1e-11, 1e-10, 1e-9):
        report(f"desc terminate-only {ts:g}", ntk, params(tol_skip=-ts, **base), ref)


@experiment("wild1", "random multi-scale mixtures: current rule vs a-priori graded panels")
def wild1():
    for N in (2, 3):
        for ln in (False, True):
            for gam in (0.0, 4/3):
                ntk, types = wild_ntk(4000, N, seed=5 + N, lognormal_others=ln)
                ref, _, _ = run(ntk, params(ninit=64, tol=1e-14), N, gam=gam, types=types)
                tag = f"N{N} ln{int(ln)} gam{gam:.1f}"
                report(tag + " current", ntk, params(), ref, N, types, gam)
                report(tag + " K15 3.5/12", ntk, gparams(), ref, N, types, gam)
                report(tag + " K15 3.5/12 td skip1e-11", ntk, gparams(topdown=1, skip_tol=1e-11), ref, N, types, gam)
                report(tag + " K15 2.5/8 td skip1e-11", ntk, gparams(c_step=2.5, c_exp=8, topdown=1, skip_tol=1e-11), ref, N, types, gam)


@experiment("wild2", "random multi-scale mixtures: ascending vs descending settings")
def wild2():
    for N in (2, 3):
        for ln in (False, True):
            for gam in (0.0, 4/3):
                ntk, types = wild_ntk(3000, N, seed=5 + N, lognormal_others=ln)
                ref, _, _ = run(ntk, params(ninit=64, tol=1e-14), N, gam=gam, types=types)
                tag = f"N{N} ln{int(ln)} gam{gam:.1f}"
                report(tag + " asc 12/1e-8 (now)", ntk, params(ninit=12, tol=1e-8), ref, N, types, gam)
                for ni, tol, ts in ((12, 1e-8, 1e-10), (10, 1e-7, 1e-10), (8, 1e-7, 1e-10), (8, 1e-6, 1e-10), (10, 1e-7, 1e-11)):
                    report(tag + f" desc {ni}/{tol:g} skip {ts:g}", ntk, params(ninit=ni, tol=tol, desc=1, tol_skip=ts), ref, N, types, gam)


@experiment("grad3", "a-priori graded panels (pole-distance model) with K15 / Gauss-Legendre nets")
def grad3():
    ntk = batch_ntk(6400)
    ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
    report("current", ntk, params(), ref)
    report("K15 3.5/12", ntk, gparams(), ref)
    for rule in (10, 12):
        for cs, ce in ((2.5, 8.0), (3.0, 10.0), (3.5, 12.0), (5.0, 12.0)):
            for tn in (1e-4, 1e-5):
                report(f"GL{rule} {cs}/{ce} net {tn:g}", ntk, gparams(rule=rule, c_step=cs, c_exp=ce, tol_net=tn), ref)


@experiment("grad4", "graded panels: top-down march, skip bound, range epsilon")
def grad4():
    ntk = batch_ntk(6400)
    ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
    for td in (0, 1):
        for sk in (0.0, 1e-12, 1e-11, 1e-10):
            report(f"K15 3.5/12 topdown {td} skip {sk:g}", ntk, gparams(topdown=td, skip_tol=sk), ref)
    for eps in (1e-13, 1e-11):
        report(f"K15 3.5/12 range_eps {eps:g}", ntk, gparams(range_eps=float(np.log(eps))), ref)


@experiment("qp1", "QUADPACK qk15 error estimate, pieces x tolerance")
def qp1():
    ntk = batch_ntk(6400)
    ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
    report("now: desc 10/1e-7 term 1e-10", ntk, params(ninit=10, tol=1e-7, desc=1, tol_skip=-1e-10), ref)
    for ni in (4, 5, 6, 8, 10):
        for tol in (1e-8, 1e-9, 1e-10):
            report(f"qk15 estimate {ni}/{tol:g}", ntk, params(ninit=ni, tol=tol, desc=1, tol_skip=-1e-10, est=1), ref)


@experiment("qp2", "plain looser tolerances vs powers / factors of the qk15 estimate")
def qp2():
    ntk = batch_ntk(6400)
    ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
    for ni in (6, 8, 10):
        for tol in (1e-6, 1e-5, 1e-4):
            report(f"plain {ni}/{tol:g}", ntk, params(ninit=ni, tol=tol, desc=1, tol_skip=-1e-10), ref)
    for pw, fac in ((1.0, 200.0), (1.0, 2000.0), (0.5, 200.0), (1.5, 2000.0), (1.5, 20000.0)):
        for ni in (6, 8):
            for tol in (1e-8, 1e-9):
                report(f"est pow {pw} fac {fac:g} {ni}/{tol:g}", ntk, params(ninit=ni, tol=tol, desc=1, tol_skip=-1e-10, est=1, est_pow=pw, est_fac=fac), ref)


@experiment("qp3", "the candidates of qp1/qp2 on random multi-scale mixtures")
def qp3():
    cands = [("now 10/1e-7", dict(ninit=10, tol=1e-7)), ("plain 8/1e-6", dict(ninit=8, tol=1e-6)), ("plain 10/1e-6", dict(ninit=10, tol=1e-6)),
             ("est1.0 8/1e-8", dict(ninit=8, tol=1e-8, est=1, est_pow=1.0)), ("est1.5 8/1e-9", dict(ninit=8, tol=1e-9, est=1)),
             ("est1.5 10/1e-9", dict(ninit=10, tol=1e-9, est=1)), ("est1.0 10/1e-8", dict(ninit=10, tol=1e-8, est=1, est_pow=1.0))]
    for N in (2, 3):
        for ln in (False, True):
            for gam in (0.0, 4/3):
                ntk, types = wild_ntk(3000, N, seed=5 + N, lognormal_others=ln)
                ref, _, _ = run(ntk, params(ninit=64, tol=1e-14), N, gam=gam, types=types)
                tag = f"N{N} ln{int(ln)} gam{gam:.1f} "
                for name, kw in cands:
                    report(tag + name, ntk, params(desc=1, tol_skip=-1e-10, **kw), ref, N, types, gam)


@experiment("qp4", "7-10 pieces at 1e-7 ... 1e-6 on the batch and on random mixtures")
def qp4():
    ntk = batch_ntk(6400)
    ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
    for ni in (7, 8, 9, 10):
        for tol in (1e-7, 3e-7, 1e-6):
            report(f"plain {ni}/{tol:g}", ntk, params(ninit=ni, tol=tol, desc=1, tol_skip=-1e-10), ref)
    for N in (2, 3):
        for ln in (False, True):
            gam = 4/3
            w, types = wild_ntk(3000, N, seed=5 + N, lognormal_others=ln)
            r, _, _ = run(w, params(ninit=64, tol=1e-14), N, gam=gam, types=types)
            for ni, tol in ((10, 1e-7), (9, 1e-7), (8, 1e-7), (8, 3e-7)):
                report(f"wild N{N} ln{int(ln)} {ni}/{tol:g}", w, params(ninit=ni, tol=tol, desc=1, tol_skip=-1e-10), r, N, types, gam)


@experiment("probe1", "fraction of a rule's range that carries the integral")
def probe1():
    ntk = batch_ntk(2000)
    def rule_profile(p, j, N=3, gam=4/3):
        th = [ntk[3*m+1, p] for m in range(N)]; k = [ntk[3*m+2, p] for m in range(N)]
        A = 2*k[j]+gam; lgA = math.lgamma(A)
        tlo = max(-690.0, min(-1.0, (math.log(1e-13)+math.lgamma(A+1))/A)); thi = math.log(A+2+math.sqrt(60*(A+2))+30)
        t = np.linspace(tlo, thi, 1601); u = np.exp(t)
        W = np.exp(A*t-u-lgA)
        lt = math.log(th[j]); cj = math.lgamma(k[j])+k[j]*lt
        up = 0; den = 1
        for m in range(N):
            if m == j: continue
            lr = (k[m]-k[j])*(t+lt) + (1-th[j]/th[m])*u - (math.lgamma(k[m])+k[m]*math.log(th[m])-cj)
            rho = np.exp(np.minimum(lr, 700)); den = den+rho
            if m > j: up = up+rho
        g = W*up/den
        return t, g, W, A
    fr = []
    for p in range(0, 2000, 7):
        for j in range(2):
            if not ntk[3*j, p] > 0: continue
            t, g, W, A = rule_profile(p, j)
            g2 = g*np.exp(2*t)
            I0 = np.trapz(g, t); I2 = np.trapz(g2, t)
            need = (g > 1e-11*I0/ (t[-1]-t[0])*1.0) | (g2 > 1e-11*I2/(t[-1]-t[0]))
            fr.append((j, need.mean(), I0, (t[-1]-t[0])))
    fr = np.array(fr)
    for j in range(2):
        s = fr[fr[:,0]==j]
        print(j, "needed fraction of range: mean %.2f median %.2f; I0 median %.2e min %.2e; range %.2f" % (s[:,1].mean(), np.median(s[:,1]), np.median(s[:,2]), s[:,2].min(), s[:,3].mean()))


@experiment("sim_sort", "cost predictors: ranking the parcels of a workgroup by predicted / actual evaluations")
def sim_sort():
    ntk = batch_ntk(25600)
    T, cost, st = run(ntk, params(ninit=12, tol=1e-8))
    T, pred, st = run(ntk, params(ninit=12, tol=1e-8), edge_cost=-1.0)
    ev = cost / 15.0
    tot = ev.sum(0); ptot = pred.sum(0)
    print("corr(pred, actual) total per parcel:", np.corrcoef(tot, ptot)[0, 1], " mean evals", tot.mean(), "mean init", ptot.mean())
    def merged_eff(tot, order_key, wg):
        n = tot.size; busy = 0
        for w0 in range(0, n - wg + 1, wg):
            c = tot[w0:w0 + wg]
            if order_key is not None:
                c = c[np.argsort(order_key[w0:w0 + wg], kind="stable")]
            busy += c.reshape(-1, 64).max(1).sum() * 64
        return tot[: (n // wg) * wg].sum() / busy
    for wg in (256, 512):
        print(f"WG {wg}: unsorted {merged_eff(tot, None, wg):.3f}  sorted by actual {merged_eff(tot, tot, wg):.3f}  sorted by predicted {merged_eff(tot, ptot, wg):.3f}")


@experiment("sim_pool", "workgroup-level pool of rule records vs the static merge")
def sim_pool():
    """lane utilisation of a workgroup-level pool of rule records (dynamic grabbing) vs static assignment"""
    ntk = batch_ntk(25600)
    def sim(cost, wg_lanes=256, order="natural"):
        # cost: (2, n) panel evaluations per rule; records of a WG: its wg_lanes parcels x 2 rules
        n = cost.shape[1]; tot_busy = 0; tot_work = 0
        for w0 in range(0, n - wg_lanes + 1, wg_lanes):
            c = cost[:, w0:w0 + wg_lanes]
            if order == "natural":   # rule 0 of all parcels first, then rule 1
                pool = list(c[0]) + list(c[1])
            elif order == "sorted":  # longest first (needs a sort: not free)
                pool = sorted(list(c[0]) + list(c[1]), reverse=True)
            pool = [int(round(x)) for x in pool]
            tot_work += sum(pool)
            # event simulation: lanes grab in order of becoming free; ties by lane id
            h = [(0, l) for l in range(wg_lanes)]
            heapq.heapify(h)
            fin = np.zeros(wg_lanes)
            for p in pool:
                t, l = heapq.heappop(h)
                heapq.heappush(h, (t + p, l))
                fin[l] = t + p
            # a wave is busy until its last lane finishes
            tot_busy += sum(fin[w:w + 64].max() * 64 for w in range(0, wg_lanes, 64))
        return tot_work / tot_busy
    T, cost, st = run(ntk, params())
    ev = cost / 15.0
    print("current rule: static per-rule lanes", (ev.sum() / sum(ev[j].reshape(-1, 64).max(1).sum() * 64 for j in range(2))))
    print("  merged static", ev.sum() / (ev.sum(0).reshape(-1, 64).max(1).sum() * 64))
    for wl in (256, 512):
        print(f"  pool WG {wl} natural", sim(ev, wl), " sorted", sim(ev, wl, "sorted"))
    T, cost, st = run(ntk, gparams(topdown=1, skip_tol=1e-11))
    ev = cost / 15.0
    print("graded td skip: static per-rule", (ev.sum() / sum(ev[j].reshape(-1, 64).max(1).sum() * 64 for j in range(2))))
    for wl in (256, 512):
        print(f"  pool WG {wl} natural", sim(ev, wl), " sorted", sim(ev, wl, "sorted"))


@experiment("trap1", "trapezoidal rule in t with interval doubling")
def trap1():
    n = 3000
    ntk = batch_ntk(n)
    ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
    sc = scales(ntk)
    N = 3; gam = 4/3
    def integrand_t(p, j, t):
        th = [ntk[3*m+1, p] for m in range(N)]; k = [ntk[3*m+2, p] for m in range(N)]
        A = 2*k[j]+gam; lgA = math.lgamma(A); u = np.exp(t)
        W = np.exp(A*t-u-lgA)
        lt = math.log(th[j]); cj = math.lgamma(k[j])+k[j]*lt
        up = 0; den = 1
        for m in range(N):
            if m == j: continue
            lr = (k[m]-k[j])*(t+lt) + (1-th[j]/th[m])*u - (math.lgamma(k[m])+k[m]*math.log(th[m])-cj)
            rho = np.exp(np.minimum(lr, 700)); den = den+rho
            if m > j: up = up+rho
        g = W*up/den
        s = u*th[j]
        return np.array([g, g*s, g*s*s])
    def rng_t(A):
        tlo = max(-690.0, min(-1.0, (math.log(1e-13)+math.lgamma(A+1))/A)); thi = math.log(A+2+math.sqrt(60*(A+2))+30)
        return tlo, thi
    res = {}
    for var in ("t", "sp"):
        need = []
        for p in range(0, n, 5):
            for j in range(2):
                if not ntk[3*j, p] > 0: continue
                k = ntk[3*j+2, p]; A = 2*k+gam
                tlo, thi = rng_t(A)
                r = ref[3*j:3*j+3, p]; s3 = sc[3*j:3*j+3, p]
                got = None
                for nn in (16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024):
                    if var == "t":
                        t = np.linspace(tlo, thi, nn+1); h = (thi-tlo)/nn
                        v = integrand_t(p, j, t)*h
                    else:
                        # u = softplus(v)/c, c = 1/sqrt(A)?  choose crossover at u ~ A/2
                        c = 2.0/A*math.log(2)
                        ulo, uhi = math.exp(tlo), math.exp(thi)
                        inv = lambda u: np.log(np.expm1(c*u)) if c*u < 30 else c*u
                        vlo, vhi = inv(ulo), inv(uhi)
                        vv = np.linspace(vlo, vhi, nn+1); h = (vhi-vlo)/nn
                        spv = np.logaddexp(0, vv); u = spv/c; t = np.log(u)
                        dudv = 1/(1+np.exp(-vv))/c
                        v = integrand_t(p, j, t)*(dudv/u)*h
                    I = v.sum(1)
                    err = np.max(np.abs(I-r)/np.maximum(np.abs(r), 1e-10*s3))
                    if err < 1e-9:
                        got = nn; break
                need.append(got if got else 2048)
        need = np.array(need)
        print(var, "nodes needed for 1e-9 rel: mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f" % (need.mean(), np.median(need), np.percentile(need,90), np.percentile(need,99), need.max()))
        print("   hist", {v: int((need==v).sum()) for v in np.unique(need)})


@experiment("trap2", "trapezoid: square-law extrapolation")
def trap2():
    n = 3000
    ntk = batch_ntk(n)
    ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
    sc = scales(ntk)
    N = 3; gam = 4/3
    def mk_integrand(th, k, j, gam):
        N = len(th)
        A = 2*k[j]+gam; lgA = math.lgamma(A)
        lt = math.log(th[j]); cj = math.lgamma(k[j])+k[j]*lt
        da = [k[m]-k[j] for m in range(N)]; cb = [1-th[j]/th[m] for m in range(N)]
        dc = [math.lgamma(k[m])+k[m]*math.log(th[m])-cj for m in range(N)]
        def f(t):
            u = np.exp(t); W = np.exp(A*t-u-lgA)
            up = 0; den = 1
            for m in range(N):
                if m == j: continue
                rho = np.exp(np.minimum(da[m]*(t+lt)+cb[m]*u-dc[m], 700)); den = den+rho
                if m > j: up = up+rho
            g = W*up/den; s = u*th[j]
            return np.array([g, g*s, g*s*s])
        return f, A
    def rng_t(A, eps=1e-13):
        tlo = max(-690.0, min(-1.0, (math.log(eps)+math.lgamma(A+1))/A)); thi = math.log(A+2+math.sqrt(60*(A+2))+30)
        return tlo, thi
    def trap_adapt(f, tlo, thi, n0, tolD, s3, floor=1e-10, nmax=16384):
        h = (thi-tlo)/n0
        t = tlo + h*np.arange(n0+1)
        v = f(t)
        Tn = v.sum(1)*h
        Th = v[:, ::2].sum(1)*(2*h)   # subset n0/2
        nodes = n0+1
        nn = n0
        while True:
            D = np.abs(Tn-Th)
            if np.all(D <= tolD*np.maximum(np.abs(Tn), floor*s3)) or nn >= nmax:
                return Tn, nodes
            # refine: midpoints
            tm = tlo + h*(np.arange(nn)+0.5)
            vm = f(tm)
            Th = Tn
            Tn = 0.5*(Tn + vm.sum(1)*h)
            h *= 0.5; nodes += nn; nn *= 2
    for n0 in (32, 48, 64):
      for tolD in (1e-4, 1e-5, 1e-6, 1e-9):
        errs = []; nds = []
        for p in range(0, n, 3):
            th = [ntk[3*m+1, p] for m in range(N)]; k = [ntk[3*m+2, p] for m in range(N)]
            for j in range(2):
                if not ntk[3*j, p] > 0: continue
                f, A = mk_integrand(th, k, j, gam)
                tlo, thi = rng_t(A)
                r = ref[3*j:3*j+3, p]; s3 = sc[3*j:3*j+3, p]
                T, nodes = trap_adapt(f, tlo, thi, n0, tolD, s3)
                # below-tlo mass ignored in both? ref includes it; add same
                u_lo = math.exp(tlo)
                # compute sigma at tlo
                g0 = f(np.array([tlo]))[0,0]
                W0 = math.exp(A*tlo-u_lo-math.lgamma(A))
                sig = g0/W0 if W0 > 0 else 0
                T = T.copy(); T[0] += sig*math.exp(A*tlo-math.lgamma(A+1))
                errs.append(np.max(np.abs(T-r)/np.maximum(np.abs(r), 1e-10*s3))); nds.append(nodes)
        errs = np.array(errs); nds = np.array(nds)
        print(f"n0 {n0} tolD {tolD:g}: nodes mean {nds.mean():.0f} p50 {np.median(nds):.0f} p90 {np.percentile(nds,90):.0f} max {nds.max()}  err max {errs.max():.1e} p99.9 {np.percentile(errs,99.9):.1e} p99 {np.percentile(errs,99):.1e}")


@experiment("trap3", "trapezoid in a softplus variable")
def trap3():
    gam = 4/3
    def setup(th, k, j, gam):
        N = len(th)
        A = 2*k[j]+gam; lgA = math.lgamma(A)
        lt = math.log(th[j]); cj = math.lgamma(k[j])+k[j]*lt
        da = np.array([k[m]-k[j] for m in range(N)]); cb = np.array([1-th[j]/th[m] for m in range(N)])
        dc = np.array([math.lgamma(k[m])+k[m]*math.log(th[m])-cj for m in range(N)])
        return A, lgA, lt, da, cb, dc
    def integrand(t, j, th, A, lgA, lt, da, cb, dc):
        N = len(da)
        u = np.exp(t); W = np.exp(A*t-u-lgA)
        up = 0; den = 1
        for m in range(N):
            if m == j: continue
            rho = np.exp(np.minimum(da[m]*(t+lt)+cb[m]*u-dc[m], 700)); den = den+rho
            if m > j: up = up+rho
        g = W*up/den; s = u*th[j]
        return np.array([g, g*s, g*s*s])
    def dmodel(t, j, A, lt, da, cb, dc, win=30.0):
        """pole-distance model on a grid t"""
        N = len(da); u = np.exp(t)
        L = [da[m]*(t+lt)+cb[m]*u-dc[m] if m != j else np.zeros_like(t) for m in range(N)]
        lam = [np.abs(da[m])+1.72*np.abs(cb[m])*u if m != j else np.zeros_like(t) for m in range(N)]
        Lmax = np.max(L, axis=0)
        d = np.full_like(t, 1e9)
        for a in range(N):
            for b in range(a+1, N):
                ok = (L[a] >= Lmax-win) & (L[b] >= Lmax-win)
                dd = np.sqrt((L[a]-L[b])**2+math.pi**2)/(lam[a]+lam[b]+1e-300)
                d = np.where(ok, np.minimum(d, dd), d)
        return d
    def rng_t(A, eps=1e-13):
        tlo = max(-690.0, min(-1.0, (math.log(eps)+math.lgamma(A+1))/A)); thi = math.log(A+2+math.sqrt(60*(A+2))+30)
        return tlo, thi
    def sig_range(A, tlo, thi, eps=1e-12):
        """where W_0 or W_2 >= eps of its max"""
        t = np.linspace(tlo, thi, 400)
        ok = np.zeros_like(t, bool)
        for m in (0, 2):
            lw = (A+m)*t-np.exp(t); ok |= lw >= lw.max()+math.log(eps)
        return t[ok][0], t[ok][-1]
    def run_case(th, k, j, ref3, s3, cfac, gam=4/3):
        A, lgA, lt, da, cb, dc = setup(th, k, j, gam)
        tlo, thi = rng_t(A)
        ta, tb = sig_range(A, tlo, thi)
        tg = np.linspace(ta, tb, 200)
        dmin = dmodel(tg, j, A, lt, da, cb, dc).min()
        h = min(cfac*dmin, 0.25)
        n = int(math.ceil((thi-tlo)/h)); h = (thi-tlo)/n
        t = tlo+h*np.arange(n+1)
        T = integrand(t, j, th, A, lgA, lt, da, cb, dc).sum(1)*h
        u_lo = math.exp(tlo); W0 = math.exp(A*tlo-u_lo-lgA)
        g0 = integrand(np.array([tlo]), j, th, A, lgA, lt, da, cb, dc)[0, 0]
        T[0] += (g0/W0 if W0 > 0 else 0)*math.exp(A*tlo-math.lgamma(A+1))
        err = np.max(np.abs(T-ref3)/np.maximum(np.abs(ref3), 1e-10*s3))
        errs = np.max(np.abs(T-ref3)/s3)
        return n, err, errs, dmin
    if __name__ == "__main__":
        n = 3000
        ntk = batch_ntk(n)
        ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
        sc = scales(ntk)
        N = 3
        for cfac in (0.27, 0.35, 0.45):
            nds, errs, errss, dm = [], [], [], []
            for p in range(0, n, 3):
                th = [ntk[3*m+1, p] for m in range(N)]; k = [ntk[3*m+2, p] for m in range(N)]
                for j in range(2):
                    if not ntk[3*j, p] > 0: continue
                    nn, e, es, d = run_case(th, k, j, ref[3*j:3*j+3, p], sc[3*j:3*j+3, p], cfac)
                    nds.append(nn); errs.append(e); errss.append(es); dm.append(d)
            nds = np.array(nds); errs = np.array(errs); errss = np.array(errss); dm = np.array(dm)
            print(f"cfac {cfac}: nodes mean {nds.mean():.0f} p50 {np.median(nds):.0f} p90 {np.percentile(nds,90):.0f} p99 {np.percentile(nds,99):.0f} max {nds.max()};"
                  f" rel err max {errs.max():.1e} p99.9 {np.percentile(errs,99.9):.1e}; of-scale max {errss.max():.1e}; dmin p1 {np.percentile(dm,1):.3f} p50 {np.median(dm):.3f}")


@experiment("xmarks", "round 5: marks graded around the crossings rho_m = 1 (the transitions of weighting_fn) next to / instead of the cores")
def xmarks():
    ntk = batch_ntk(6400)
    ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
    base = dict(tol=1e-7, desc=1, tol_skip=-1e-10)
    for ni in (10, 8, 6):
        for xm in (0, 1, 2):
            report(f"desc {ni}/1e-7 xmarks {xm}", ntk, params(ninit=ni, xmarks=xm, **base), ref)


@experiment("graded5", "round 5: the a-priori graded layout against the current rule (with cost hints the lane balance no longer decides)")
def graded5():
    ntk = batch_ntk(6400)
    ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
    report("current: desc 10/1e-7 term 1e-10", ntk, params(ninit=10, tol=1e-7, desc=1, tol_skip=-1e-10), ref)
    for cs, ce in ((3.5, 12.0), (2.5, 8.0), (5.0, 12.0)):
        for tn in (1e-5, 1e-6, 1e-7):
            for td, sk in ((1, 1e-10), (0, 0.0)):
                report(f"graded K15 {cs}/{ce} net {tn:g} td{td} skip {sk:g}", ntk, gparams(c_step=cs, c_exp=ce, tol_net=tn, topdown=td, skip_tol=sk), ref)


if __name__ == "__main__":
    args = sys.argv[1:]
    if not args or args[0] in ("-l", "--list"):
        for k, (_, what) in EXPERIMENTS.items():
            print(f"{k:10s} {what}")
        raise SystemExit(0)
    for a in args:
        print(f"==== {a}: {EXPERIMENTS[a][1]}", flush=True)
        EXPERIMENTS[a][0]()
