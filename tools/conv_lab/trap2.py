import numpy as np, math, sys
from lab import *
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
