import numpy as np, math, sys
from lab import *
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
