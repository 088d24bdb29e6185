import numpy as np, math
from lab import *
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
