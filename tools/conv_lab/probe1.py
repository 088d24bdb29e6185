import numpy as np, math
from lab import *
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
