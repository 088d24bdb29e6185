from lab import *
import sys
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
