from lab import *
import sys
for N in (2, 3):
    for ln in (False, True):
        for gam in (0.0, 4/3):
            ntk, types = wild_ntk(3000, N, seed=5 + N, lognormal_others=ln)
            ref, _, _ = run(ntk, params(ninit=64, tol=1e-14), N, gam=gam, types=types)
            tag = f"N{N} ln{int(ln)} gam{gam:.1f}"
            report(tag + " asc 12/1e-8 (now)", ntk, params(ninit=12, tol=1e-8), ref, N, types, gam)
            for ni, tol, ts in ((12, 1e-8, 1e-10), (10, 1e-7, 1e-10), (8, 1e-7, 1e-10), (8, 1e-6, 1e-10), (10, 1e-7, 1e-11)):
                report(tag + f" desc {ni}/{tol:g} skip {ts:g}", ntk, params(ninit=ni, tol=tol, desc=1, tol_skip=ts), ref, N, types, gam)
