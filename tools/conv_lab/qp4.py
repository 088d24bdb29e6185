from lab import *
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
