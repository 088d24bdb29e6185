from lab import *
ntk = batch_ntk(6400)
ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
for ni in (8, 10, 12):
    for tol in (1e-8, 1e-7, 1e-6):
        report(f"desc ninit {ni} tol {tol:g} skip 1e-10", ntk, params(ninit=ni, tol=tol, desc=1, tol_skip=1e-10), ref)
