from lab import *
ntk = batch_ntk(6400)
ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
report("now: desc 10/1e-7 term 1e-10", ntk, params(ninit=10, tol=1e-7, desc=1, tol_skip=-1e-10), ref)
for ni in (4, 5, 6, 8, 10):
    for tol in (1e-8, 1e-9, 1e-10):
        report(f"qk15 estimate {ni}/{tol:g}", ntk, params(ninit=ni, tol=tol, desc=1, tol_skip=-1e-10, est=1), ref)
