from lab import *
ntk = batch_ntk(6400)
ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
for ni in (6, 8, 10):
    for tol in (1e-6, 1e-5, 1e-4):
        report(f"plain {ni}/{tol:g}", ntk, params(ninit=ni, tol=tol, desc=1, tol_skip=-1e-10), ref)
for pw, fac in ((1.0, 200.0), (1.0, 2000.0), (0.5, 200.0), (1.5, 2000.0), (1.5, 20000.0)):
    for ni in (6, 8):
        for tol in (1e-8, 1e-9):
            report(f"est pow {pw} fac {fac:g} {ni}/{tol:g}", ntk, params(ninit=ni, tol=tol, desc=1, tol_skip=-1e-10, est=1, est_pow=pw, est_fac=fac), ref)
