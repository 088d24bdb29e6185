from lab import *
ntk = batch_ntk(6400)
ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
for ni in (6, 8, 10, 12, 16):
    for tol in (1e-8, 1e-7, 1e-6):
        report(f"ninit {ni} tol {tol:g}", ntk, params(ninit=ni, tol=tol), ref)
for ni in (8, 12):
    for tol in (1e-9, 1e-10):
      for ep in (1.5, 2.0):
        report(f"ninit {ni} tol {tol:g} est_pow {ep}", ntk, params(ninit=ni, tol=tol, est_pow=ep), ref)
