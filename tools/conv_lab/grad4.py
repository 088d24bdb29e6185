from lab import *
ntk = batch_ntk(6400)
ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
for td in (0, 1):
    for sk in (0.0, 1e-12, 1e-11, 1e-10):
        report(f"K15 3.5/12 topdown {td} skip {sk:g}", ntk, gparams(topdown=td, skip_tol=sk), ref)
for eps in (1e-13, 1e-11):
    report(f"K15 3.5/12 range_eps {eps:g}", ntk, gparams(range_eps=float(np.log(eps))), ref)
