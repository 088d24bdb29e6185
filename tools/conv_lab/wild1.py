from lab import *
for N in (2, 3):
    for ln in (False, True):
        for gam in (0.0, 4/3):
            ntk, types = wild_ntk(4000, N, seed=5 + N, lognormal_others=ln)
            ref, _, _ = run(ntk, params(ninit=64, tol=1e-14), N, gam=gam, types=types)
            tag = f"N{N} ln{int(ln)} gam{gam:.1f}"
            report(tag + " current", ntk, params(), ref, N, types, gam)
            report(tag + " K15 3.5/12", ntk, gparams(), ref, N, types, gam)
            report(tag + " K15 3.5/12 td skip1e-11", ntk, gparams(topdown=1, skip_tol=1e-11), ref, N, types, gam)
            report(tag + " K15 2.5/8 td skip1e-11", ntk, gparams(c_step=2.5, c_exp=8, topdown=1, skip_tol=1e-11), ref, N, types, gam)
