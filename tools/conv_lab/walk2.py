from lab import *
ntk = batch_ntk(6400)
ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
base = dict(ninit=12, tol=1e-8)
report("now (12, 1e-8) ascending", ntk, params(**base), ref)
report("descending, no skip", ntk, params(desc=1, **base), ref)
for ts in (1e-12, 1e-11, 1e-10):
    report(f"ascending skip {ts:g}", ntk, params(tol_skip=ts, **base), ref)
    report(f"descending skip {ts:g}", ntk, params(desc=1, tol_skip=ts, **base), ref)
