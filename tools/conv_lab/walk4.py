from lab import *
ntk = batch_ntk(6400)
ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
base = dict(ninit=12, tol=1e-8, desc=1)
report("desc no skip", ntk, params(**base), ref)
report("desc per-panel skip 1e-10", ntk, params(tol_skip=1e-10, **base), ref)
for ts in (1e-11, 1e-10, 1e-9):
    report(f"desc terminate-only {ts:g}", ntk, params(tol_skip=-ts, **base), ref)
