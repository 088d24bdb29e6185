from lab import *
ntk = batch_ntk(6400)
ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
for cs in (2.4, 3.5, 5.0, 8.0):
    for ce in (8.0, 12.0, 16.0, 24.0):
        report_graded(f"K15 c_step {cs} c_exp {ce} wmax 3", ntk, gparams(c_step=cs, c_exp=ce, wmax=3.0), ref)
