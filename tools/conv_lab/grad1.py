from lab import *
ntk = batch_ntk(6400)
ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
for cs in (1.2, 1.7, 2.4):
    for ce in (4.0, 6.0, 8.0):
        report_graded(f"K15 c_step {cs} c_exp {ce}", ntk, gparams(c_step=cs, c_exp=ce), ref)
for rule, cs in ((10, 1.0), (10, 1.4), (12, 1.4), (12, 1.8), (16, 2.0), (16, 2.5)):
    for ce in (4.0, 6.0):
        report_graded(f"GL{rule} c_step {cs} c_exp {ce}", ntk, gparams(rule=rule, c_step=cs, c_exp=ce), ref)
