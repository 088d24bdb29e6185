from lab import *
ntk = batch_ntk(6400)
ref, _, _ = run(ntk, params(ninit=64, tol=1e-14))
report("current", ntk, params(), ref)
report("K15 3.5/12", ntk, gparams(), ref)
for rule in (10, 12):
    for cs, ce in ((2.5, 8.0), (3.0, 10.0), (3.5, 12.0), (5.0, 12.0)):
        for tn in (1e-4, 1e-5):
            report(f"GL{rule} {cs}/{ce} net {tn:g}", ntk, gparams(rule=rule, c_step=cs, c_exp=ce, tol_net=tn), ref)
