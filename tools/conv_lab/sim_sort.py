import numpy as np
from lab import *
ntk = batch_ntk(25600)
T, cost, st = run(ntk, params(ninit=12, tol=1e-8))
T, pred, st = run(ntk, params(ninit=12, tol=1e-8), edge_cost=-1.0)
ev = cost / 15.0
tot = ev.sum(0); ptot = pred.sum(0)
print("corr(pred, actual) total per parcel:", np.corrcoef(tot, ptot)[0, 1], " mean evals", tot.mean(), "mean init", ptot.mean())
def merged_eff(tot, order_key, wg):
    n = tot.size; busy = 0
    for w0 in range(0, n - wg + 1, wg):
        c = tot[w0:w0 + wg]
        if order_key is not None:
            c = c[np.argsort(order_key[w0:w0 + wg], kind="stable")]
        busy += c.reshape(-1, 64).max(1).sum() * 64
    return tot[: (n // wg) * wg].sum() / busy
for wg in (256, 512):
    print(f"WG {wg}: unsorted {merged_eff(tot, None, wg):.3f}  sorted by actual {merged_eff(tot, tot, wg):.3f}  sorted by predicted {merged_eff(tot, ptot, wg):.3f}")
