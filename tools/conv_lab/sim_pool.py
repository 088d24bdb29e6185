"""lane utilisation of a workgroup-level pool of rule records (dynamic grabbing) vs static assignment"""
import numpy as np, heapq
from lab import *
ntk = batch_ntk(25600)
def sim(cost, wg_lanes=256, order="natural"):
    # cost: (2, n) panel evaluations per rule; records of a WG: its wg_lanes parcels x 2 rules
    n = cost.shape[1]; tot_busy = 0; tot_work = 0
    for w0 in range(0, n - wg_lanes + 1, wg_lanes):
        c = cost[:, w0:w0 + wg_lanes]
        if order == "natural":   # rule 0 of all parcels first, then rule 1
            pool = list(c[0]) + list(c[1])
        elif order == "sorted":  # longest first (needs a sort: not free)
            pool = sorted(list(c[0]) + list(c[1]), reverse=True)
        pool = [int(round(x)) for x in pool]
        tot_work += sum(pool)
        # event simulation: lanes grab in order of becoming free; ties by lane id
        h = [(0, l) for l in range(wg_lanes)]
        heapq.heapify(h)
        fin = np.zeros(wg_lanes)
        for p in pool:
            t, l = heapq.heappop(h)
            heapq.heappush(h, (t + p, l))
            fin[l] = t + p
        # a wave is busy until its last lane finishes
        tot_busy += sum(fin[w:w + 64].max() * 64 for w in range(0, wg_lanes, 64))
    return tot_work / tot_busy
T, cost, st = run(ntk, params())
ev = cost / 15.0
print("current rule: static per-rule lanes", (ev.sum() / sum(ev[j].reshape(-1, 64).max(1).sum() * 64 for j in range(2))))
print("  merged static", ev.sum() / (ev.sum(0).reshape(-1, 64).max(1).sum() * 64))
for wl in (256, 512):
    print(f"  pool WG {wl} natural", sim(ev, wl), " sorted", sim(ev, wl, "sorted"))
T, cost, st = run(ntk, gparams(topdown=1, skip_tol=1e-11))
ev = cost / 15.0
print("graded td skip: static per-rule", (ev.sum() / sum(ev[j].reshape(-1, 64).max(1).sum() * 64 for j in range(2))))
for wl in (256, 512):
    print(f"  pool WG {wl} natural", sim(ev, wl), " sorted", sim(ev, wl, "sorted"))
