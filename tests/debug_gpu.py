"""ad-hoc GPU diagnostics (not a test)"""
import sys, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge, bench
from oracle import cloudy_oracle as O
pkg = ge.load_package()
np.set_printoptions(precision=17, linewidth=200)
for name, n in (("cfg2", 200000), ("cfg3a", 200000), ("cfg3b", 20000)):
    wl = bench.make_workload(name, n, seed=99)
    m = pkg.DeviceArray.from_numpy(wl["mom"]); dm = pkg.DeviceArray.zeros(*wl["mom"].shape)
    pkg.make_box_model_rhs(pkg.AnalyticalCoalStyle())(dm, m, wl["par"], 0.0)
    d = dm.to_numpy()
    want, scale = O.rhs_coal_batch(bench.oracle_params(name), wl["mom"], with_scale=True)
    plan = wl["coal_data"].plan(wl["dist_types"])
    prm = pkg.update_dist_from_moments(plan, m).to_numpy()
    oprm = O.update_dist_batch(bench.oracle_params(name), wl["mom"])
    print(name, "params equal:", np.array_equal(prm, oprm), "n diff", (prm != oprm).sum())
    with np.errstate(all="ignore"):
        rel = np.abs(d - want) / scale
    bad = np.argwhere(~(rel <= 1e-10) & ~(np.isnan(d) & np.isnan(want)))
    print(name, "bad entries", len(bad), "nan gpu", np.isnan(d).sum(), "nan oracle", np.isnan(want).sum(), "max rel", np.nanmax(rel))
    for q, i in bad[:6]:
        print("  parcel", i, "plane", q, "gpu", d[q, i], "oracle", want[q, i], "scale", scale[q, i])
        print("    mom", wl["mom"][:, i]); print("    gpu params", prm[:, i]); print("    ora params", oprm[:, i])
