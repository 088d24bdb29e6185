import numpy as np, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import __graft_entry__ as g
pkg = g.load_package()
from oracle import cloudy_oracle as O
import bench
from test_gpu_parity import mixed_moments
NORMS = bench.NORMS
np.set_printoptions(precision=3, linewidth=220)
dist_types = [1, 1]
for kind, kf, okf0 in (("linear", pkg.LinearKernelFunction(5.0), O.kernel_func(O.KF_LINEAR, 5.0)),
                       ("hydro", pkg.HydrodynamicKernelFunction(1e2 * np.pi), O.kernel_func(O.KF_HYDRODYNAMIC, 1e2 * np.pi))):
    q = 8
    okf = O.get_normalized_kernel_func(okf0, NORMS)
    kfn = pkg.get_normalized_kernel_func(kf, NORMS)
    op = O.make_params(list(dist_types), np.zeros((1, 1)), (float("inf"),) * 2, norms=NORMS)
    mom = mixed_moments(dist_types, 1500, seed=300 + 14 + q)
    want, sc = O.rhs_coal_numerical_converged_batch(op, okf, q, mom, with_scale=True)
    m = pkg.DeviceArray.from_numpy(mom)
    outs = {}
    for name, spec in (("jit", 1), ("aot", -1)):
        plan = pkg.NumericalPlan(dist_types, kfn, NORMS, q, quad_mode=1, specialize=spec)
        dm = pkg.DeviceArray.zeros(*mom.shape)
        pkg._lib.check(pkg.lib().cloudy_coal_rhs(plan.handle, 1500, 1500, m.ptr, dm.ptr, None))
        outs[name] = dm.to_numpy()
        err = np.abs(outs[name] - want) / np.maximum(sc, 1e-300)
        print(kind, name, "max err", np.nanmax(err), "n>1e-12:", (np.nanmax(err, axis=0) > 1e-12).sum())
    # closure parameters of both paths
    cd = pkg.CoalescenceData(pkg.CoalescenceTensor(np.array([[0.0]])), (3, 3), (float("inf"),) * 2, NORMS)
    P = pkg.DeviceArray.zeros(6, 1500)
    pkg._lib.check(pkg.lib().cloudy_update_dist_from_moments(cd.plan(dist_types).handle, 1500, 1500, m.ptr, P.ptr, None))
    prm_o = O.update_dist_batch(op, mom)
    print("params bit-equal:", np.array_equal(P.to_numpy(), prm_o))
    i = 254
    print("parcel 254 jit-want", (outs["jit"][:, i] - want[:, i]) / sc[:, i], "aot-want", (outs["aot"][:, i] - want[:, i]) / sc[:, i])
