"""Driver of the conv_lab sandbox: node counts / errors of candidate T_m rules on the cfg4q batch and on wild mixtures (CPU)."""
import ctypes as C
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import cloudy_oracle as O  # noqa: E402


class Params(C.Structure):
    _fields_ = [("ninit", C.c_int), ("tol", C.c_double), ("marks", C.c_int), ("lmax", C.c_int), ("floor_", C.c_double),
                ("range_eps", C.c_double), ("imax", C.c_int), ("desc", C.c_int), ("tol_skip", C.c_double), ("est", C.c_int),
                ("est_pow", C.c_double), ("est_fac", C.c_double), ("xmarks", C.c_int)]


class GradParams(C.Structure):
    _fields_ = [("c_step", C.c_double), ("c_exp", C.c_double), ("wmax", C.c_double), ("win", C.c_double), ("rule", C.c_int),
                ("tol_net", C.c_double), ("floor_", C.c_double), ("range_eps", C.c_double), ("skip_tol", C.c_double),
                ("lmax", C.c_int), ("topdown", C.c_int)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_long) for n in ("nodes", "edges", "evals", "rejects", "skipped", "init_panels")]


def params(**kw):
    d = dict(ninit=16, tol=1e-9, marks=1, lmax=12, floor_=1e-10, range_eps=float(np.log(1e-13)), imax=12, desc=0, tol_skip=0.0, est=0, est_pow=1.5, est_fac=200.0, xmarks=0)
    d.update(kw)
    return Params(**d)


def gparams(**kw):
    d = dict(c_step=3.5, c_exp=12.0, wmax=3.0, win=30.0, rule=15, tol_net=1e-5, floor_=1e-10,
             range_eps=float(np.log(1e-13)), skip_tol=0.0, lmax=12, topdown=0)
    d.update(kw)
    return GradParams(**d)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if os.system(f"gcc -O2 -shared -fPIC -o {HERE}/liblab.so {HERE}/lab.c -lm"):
            raise RuntimeError("lab.c does not compile")
        _lib = C.CDLL(os.path.join(HERE, "liblab.so"))
    return _lib


def batch_ntk(n, N=3, seed=bench.SEED):
    mom = bench.synth_moments(N, n, seed)
    p = O.make_params([O.GAMMA] * N, np.zeros((1, 1)), (np.inf,) * N, norms=bench.NORMS)
    return np.ascontiguousarray(O.update_dist_batch(p, mom))


def wild_ntk(n, N, seed=11, lognormal_others=True):
    """(n, theta, k) planes like the random multi-scale mixtures of tests/test_numerical_oracle.py; types per mode are
    fixed for the batch (Gamma-family rule modes; the LAST mode may be Lognormal)."""
    rng = np.random.default_rng(seed)
    ntk = np.zeros((3 * N, n))
    types = [0] * N
    if lognormal_others:
        types[-1] = 1
    for m in range(N):
        ntk[3 * m] = 10 ** rng.uniform(-1, 2, n)
        if types[m] == 1:
            ntk[3 * m + 1] = rng.uniform(-3, 2, n)
            ntk[3 * m + 2] = np.where(rng.random(n) < 0.5, rng.uniform(0.1, 1.2, n), rng.uniform(0.01, 0.1, n))
        else:
            ntk[3 * m + 1] = 10 ** rng.uniform(-2, 1.5, n)
            c = rng.integers(0, 4, n)
            ntk[3 * m + 2] = np.select([c == 0, c == 1, c == 2, c == 3],
                                       [rng.uniform(0.05, 1.0, n), rng.uniform(1, 10, n), 10 ** rng.uniform(-3, -1, n),
                                        np.ones(n)])
    return np.ascontiguousarray(ntk), types


def run(ntk, P, N=3, gam=4.0 / 3.0, edge_cost=0.6, types=None):
    n = ntk.shape[1]
    T = np.zeros((3 * (N - 1), n))
    cost = np.zeros((N - 1, n))
    st = Stats()
    dp = C.POINTER(C.c_double)
    ty = (C.c_int * N)(*(types or [0] * N))
    which = 1 if isinstance(P, GradParams) else 0
    lib().lab_batch(which, N, ty, C.c_long(n), ntk.ctypes.data_as(dp), C.c_double(gam), C.byref(P), T.ctypes.data_as(dp),
                    cost.ctypes.data_as(dp), C.c_double(edge_cost), C.byref(st))
    return T, cost, st


def scales(ntk, N=3, gam=4.0 / 3.0):
    out = []
    for j in range(N - 1):
        th, k = ntk[3 * j + 1], ntk[3 * j + 2]
        A = 2 * k + gam
        out += [np.ones_like(A), A * th, A * (A + 1) * th * th]
    return np.array(out)


def report(name, ntk, P, ref, N=3, types=None, gam=4.0 / 3.0):
    T, cost, st = run(ntk, P, N, gam=gam, types=types)
    sc = scales(ntk, N, gam)
    with np.errstate(invalid="ignore", divide="ignore"):
        e_sc = np.abs(T - ref) / sc
        e_rel = np.abs(T - ref) / np.maximum(np.abs(ref), 1e-10 * sc)
    n = ntk.shape[1]
    per = cost.sum(0)
    nw = n // 64
    wave = sum(cost[j, :nw * 64].reshape(nw, 64).max(1).sum() for j in range(N - 1)) / nw
    merged = per[:nw * 64].reshape(nw, 64).max(1).mean()
    print(f"{name:38s} cost {per.mean():6.1f} nodes {st.nodes / n:6.1f} wave-max {wave:6.1f} (lanes {per.mean() / wave:.2f}; merged {per.mean() / merged:.2f})"
          f" evals {st.evals / n:5.1f} rej {st.rejects / n:4.1f} skip {st.skipped / n:4.1f} init {st.init_panels / n:5.1f} edges {st.edges / n:5.1f}"
          f" | err/scale {np.nanmax(e_sc):.1e} rel max {np.nanmax(e_rel):.1e} p99.9 {np.nanpercentile(e_rel, 99.9):.1e} >1e-9: {(e_rel > 1e-9).any(0).mean() * 100:.2f}%",
          flush=True)
    return T, cost
