"""CoalescenceTensor: symmetric P x P polynomial coefficients c[a, b] of x^a y^b (src/Kernels/KernelTensors.jl).

The reference fits kernels with an un-vendored Nelder-Mead optimiser (KernelTensors.jl:78-146); its coefficients are
only reproducible to ~1e-5, so for parity fitted tensors are INPUTS (CoalescenceTensor(c)).  The
constructor from a kernel function returns the exact coefficients for kernels that are polynomials themselves
(constant, linear, each branch of Long's kernel) and otherwise solves the reference's least-squares problem
directly (`polyfit`); the forced constant term C_1_1 = max(eps, K(0, 0)) (KernelTensors.jl:115) is reproduced.
"""
import numpy as np

from .KernelFunctions import (ConstantKernelFunction, LinearKernelFunction, LongKernelFunction)

EPS = float(np.finfo(np.float64).eps)


def check_symmetry(array):
    """KernelTensors.jl:157-171 (exact comparison)."""
    a = np.asarray(array, dtype=np.float64)
    if a.size > 1:
        if a.ndim != 2 or a.shape[0] != a.shape[1]:
            raise ValueError("array needs to be quadratic in order to be symmetric.")
        if not np.array_equal(a, a.T):
            raise ValueError("array not symmetric.")


class CoalescenceTensor:
    """CoalescenceTensor(c) or CoalescenceTensor(kernel_func, order, limit[, lower_limit, norms])."""

    def __init__(self, c_or_kernel, order=None, limit=None, lower_limit=0.0, norms=(1e6, 1e-9)):
        if order is None:
            c = np.array(c_or_kernel, dtype=np.float64, ndmin=2)
            check_symmetry(c)
            self.c = c
        else:
            self.c = _exact_polynomial_tensor(c_or_kernel, int(order), float(limit), float(lower_limit), norms)

    @property
    def P(self):
        return self.c.shape[0]

    def __repr__(self):
        return f"CoalescenceTensor({self.c.tolist()})"


def polyfit(kernel_func, r, limit, lower_limit=0.0, norms=(1e6, 1e-9), npoints=10):
    """Least-squares fit of a symmetric order-r polynomial to a kernel function on the reference's sample set
    (KernelTensors.jl:78-146): points (x, y) of the npoints x npoints grid with y >= lower_limit and y >= x, the
    residual taken over ALL pairs (x_p, y_q) of the kept coordinates, the constant term forced to
    C_1_1 = max(eps, K(0, 0)).  The reference minimises the Frobenius norm with Optim's Nelder-Mead
    (g_abstol = sqrt(eps)) and is only reproducible to ~1e-5; this solves the same linear least-squares problem
    directly, so fitted tensors agree with the reference's to its own optimiser tolerance, not bit for bit
    (outside the parity surface).  Plan-time host code."""
    from .KernelFunctions import CoalescenceKernelFunction, get_normalized_kernel_func

    if isinstance(kernel_func, CoalescenceKernelFunction):
        kf = get_normalized_kernel_func(kernel_func, norms)
    else:
        kf = kernel_func
        norms = (1.0, 1.0)
    limit_n, lower_n = limit / norms[1], lower_limit / norms[1]
    if limit_n <= lower_n or lower_n < 0:
        raise ValueError("polyfit limits improperly specified")
    rng = np.random.default_rng(0)  # check_symmetry(FT, func), KernelTensors.jl:173-181
    for a, b in rng.random((1000, 2)):
        if abs(kf(a, b) - kf(b, a)) > 1e-6:
            raise ValueError("function likely not symmetric.")
    delta = limit_n / (npoints - 1)
    idx = np.arange(npoints * npoints)
    x_, y_ = (idx % npoints) * delta, np.floor(idx / npoints) * delta
    keep = (y_ >= lower_n) & (y_ - x_ >= 0)
    x, y = x_[keep], y_[keep]
    C11 = max(EPS, kf(0.0, 0.0))
    P = r + 1
    if r == 0:
        return np.array([[C11 / norms[0]]])
    X, Y = np.meshgrid(x, y, indexing="ij")  # all pairs (x_p, y_q)
    target = np.vectorize(kf)(X, Y) - C11
    cols, where = [], []
    for i in range(P):
        for j in range(i, P):
            if i == 0 and j == 0:
                continue
            basis = X**i * Y**j + (X**j * Y**i if i != j else 0.0)
            cols.append(basis.ravel())
            where.append((i, j))
    A = np.stack(cols, axis=1)
    scale = np.linalg.norm(A, axis=0)
    sol = np.linalg.lstsq(A / scale, target.ravel(), rcond=None)[0] / scale
    c = np.zeros((P, P))
    c[0, 0] = C11
    for (i, j), v in zip(where, sol):
        c[i, j] = c[j, i] = v
    for i in range(P):
        for j in range(P):
            c[i, j] /= norms[0] * norms[1] ** float(i + j)
    return c


def _exact_polynomial_tensor(kernel_func, order, limit, lower_limit, norms=(1e6, 1e-9)):
    if limit <= lower_limit or lower_limit < 0:
        raise ValueError("polyfit limits improperly specified")
    P = order + 1
    c = np.zeros((P, P))
    # the forced constant term is set in NORMALISED units and then de-normalised (KernelTensors.jl:115, 141-145):
    # C_1_1 = max(eps, K_n(0, 0)) / norms[1]
    if isinstance(kernel_func, ConstantKernelFunction):
        c[0, 0] = max(EPS, kernel_func.coll_coal_rate * norms[0]) / norms[0]
        return c
    if isinstance(kernel_func, LinearKernelFunction):
        if order < 1:
            raise ValueError("a linear kernel needs order >= 1")
        c[0, 0] = EPS / norms[0]  # K(0, 0) = 0
        c[0, 1] = c[1, 0] = kernel_func.coll_coal_rate
        return c
    if isinstance(kernel_func, LongKernelFunction):
        if order < 2:
            raise ValueError("Long's kernel needs order >= 2")
        # the sample grid of polyfit (KernelTensors.jl:100-109): 0 <= x <= y, lower_limit <= y <= limit
        npoints = 10
        delta = limit / (npoints - 1)
        below = above = 0
        for i in range(npoints * npoints):
            x, y = (i % npoints) * delta, float(i // npoints) * delta
            if y >= lower_limit and y - x >= 0:
                if x < kernel_func.x_threshold and y < kernel_func.x_threshold:
                    below += 1
                else:
                    above += 1
        c[0, 0] = EPS / norms[0]  # K(0, 0) = 0
        if above == 0:
            c[0, 2] = c[2, 0] = kernel_func.coal_rate_below_threshold
        elif below <= 1:  # only the (0, 0) sample can sit below the kink, where both branches vanish
            c[0, 1] = c[1, 0] = kernel_func.coal_rate_above_threshold
        else:  # the fit window straddles the kink: genuine least-squares fit
            return polyfit(kernel_func, order, limit, lower_limit, norms)
        return c
    return polyfit(kernel_func, order, limit, lower_limit, norms)


def get_normalized_kernel_tensor(kernel, norms):
    """KernelTensors.jl:189-199: c[i, j] * norms[1] * norms[2]^(i+j-2) (1-based i, j)."""
    P = kernel.P
    c = np.array([[kernel.c[i, j] * (norms[0] * norms[1] ** float(i + j)) for j in range(P)] for i in range(P)])
    return CoalescenceTensor(c)
