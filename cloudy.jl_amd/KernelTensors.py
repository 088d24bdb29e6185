"""CoalescenceTensor: symmetric P x P polynomial coefficients c[a, b] of x^a y^b (src/Kernels/KernelTensors.jl).

The reference fits non-polynomial kernels with an un-vendored Nelder-Mead optimiser (KernelTensors.jl:78-146);
those coefficients are only reproducible to ~1e-5, so fitted tensors are INPUTS here.  The
constructor from a kernel function is provided for the kernels that are polynomials themselves (constant,
linear, and each branch of Long's kernel), where the least-squares fit is exact; the forced constant term
C_1_1 = max(eps, K(0, 0)) (KernelTensors.jl:115) is reproduced.
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
            self.c = _exact_polynomial_tensor(c_or_kernel, int(order), float(limit), float(lower_limit))

    @property
    def P(self):
        return self.c.shape[0]

    def __repr__(self):
        return f"CoalescenceTensor({self.c.tolist()})"


def _exact_polynomial_tensor(kernel_func, order, limit, lower_limit):
    if limit <= lower_limit or lower_limit < 0:
        raise ValueError("polyfit limits improperly specified")
    P = order + 1
    c = np.zeros((P, P))
    if isinstance(kernel_func, ConstantKernelFunction):
        c[0, 0] = max(EPS, kernel_func.coll_coal_rate)
        return c
    if isinstance(kernel_func, LinearKernelFunction):
        if order < 1:
            raise ValueError("a linear kernel needs order >= 1")
        c[0, 0] = EPS  # C_1_1 = max(eps, K(0,0)) with K(0,0) = 0
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
        c[0, 0] = EPS  # C_1_1 = max(eps, K(0, 0)), K(0, 0) = 0
        if above == 0:
            c[0, 2] = c[2, 0] = kernel_func.coal_rate_below_threshold
        elif below <= 1:  # only the (0, 0) sample can sit below the kink, where both branches vanish
            c[0, 1] = c[1, 0] = kernel_func.coal_rate_above_threshold
        else:
            raise NotImplementedError("fit window straddles Long's threshold: needs the polyfit (out of scope, F6)")
        return c
    raise NotImplementedError(
        "least-squares polyfit of a non-polynomial kernel (KernelTensors.jl:78-146, Optim.jl) is out of scope: "
        "pass the fitted coefficient matrix as CoalescenceTensor(c)")


def get_normalized_kernel_tensor(kernel, norms):
    """KernelTensors.jl:189-199: c[i, j] * norms[1] * norms[2]^(i+j-2) (1-based i, j)."""
    P = kernel.P
    c = np.array([[kernel.c[i, j] * (norms[0] * norms[1] ** float(i + j)) for j in range(P)] for i in range(P)])
    return CoalescenceTensor(c)
