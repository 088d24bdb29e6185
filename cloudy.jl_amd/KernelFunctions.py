"""Closed-form collision kernels K(x, y) (src/Kernels/KernelFunctions.jl).  Host-side, plan-time only:
they feed CoalescenceTensor construction and carry the unit normalisation."""
import math
from dataclasses import dataclass


class KernelFunction:
    pass


class CoalescenceKernelFunction(KernelFunction):
    pass


@dataclass(frozen=True)
class ConstantKernelFunction(CoalescenceKernelFunction):
    coll_coal_rate: float

    def __call__(self, x, y):  # KernelFunctions.jl:94-96
        return self.coll_coal_rate


@dataclass(frozen=True)
class LinearKernelFunction(CoalescenceKernelFunction):
    coll_coal_rate: float

    def __call__(self, x, y):  # :98-100
        return self.coll_coal_rate * (x + y)


@dataclass(frozen=True)
class HydrodynamicKernelFunction(CoalescenceKernelFunction):
    coal_eff: float

    def __call__(self, x, y):  # :102-108
        r1 = (3 / 4 / math.pi * x) ** (1 / 3)
        r2 = (3 / 4 / math.pi * y) ** (1 / 3)
        A1 = math.pi * r1**2
        A2 = math.pi * r2**2
        return self.coal_eff * (r1 + r2) ** 2 * abs(A1 - A2)


@dataclass(frozen=True)
class LongKernelFunction(CoalescenceKernelFunction):
    x_threshold: float
    coal_rate_below_threshold: float
    coal_rate_above_threshold: float

    def __call__(self, x, y):  # :110-116
        if x < self.x_threshold and y < self.x_threshold:
            return self.coal_rate_below_threshold * (x**2 + y**2)
        return self.coal_rate_above_threshold * (x + y)


def get_normalized_kernel_func(kern, norms):
    """KernelFunctions.jl:124-154."""
    if isinstance(kern, ConstantKernelFunction):
        return ConstantKernelFunction(kern.coll_coal_rate * norms[0])
    if isinstance(kern, LinearKernelFunction):
        return LinearKernelFunction(kern.coll_coal_rate * norms[0] * norms[1])
    if isinstance(kern, HydrodynamicKernelFunction):
        return HydrodynamicKernelFunction(kern.coal_eff * norms[0] * norms[1] ** (4 / 3))
    if isinstance(kern, LongKernelFunction):
        return LongKernelFunction(kern.x_threshold / norms[1], kern.coal_rate_below_threshold * norms[0] * norms[1] ** 2,
                                  kern.coal_rate_above_threshold * norms[0] * norms[1])
    raise TypeError("unknown kernel function")
