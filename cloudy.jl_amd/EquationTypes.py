"""Dispatch tags, mirroring src/Sources/EquationTypes.jl:15-22."""


class AbstractStyle:
    pass


class CoalescenceStyle(AbstractStyle):
    pass


class NumericalCoalStyle(CoalescenceStyle):
    """Integrals of a kernel FUNCTION over the densities (Coalescence.jl:470-708).  The reference nests adaptive quadgk;
    the device path evaluates every integral by one fixed `quad_order`-point Gauss rule per distribution
    (csrc/quad.hpp; the order is a field of the ODE parameters, default 10)."""


class AnalyticalCoalStyle(CoalescenceStyle):
    pass


class ThresholdStyle:
    pass


class MovingThreshold(ThresholdStyle):
    pass


class FixedThreshold(ThresholdStyle):
    pass
