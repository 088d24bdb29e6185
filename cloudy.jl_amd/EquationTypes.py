"""Dispatch tags, mirroring src/Sources/EquationTypes.jl:15-22."""


class AbstractStyle:
    pass


class CoalescenceStyle(AbstractStyle):
    pass


class NumericalCoalStyle(CoalescenceStyle):
    """Integrals of a kernel FUNCTION over the densities (Coalescence.jl:470-708).  The reference nests adaptive quadgk;
    the device path splits the integrals along the kernel function's non-smooth sets -- closed forms plus one adaptive
    Gauss-Kronrod rule per mode (csrc/quad_conv.hpp, the default: within 1e-8 of the reference's answer); a fixed
    `quad_order`-point Gauss rule per distribution (csrc/quad.hpp) is the opt-in `quad_mode = QUAD_FIXED` field of the
    ODE parameters."""


class AnalyticalCoalStyle(CoalescenceStyle):
    pass


class ThresholdStyle:
    pass


class MovingThreshold(ThresholdStyle):
    pass


class FixedThreshold(ThresholdStyle):
    pass
