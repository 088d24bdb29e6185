"""Dispatch tags, mirroring src/Sources/EquationTypes.jl:15-22."""


class AbstractStyle:
    pass


class CoalescenceStyle(AbstractStyle):
    pass


class NumericalCoalStyle(CoalescenceStyle):
    """Nested adaptive quadrature (Coalescence.jl:470-708): not built for the GPU (out of scope)."""


class AnalyticalCoalStyle(CoalescenceStyle):
    pass


class ThresholdStyle:
    pass


class MovingThreshold(ThresholdStyle):
    pass


class FixedThreshold(ThresholdStyle):
    pass
