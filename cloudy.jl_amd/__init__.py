"""cloudy.jl_amd -- host-side mirror of the Cloudy.jl interface for the batched collision-coalescence
moment RHS, over the C ABI of libcloudy_hip.so (gfx950 HIP kernels).  Loaded as module `cloudy_jl_amd`
by __graft_entry__.load_package() (the directory name is not a Python identifier).

Names follow the reference modules (src/Cloudy.jl:1-12): KernelFunctions, KernelTensors,
ParticleDistributions, EquationTypes, Coalescence, Sedimentation, helper functions; plus the box-model
RHS factory of test/examples/utils/box_model_helpers.jl.  There is no CPU compute path in this package.
"""
from . import _lib
from ._lib import CloudyError, device_count, lib
from .device import DeviceArray, plane_dtype
from .EquationTypes import (AbstractStyle, AnalyticalCoalStyle, CoalescenceStyle, FixedThreshold, MovingThreshold,
                            NumericalCoalStyle, ThresholdStyle)
from .helper_functions import (get_dist_moment_ind, get_dist_moments_ind_range, get_moments_normalizing_factors,
                               rflatten)
from .KernelFunctions import (ConstantKernelFunction, HydrodynamicKernelFunction, LinearKernelFunction,
                              LongKernelFunction, get_normalized_kernel_func)
from .KernelTensors import CoalescenceTensor, check_symmetry, get_normalized_kernel_tensor, polyfit
from .ParticleDistributions import (ExponentialPrimitiveParticleDistribution, GammaPrimitiveParticleDistribution,
                                    LognormalPrimitiveParticleDistribution, MonodispersePrimitiveParticleDistribution,
                                    compute_thresholds, get_moments, get_standard_N_q, nparams, pack_params, update_dist_from_moments,
                                    check_moment_consistency, closure_stats, CLOSURE_STATS_FIELDS)
from .Coalescence import (CoalescenceData, NumericalPlan, Plan, get_coal_ints, get_finite_2d_integrals,
                          kernel_func_code, numerical_plan, QUAD_CONVERGED, QUAD_FIXED, F64, F32, F32_FAST,
                          F64_RELAXED)
from .Sedimentation import (get_sedimentation_flux, make_rainshaft_rhs, rainshaft_sources, rhs_condensation,
                            solve_rainshaft_ssprk33)
from .box_model import ODEParameters, make_box_model_rhs, rhs_coal, solve_ssprk33, solve_tsit5
from .sharding import Communicator, allreduce_sums, mode_sums, moment_sums, shard_range

__all__ = [n for n in dir() if not n.startswith("_")]
