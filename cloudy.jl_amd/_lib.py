"""ctypes binding of libcloudy_hip.so (include/cloudy_hip.h).

The product path has no CPU fallback: if the HIP library is missing, or a batched call is made
without a GPU, this module raises -- it never routes through oracle/ or numpy.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CLOUDY_HIP_LIB selects another build of the SAME C ABI (kernel A/B experiments); never a CPU library.
LIB_PATH = os.environ.get("CLOUDY_HIP_LIB") or os.path.join(_HERE, "libcloudy_hip.so")

MAX_MODES, MAX_P, MAX_VEL = 8, 8, 4
OK, EINVAL, ENOTSYMMETRIC, EHIP, ENOMEM, EUNSUPPORTED, ENODEVICE, ECOMM = 0, -1, -2, -3, -4, -5, -6, -7
COMM_ID_BYTES = 128


class CloudyError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libcloudy_hip error {code}: {msg}")
        self.code = code
        self.msg = msg


class PlanDesc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_modes", C.c_int32),
        ("dist_type", C.c_int32 * MAX_MODES),
        ("tensor_p", C.c_int32),
        ("kernel_layout", C.c_int32),
        ("kernel_is_normalized", C.c_int32),
        ("kernel_c", C.POINTER(C.c_double)),
        ("dist_thresholds", C.c_double * MAX_MODES),
        ("threshold_style", C.c_int32),
        ("norms", C.c_double * 2),
        ("k_range", C.c_double * 2),
        ("n_bins_per_log_unit", C.c_int32),
        ("dtype", C.c_int32),
        ("n_vel", C.c_int32),
        ("vel", C.c_double * (MAX_VEL * 2)),
        ("device", C.c_int32),
        ("specialize", C.c_int32),
        ("coal_style", C.c_int32),
        ("kernel_func", C.c_int32),
        ("kernel_func_is_normalized", C.c_int32),
        ("quad_order", C.c_int32),
        ("kernel_func_params", C.c_double * 3),
        ("thresholds_are_normalized", C.c_int32),
        ("quad_mode", C.c_int32),
    ]


# every symbol include/cloudy_hip.h declares: (restype, argtypes)
_vp, _sz, _i, _dp = C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)
SYMBOLS = {
    "cloudy_plan_desc_init": (None, [C.POINTER(PlanDesc)]),
    "cloudy_plan_create": (_i, [C.POINTER(PlanDesc), C.POINTER(_vp)]),
    "cloudy_plan_destroy": (None, [_vp]),
    "cloudy_plan_specialized": (_i, [_vp]),
    "cloudy_plan_jit_log": (C.c_char_p, [_vp]),
    "cloudy_jit_selfcheck": (_i, [C.POINTER(PlanDesc), C.c_char_p]),
    "cloudy_plan_desc_layout": (_i, [C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _i]),
    "cloudy_quad_rule_host": (_i, [_i, C.c_double, C.c_double, _dp, _dp]),
    "cloudy_plan_nmom": (_i, [_vp]),
    "cloudy_plan_device": (_i, [_vp]),
    "cloudy_plan_nparams": (_i, [_vp]),
    "cloudy_plan_get": (_i, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _dp, _dp, _dp]),
    "cloudy_coal_rhs": (_i, [_vp, _sz, _sz, _vp, _vp, _vp]),
    "cloudy_coal_rhs_host": (_i, [_vp, _sz, _sz, _vp, _vp]),
    "cloudy_get_coal_ints": (_i, [_vp, _sz, _sz, _vp, _vp, _vp]),
    "cloudy_ssprk33_steps": (_i, [_vp, _sz, _sz, _vp, _vp, C.c_double, _i, _vp]),
    "cloudy_tsit5_steps": (_i, [_vp, _sz, _sz, _vp, _vp, C.c_double, _i, _vp]),
    "cloudy_update_dist_from_moments": (_i, [_vp, _sz, _sz, _vp, _vp, _vp]),
    "cloudy_closure_stats": (_i, [_vp, _sz, _sz, _vp, C.POINTER(C.c_uint64), _vp]),
    "cloudy_finite_2d_integrals": (_i, [_vp, _sz, _sz, _vp, _vp, _vp]),
    "cloudy_compute_thresholds": (_i, [_vp, _sz, _sz, _vp, _vp, _vp]),
    "cloudy_sedimentation_flux": (_i, [_vp, _sz, _sz, _vp, _vp, _vp]),
    "cloudy_cond_evap": (_i, [_vp, _sz, _sz, _vp, _vp, C.c_double, C.c_double, _vp, _vp]),
    "cloudy_standard_N_q": (_i, [_vp, _sz, _sz, _vp, C.c_double, _vp, _vp]),
    "cloudy_rainshaft_sources": (_i, [_vp, _sz, _sz, _vp, _vp, _vp, _vp]),
    "cloudy_rainshaft_rhs": (_i, [_vp, _sz, _sz, _sz, _vp, C.c_double, _vp, _vp, _vp]),
    "cloudy_rainshaft_ssprk33_steps": (_i, [_vp, _sz, _sz, _sz, _vp, _vp, C.c_double, C.c_double, C.c_int, _vp]),
    "cloudy_moment_sums": (_i, [_vp, _sz, _sz, _i, _vp, _vp, _vp]),
    "cloudy_moment_sums_workspace_bytes": (_sz, [_i]),
    "cloudy_moment_sums_ws": (_i, [_vp, _sz, _sz, _i, _vp, _vp, _vp, _sz, _vp]),
    "cloudy_comm_rccl_version": (_i, []),
    "cloudy_comm_unique_id": (_i, [_vp]),
    "cloudy_comm_create": (_i, [_i, _i, _vp, _i, C.POINTER(_vp)]),
    "cloudy_comm_create_all": (_i, [_i, C.POINTER(C.c_int), C.POINTER(_vp)]),
    "cloudy_comm_destroy": (None, [_vp]),
    "cloudy_comm_rank": (_i, [_vp]),
    "cloudy_comm_world_size": (_i, [_vp]),
    "cloudy_comm_device": (_i, [_vp]),
    "cloudy_comm_group_start": (_i, []),
    "cloudy_comm_group_end": (_i, []),
    "cloudy_allreduce_sum_f64": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "cloudy_moment_sums_allreduce": (_i, [_vp, _vp, _sz, _sz, _i, _vp, _vp, _vp]),
    "cloudy_device_count": (_i, []),
    "cloudy_set_device": (_i, [_i]),
    "cloudy_device_pci_bus_id": (_i, [_i, C.c_char_p, _i]),
    "cloudy_malloc": (_i, [C.POINTER(_vp), _sz]),
    "cloudy_free": (_i, [_vp]),
    "cloudy_memcpy_h2d": (_i, [_vp, _vp, _sz, _vp]),
    "cloudy_memcpy_d2h": (_i, [_vp, _vp, _sz, _vp]),
    "cloudy_memset": (_i, [_vp, _i, _sz, _vp]),
    "cloudy_stream_synchronize": (_i, [_vp]),
    "cloudy_time_coal_rhs": (_i, [_vp, _sz, _sz, _vp, _vp, _vp, _i, C.POINTER(C.c_float)]),
    "cloudy_timer_begin": (_i, [_vp, C.POINTER(_vp)]),
    "cloudy_timer_end": (_i, [_vp, _vp, C.POINTER(C.c_float)]),
    "cloudy_last_error": (C.c_char_p, []),
    "cloudy_version": (_i, []),
    "cloudy_source_hash": (C.c_ulonglong, [_i]),
}

_lib = None


def lib():
    """The loaded library; raises if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the coalescence RHS.")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)  # AttributeError if the ABI lost a symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != OK:
        raise CloudyError(rc, lib().cloudy_last_error().decode("utf-8", "replace"))
    return rc


def device_count():
    return lib().cloudy_device_count()
