"""ctypes binding of liblbhip.so (C ABI: include/lb_hip.h).

The library is the only compute backend.  If it is missing or no GPU is visible the
functions raise; nothing falls back to the CPU."""
import ctypes as ct
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LB_LIB") or os.path.join(_HERE, "liblbhip.so")   # LB_LIB: diagnostic builds only

LB_BC_PIPE, LB_BC_PERIODIC, LB_BC_CAVITY, LB_BC_VELOCITY_INLET = 0, 1, 2, 3
LB_FLAG_HALO = 1
LB_FLAG_PLANAR = 2
LB_FLAG_EAGER_MACRO = 4
LB_MASK_HALO_ROWS = 13
LB_PEER_HANDLE_BYTES = 384
LB_DEVICE_CPU = -1
LB_SEM_OPENCL, LB_SEM_CYTHON, LB_SEM_OPENCL_D2Q9I = 0, 1, 2
BC_NAMES = {"pipe": LB_BC_PIPE, "periodic": LB_BC_PERIODIC, "cavity": LB_BC_CAVITY,
            "velocity_inlet": LB_BC_VELOCITY_INLET}

ABI_VERSION = 10

# every symbol include/lb_hip.h declares (checked by tests/test_abi.py)
EXPORTS = (
    "lb_abi_version", "lb_device_count", "lb_last_error", "lb_create", "lb_destroy", "lb_sync", "lb_set_stream",
    "lb_set_macro", "lb_get_macro", "lb_set_f", "lb_get_f", "lb_get_feq", "lb_set_mask",
    "lb_move", "lb_move_bcs", "lb_update_hydro", "lb_update_feq", "lb_collide_particles",
    "lb_zero_velocity_in_obstacle", "lb_init_pop", "lb_run",
    "lb_step_boundary", "lb_step_interior", "lb_step_finish", "lb_halo_export", "lb_halo_import",
    "lb_halo_floats", "lb_set_mask_halo", "lb_run_group", "lb_run_batch",
    "lb_comm_available", "lb_comm_unique_id", "lb_comm_init", "lb_timer_start", "lb_timer_stop", "lb_layout", "lb_set_variant", "lb_copy_calibration", "lb_steps_per_launch", "lb_plan_launches", "lb_autotune",
    "lb_autotune_quick", "lb_hot_kernel", "lb_get_corner_state", "lb_set_corner_state", "lb_check", "lb_set_debug_sync",
    "lb_peer_export", "lb_peer_connect", "lb_set_params_f64", "lb_set_slab_cycle", "lb_exchange_timing", "lb_exchange_stats",
    "lb_set_exchange_inline",
)


class LbParams(ct.Structure):
    _fields_ = [("nx", ct.c_int32), ("ny", ct.c_int32), ("y0", ct.c_int32), ("local_ny", ct.c_int32),
                ("bc_mode", ct.c_int32), ("device", ct.c_int32),
                ("omega", ct.c_float), ("inlet_rho", ct.c_float), ("outlet_rho", ct.c_float),
                ("lid_u", ct.c_float), ("rho0", ct.c_float), ("flags", ct.c_int32), ("semantics", ct.c_int32),
                ("inlet_u", ct.c_float), ("outlet_u", ct.c_float), ("reserved", ct.c_int32 * 1)]


class LbError(RuntimeError):
    pass


_lib = None


def lib():
    """Load liblbhip.so once.  Raises LbError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LbError("%s not found: build it with `python 2d-lb_amd/build.py` "
                      "(the engine has no CPU fallback)" % LIB_PATH)
    L = ct.CDLL(LIB_PATH)
    h, fp, vp, I = ct.c_void_p, ct.POINTER(ct.c_float), ct.c_void_p, ct.c_int
    L.lb_last_error.restype = ct.c_char_p
    L.lb_create.argtypes = [ct.POINTER(LbParams), ct.POINTER(h)]
    for name in ("lb_destroy", "lb_sync", "lb_move", "lb_move_bcs", "lb_update_hydro", "lb_update_feq",
                 "lb_collide_particles", "lb_zero_velocity_in_obstacle", "lb_init_pop", "lb_step_finish",
                 "lb_timer_start", "lb_steps_per_launch", "lb_autotune"):
        getattr(L, name).argtypes = [h]
    L.lb_plan_launches.argtypes = [h, I, ct.POINTER(ct.c_int), I]
    L.lb_set_stream.argtypes = [h, vp]
    L.lb_set_macro.argtypes = [h, vp, vp, vp]
    L.lb_get_macro.argtypes = [h, vp, vp, vp]
    L.lb_set_f.argtypes = [h, vp]
    L.lb_get_f.argtypes = [h, vp]
    L.lb_get_feq.argtypes = [h, vp]
    L.lb_set_mask.argtypes = [h, vp]
    L.lb_run.argtypes = [h, I]
    L.lb_step_boundary.argtypes = [h, I]
    L.lb_step_interior.argtypes = [h, I]
    L.lb_halo_export.argtypes = [h, I, vp]
    L.lb_halo_import.argtypes = [h, I, vp]
    L.lb_halo_floats.argtypes = [h]
    L.lb_set_mask_halo.argtypes = [h, vp, vp]
    L.lb_run_group.argtypes = [ct.POINTER(h), I, I]
    L.lb_run_batch.argtypes = [ct.POINTER(h), I, I]
    L.lb_comm_unique_id.argtypes = [vp]
    # (an older diagnostic build selected with LB_LIB -- A/B timing of kernels across rounds -- lacks the newest entry points)
    older = bool(os.environ.get("LB_LIB")) and L.lb_abi_version() < ABI_VERSION
    if not older:
        L.lb_peer_export.argtypes = [h, vp]
        L.lb_peer_connect.argtypes = [h, I, I, vp, vp, I]
        L.lb_set_params_f64.argtypes = [h, ct.c_double, ct.c_double, ct.c_double]
    if L.lb_abi_version() >= 9:
        L.lb_set_slab_cycle.argtypes = [h, I]
        L.lb_exchange_timing.argtypes = [h, I]
        L.lb_exchange_stats.argtypes = [h, ct.POINTER(ct.c_int64), ct.POINTER(ct.c_double), ct.POINTER(ct.c_double),
                                        ct.POINTER(ct.c_int), ct.POINTER(ct.c_int)]
    if L.lb_abi_version() >= 10:
        L.lb_set_exchange_inline.argtypes = [h, I]
    L.lb_comm_init.argtypes = [h, vp, I, I]
    L.lb_timer_stop.argtypes = [h, fp]
    L.lb_layout.argtypes = [h, ct.POINTER(ct.c_int64), ct.POINTER(ct.c_int64), ct.POINTER(ct.c_int64)]
    L.lb_set_variant.argtypes = [h, I]
    L.lb_copy_calibration.argtypes = [h, I, ct.POINTER(ct.c_int64)]
    L.lb_autotune_quick.argtypes = [h, I]
    L.lb_get_corner_state.argtypes = [h, vp]
    L.lb_set_corner_state.argtypes = [h, vp]
    L.lb_hot_kernel.argtypes = [h, ct.c_char_p, I]
    L.lb_check.argtypes = [h, I, ct.POINTER(ct.c_int64), fp, ct.POINTER(ct.c_double)]
    L.lb_set_debug_sync.argtypes = [I]
    if L.lb_abi_version() != ABI_VERSION and not older:
        raise LbError("liblbhip.so ABI %d != binding ABI %d: rebuild" % (L.lb_abi_version(), ABI_VERSION))
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise LbError("liblbhip: %s (status %d)" % (lib().lb_last_error().decode(), rc))


def device_count():
    n = lib().lb_device_count()
    if n < 0:
        raise LbError("liblbhip: %s" % lib().lb_last_error().decode())
    return n
