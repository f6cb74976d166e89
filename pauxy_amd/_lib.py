"""ctypes binding of libafqmc_hip.so (include/afqmc_hip.h).

The library is the product: there is no CPU fallback.  Importing this module
never touches the GPU; the first call that needs the library loads it and
raises ``AfqLibraryError`` if it is missing or does not export the full C ABI.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
# AFQ_LIBRARY selects another build of the same C ABI (the tuning build of `make TUNING=1`, tools/ab_bench.sh)
LIB_PATH = os.environ.get("AFQ_LIBRARY") or os.path.join(HERE, "libafqmc_hip.so")

AFQ_OK = 0
AFQ_EWEIGHT, AFQ_EOVERFLOW, AFQ_ECOMM = -6, -7, -8
AFQ_COMM_NSTATS = 12
# the caller's all-gather lent to afq_comm_init_ipc: int (*)(const void *send, void *recv, int bytes, void *user)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p)
AFQ_SYS_GENERIC, AFQ_SYS_HUBBARD, AFQ_SYS_UEG = 1, 2, 3
AFQ_PROP_HYBRID, AFQ_PROP_FORCE_BIAS, AFQ_PROP_FREE_PROJECTION, AFQ_PROP_HUBBARD_SPIN = 1, 2, 4, 8
(F_PHI, F_WEIGHT, F_UNSCALED_WEIGHT, F_OT, F_HYBRID_ENERGY, F_PHASE, F_DETR, F_ELOC, F_GHALF, F_G,
 F_XBAR, F_XSHIFTED, F_ENERGY, F_LOG_DETR) = range(14)

_h = c_void_p
_dp = c_void_p        # const double* / void* passed as raw addresses
_ip = c_void_p

# name -> argtypes (restype is int unless noted); mirrors include/afqmc_hip.h one to one
SIGNATURES = {
    "afq_version": [],
    "afq_create": [c_int, POINTER(_h)],
    "afq_destroy": [_h],
    "afq_last_error": [_h],
    "afq_sync": [_h],
    "afq_set_system_generic": [_h, c_int, c_int, c_int, c_int, _dp, _dp, _dp, c_double],
    "afq_set_system_hubbard": [_h, c_int, c_int, c_int, c_double, _dp],
    "afq_set_system_ueg": [_h, c_int, c_int, c_int, c_int, _ip, _ip, _dp, _ip, _ip, _dp,
                           _ip, _ip, _ip, _ip, _ip, _ip, _dp, c_double, _dp, c_double],
    "afq_set_trial": [_h, _dp],
    "afq_set_propagator": [_h, _dp, _dp, c_double, c_int, c_int],
    "afq_walkers_alloc": [_h, c_int],
    "afq_walkers_set": [_h, c_int, c_void_p, c_int, c_int],
    "afq_walkers_get": [_h, c_int, c_void_p, c_int, c_int],
    "afq_walkers_device_ptr": [_h, c_int, POINTER(c_void_p), POINTER(c_int64)],
    "afq_greens": [_h, c_int, _dp],
    "afq_calc_overlap": [_h, _dp],
    "afq_propagate": [_h, _dp, c_double, c_double],
    "afq_propagate_begin": [_h, _dp],
    "afq_propagate_finish": [_h, c_double, c_double],
    "afq_reortho": [_h, _dp],
    "afq_set_log_shift": [_h, c_int, c_double, c_double],
    "afq_log_ovlp_sums": [_h, _dp],
    "afq_local_energy": [_h, _dp],
    "afq_force_bias": [_h, _dp],
    "afq_shift_fields": [_h, _dp, _dp, _dp, _dp, _dp],
    "afq_vhs": [_h, _dp, _dp],
    "afq_vhs_count": [_h, POINTER(c_int)],
    "afq_apply_exponential": [_h, _dp],
    "afq_kinetic": [_h],
    "afq_cap_weights": [_h, c_double, c_double],
    "afq_set_weight_cap": [_h, c_double, c_double],
    "afq_popcontrol_comb": [_h, c_double, c_double, c_void_p, POINTER(c_double)],
    "afq_walkers_scale_weights": [_h, c_double],
    "afq_walkers_copy": [_h, c_int, c_int],
    "afq_walker_pack_bytes": [_h, POINTER(c_int64)],
    "afq_walker_pack": [_h, c_int, c_void_p],
    "afq_walker_unpack": [_h, c_int, c_void_p],
    "afq_walkers_reset_weights": [_h],
    "afq_estimates_update": [_h, c_int],
    "afq_estimates_get": [_h, _dp, c_int],
    "afq_estimates_fuse_next": [_h],
    "afq_estimates_get_begin": [_h, c_int],
    "afq_estimates_update_publish": [_h, c_int, c_int],
    "afq_estimates_get_end": [_h, _dp],
    "afq_estimates_rdm": [_h, c_int],
    "afq_estimates_rdm_get": [_h, _dp, c_int],
    "afq_rng_seed": [_h, c_uint64, c_uint64],
    "afq_counters": [_h, c_void_p, c_int],
    "afq_counters_ext": [_h, c_void_p, c_int, c_int],
    "afq_propagator_issued_flops": [_h, POINTER(c_double), POINTER(c_double)],
    "afq_timers": [_h, _dp, c_int],
    "afq_enable_timers": [_h, c_int],
    "afq_stream": [_h, POINTER(c_void_p)],
    "afq_inverse_overlap": [_h, _dp, _dp],
    "afq_set_propagator_hirsch": [_h, _dp, c_double, c_int],
    "afq_propagate_hirsch": [_h, c_double],
    "afq_hirsch_free_projection": [_h, c_int],
    "afq_hirsch_single_site": [_h, c_int],
    "afq_propagate_hirsch_free": [_h, _dp, c_void_p, c_double],
    "afq_hirsch_kinetic": [_h],
    "afq_hirsch_two_body": [_h, _dp, _dp, _dp],
    "afq_hirsch_finish": [_h, c_double],
    "afq_bp_configure": [_h, c_int],
    "afq_bp_steps": [_h, _dp],
    "afq_bp_update": [_h, _dp, c_int, c_int, c_int, c_int, _dp],
    "afq_local_energy_full_g": [_h, _dp, c_int, _dp],
    "afq_set_trial_multi": [_h, c_int, _dp, _dp, _dp],
    "afq_walkers_det_weights": [_h, _dp],
    "afq_rng_normal": [_h, _dp, c_int64],
    "afq_rng_philox4x32": [_h, c_void_p, c_void_p, c_int],
    "afq_debug": [_h, c_int, c_int],
    "afq_last_launch": [_h, c_void_p, c_int, POINTER(c_uint64), POINTER(c_uint64)],
    "afq_comm_unique_id": [c_void_p],
    "afq_comm_available": [],
    "afq_comm_init": [_h, c_void_p, c_int, c_int],
    "afq_comm_init_ipc": [_h, c_int, c_int, ALLGATHER_FN, c_void_p],
    "afq_comm_set_transport": [_h, c_int],
    "afq_comm_probe": [_h, c_void_p],
    "afq_comm_destroy": [_h],
    "afq_comm_set_capacity": [_h, c_int],
    "afq_comm_set_timeout": [_h, c_double],
    "afq_launch_trace": [_h, c_int],
    "afq_launch_trace_get": [_h, c_void_p, c_int, c_void_p, c_void_p, c_int, POINTER(c_int)],
    "afq_comm_stats": [_h, c_void_p],
    "afq_comm_parent_ix": [_h, c_void_p],
    "afq_estimates_allreduce": [_h, _dp, c_int],
    "afq_comm_init_local": [POINTER(_h), c_int],
    "afq_popcontrol_comb_local": [POINTER(_h), c_int, c_double, c_double, c_void_p, POINTER(c_double)],
    "afq_estimates_allreduce_local": [POINTER(_h), c_int],
    "afq_set_exchange_algorithm": [_h, c_int],
    "afq_exchange_algorithm": [_h, POINTER(c_int)],
    "afq_set_msd_force_bias": [_h, c_int],
    "afq_msd_force_bias": [_h, POINTER(c_int)],
    "afq_kernel_trace": [_h, c_int],
    "afq_kernel_trace_stride": [_h, c_int, c_int],
    "afq_kernel_trace_get": [_h, c_int, _dp, c_int, POINTER(c_int)],
    "afq_kernel_issued_flops": [_h, c_int, POINTER(c_double)],
}


class AfqLibraryError(RuntimeError):
    pass


class AfqError(RuntimeError):
    def __init__(self, code, text):
        RuntimeError.__init__(self, "libafqmc_hip error %d: %s" % (code, text))
        self.code = code


_lib = None


def load(path=None):
    """Load the shared library and declare every prototype.  No GPU needed."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise AfqLibraryError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C pauxy_amd/csrc`).  There is no CPU fallback." % p)
    try:
        lib = ctypes.CDLL(p)
    except OSError as e:
        raise AfqLibraryError("cannot load %s: %s" % (p, e))
    for name, args in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise AfqLibraryError("%s does not export %s" % (p, name))
        fn.argtypes = args
        fn.restype = c_char_p if name == "afq_last_error" else c_int
    if path is None:
        _lib = lib
    return lib

# afq_kernel_trace kinds (include/afqmc_hip.h AFQ_K_*)
K_PROPAGATOR, K_EXCHANGE, K_VHS, K_FORCE_BIAS, K_GREENS = range(5)
