"""ctypes binding of libdust_amd.so (include/dust_amd.h).

There is no CPU fallback: if the HIP library has not been built, importing this module raises; if it is built but no
MI355X is visible, `dust_create` fails with DUST_ERR_NO_DEVICE and `check()` raises `DustError`.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DUST_AMD_LIB", os.path.join(_HERE, "libdust_amd.so"))  # override: diagnostic builds only

ABI_VERSION = 2
OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_HIP, ERR_STATE = range(6)
MODEL_PENDULUM, MODEL_PARTICLE, MODEL_SKID_STEER = 0, 1, 2
COST_PENDULUM_QUADCOS, COST_PARTICLE_DEFAULT, COST_QUADRATIC = 0, 1, 2
KERNEL_K1_RBF, KERNEL_K2_IIDMP, KERNEL_K2_SHARED, KERNEL_IMQ = 0, 1, 2, 3
LIK_EXP_UTILITY, LIK_EXPECTED_COST = 0, 1
OPT_SGD, OPT_ADAM = 0, 1
ROLL_REPEAT, ROLL_MEAN, ROLL_RESAMPLE = 0, 1, 2
STEP_ARGMAX, STEP_AVERAGE, STEP_EXTERNAL = 0, 1, 2
PARAM_PYFLOAT, PARAM_SAMPLED, PARAM_TENSOR0D = 0, 1, 2
CONTROL_ACCELERATION, CONTROL_VELOCITY = 0, 1
PTR_DEVICE, STORE_STATES, EPS_AROUND_A_MAT, EPS_F16, STORE_F16 = 1, 2, 4, 8, 16
K_ROLLOUT, K_PRIOR_SCORE, K_STEIN, K_UPDATE, K_FORWARD, K_BANDWIDTH, K_MPF, K_COUNT = 0, 1, 2, 3, 4, 5, 6, 8


class DustError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("libdust_amd status %d: %s" % (status, msg))
        self.status = status


class Param(C.Structure):
    _fields_ = [("kind", C.c_int32), ("column", C.c_int32), ("value", C.c_double)]


class SkidConfig(C.Structure):  # dust_skid_config
    _fields_ = [("x_icr", Param), ("wheel_radius", Param), ("axial_distance", Param), ("min_wheel_speed", C.c_float * 2),
                ("max_wheel_speed", C.c_float * 2), ("goal", C.c_float * 5), ("w_state", C.c_float * 5), ("w_term", C.c_float * 5),
                ("w_ctrl", C.c_float * 2)]


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32),
        ("n_policies", C.c_int32), ("n_samples", C.c_int32), ("n_params", C.c_int32), ("horizon", C.c_int32),
        ("dim_a", C.c_int32), ("dim_s", C.c_int32), ("dim_p", C.c_int32),
        ("shard_offset", C.c_int32), ("shard_size", C.c_int32),
        ("model", C.c_int32), ("cost", C.c_int32), ("kernel", C.c_int32), ("likelihood", C.c_int32),
        ("optimizer", C.c_int32), ("roll_strategy", C.c_int32),
        ("weighted_prior", C.c_int32), ("params_log_space", C.c_int32), ("params_interleave", C.c_int32),
        ("alpha", C.c_float), ("temperature", C.c_float), ("a_reg", C.c_float),
        ("lr", C.c_float), ("adam_beta1", C.c_float), ("adam_beta2", C.c_float), ("adam_eps", C.c_float),
        ("chol_a", C.c_float * 4), ("sigma_a", C.c_float * 4), ("a_pre", C.c_float * 4), ("sigma_p", C.c_float * 4),
        ("bw_scale", C.c_float), ("imq_ell", C.c_float),
        ("min_a", C.c_float * 4), ("max_a", C.c_float * 4),
        ("seed", C.c_uint64),
        ("dt", C.c_double),
        ("g", Param), ("mass", Param), ("length", Param),
        ("max_torque", C.c_double), ("max_speed_pend", C.c_double), ("w_cos", C.c_double), ("w_vel", C.c_double),
        ("max_speed", C.c_float), ("max_accel", C.c_float),
        ("can_crash", C.c_int32), ("with_obstacle", C.c_int32),
        ("cell_size", C.c_double),
        ("target", C.c_float * 4), ("w_state", C.c_float * 4), ("w_term", C.c_float * 4), ("w_ctrl", C.c_float * 2),
        ("w_obs", C.c_float),
        ("control_type", C.c_int32), ("ctrl_noise", C.c_int32), ("dyn_std", C.c_float * 2),
        ("full_cov", C.c_int32), ("chol_a_off", C.c_float), ("a_pre_off", C.c_float), ("chol_p", C.c_float * 3),
    ]


class MpfConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("n_particles", C.c_int32),
        ("dim_p", C.c_int32), ("dim_s", C.c_int32), ("dim_a", C.c_int32), ("model", C.c_int32), ("log_space", C.c_int32),
        ("obs_std", C.c_float), ("lr", C.c_float), ("bw_scale", C.c_float), ("init_bw", C.c_float),
        ("model_cfg", Config),
    ]


FP = C.POINTER(C.c_float)
VP = C.c_void_p

# every symbol include/dust_amd.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "dust_last_error": (C.c_char_p, []),
    "dust_abi_version": (C.c_int, []),
    "dust_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "dust_create": (C.c_int, [C.POINTER(Config), C.POINTER(VP)]),
    "dust_clone": (C.c_int, [VP, C.POINTER(VP)]),
    "dust_destroy": (None, [VP]),
    "dust_sync": (C.c_int, [VP]),
    "dust_tick_stats": (C.c_int, [VP, C.POINTER(C.c_longlong)]),
    "dust_get_config": (C.c_int, [VP, C.POINTER(Config)]),
    "dust_set_model_param": (C.c_int, [VP, C.c_char_p, C.c_double, C.c_int]),
    "dust_set_param_weights": (C.c_int, [VP, FP]),
    "dust_set_grid": (C.c_int, [VP, FP, C.c_int, C.c_int, C.c_float, C.c_float]),
    "dust_set_ctrl_noise": (C.c_int, [VP, FP, C.c_int]),
    "dust_mpf_set_ctrl_noise": (C.c_int, [VP, FP, C.c_int]),
    "dust_set_theta": (C.c_int, [VP, FP]),
    "dust_get_theta": (C.c_int, [VP, FP]),
    "dust_set_prior": (C.c_int, [VP, FP, FP]),
    "dust_get_prior": (C.c_int, [VP, FP, FP]),
    "dust_set_a_mat": (C.c_int, [VP, FP]),
    "dust_get_a_mat": (C.c_int, [VP, FP]),
    "dust_get_a_mix": (C.c_int, [VP, FP]),
    "dust_set_a_seq": (C.c_int, [VP, FP]),
    "dust_get_a_seq": (C.c_int, [VP, FP]),
    "dust_disco_forward": (C.c_int, [VP, FP, VP, FP, C.c_int, FP, FP, FP, FP]),
    "dust_disco_step": (C.c_int, [VP, C.c_int, C.c_int, FP, FP]),
    "dust_likelihood_sample": (C.c_int, [VP, FP, VP, FP, C.c_int, FP, FP]),
    "dust_likelihood_log_prob": (C.c_int, [VP, FP]),
    "dust_svmpc_phi": (C.c_int, [VP, FP, FP, FP, FP, FP]),
    "dust_svmpc_step": (C.c_int, [VP, FP, VP, FP, C.c_int]),
    "dust_svmpc_optimize": (C.c_int, [VP, FP, C.c_int, VP, FP, C.c_int]),
    "dust_svmpc_forward": (C.c_int, [VP, FP, FP]),
    "dust_svmpc_get_weights": (C.c_int, [VP, FP]),
    "dust_svmpc_roll": (C.c_int, [VP, C.c_int, C.c_int, FP]),
    "dust_svmpc_update_prior": (C.c_int, [VP, FP]),
    "dust_svmpc_forward_ex": (C.c_int, [VP, C.c_int, FP, FP, FP]),
    "dust_likelihood_sample_at": (C.c_int, [VP, FP, FP, VP, FP, C.c_int, FP, FP]),
    "dust_svmpc_tick": (C.c_int, [VP, FP, C.c_int, VP, FP, C.c_int, FP, FP]),
    "dust_svmpc_serve_start": (C.c_int, [VP, C.c_int, C.c_double]),
    "dust_svmpc_serve_stop": (C.c_int, [VP]),
    "dust_get_costs": (C.c_int, [VP, FP]),
    "dust_get_actions": (C.c_int, [VP, FP]),
    "dust_get_states_rows": (C.c_int, [VP, C.POINTER(C.c_longlong), C.c_int, VP]),
    "dust_get_score": (C.c_int, [VP, FP]),
    "dust_get_phi": (C.c_int, [VP, FP]),
    "dust_get_score_parts": (C.c_int, [VP, FP, FP]),
    "dust_get_log_weights": (C.c_int, [VP, FP, FP]),
    "dust_get_bandwidths": (C.c_int, [VP, FP]),
    "dust_gather_buffers": (C.c_int, [VP, C.POINTER(VP), C.POINTER(VP), C.POINTER(C.c_size_t)]),
    "dust_svmpc_local_score": (C.c_int, [VP, FP, VP, FP, C.c_int]),
    "dust_svmpc_local_rollout": (C.c_int, [VP, FP, VP, FP, C.c_int]),
    "dust_svmpc_local_prior_score": (C.c_int, [VP]),
    "dust_svmpc_apply_phi": (C.c_int, [VP]),
    "dust_svmpc_forward_local": (C.c_int, [VP, C.POINTER(VP), C.POINTER(C.c_size_t)]),
    "dust_svmpc_forward_finish": (C.c_int, [VP, FP, FP]),
    "dust_comm_unique_id": (C.c_int, [VP]),
    "dust_comm_validate": (C.c_int, [VP, C.c_int, C.c_int]),
    "dust_comm_init": (C.c_int, [VP, VP, C.c_int, C.c_int]),
    "dust_comm_destroy": (C.c_int, [VP]),
    "dust_comm_probe": (C.c_int, [VP, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "dust_comm_peer_gather": (C.c_int, [VP, C.c_int]),
    "dust_set_stream": (C.c_int, [VP, VP]),
    "dust_profile_enable": (C.c_int, [VP, C.c_int]),
    "dust_profile_get": (C.c_int, [VP, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "dust_profile_reset": (C.c_int, [VP]),
    "dust_profile_rollout": (C.c_int, [VP, FP, VP, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "dust_kernel_name": (C.c_char_p, [C.c_int]),
    "dust_rollout_algorithmic_bytes": (C.c_int, [VP, C.c_int, C.POINTER(C.c_double)]),
    "dust_device_noise_alloc": (C.c_int, [VP, C.c_size_t, C.c_uint64, C.c_int, C.POINTER(VP)]),
    "dust_device_free": (C.c_int, [VP, VP]),
    "dust_mpf_create": (C.c_int, [C.POINTER(MpfConfig), FP, FP, C.POINTER(VP)]),
    "dust_mpf_clone": (C.c_int, [VP, C.POINTER(VP)]),
    "dust_set_k2_bandwidth": (C.c_int, [VP, C.c_float, C.c_float]),
    "dust_mpf_set_optimizer": (C.c_int, [VP, C.c_int, C.c_float, C.c_float, C.c_float]),
    "dust_mpf_destroy": (None, [VP]),
    "dust_mpf_optimize": (C.c_int, [VP, FP, FP, C.c_float, C.c_int, FP]),
    "dust_mpf_phi": (C.c_int, [VP, C.c_float, FP]),
    "dust_mpf_condition": (C.c_int, [VP, FP, FP]),
    "dust_mpf_set_grid": (C.c_int, [VP, FP, C.c_int, C.c_int, C.c_float, C.c_float]),
    "dust_mpf_get_particles": (C.c_int, [VP, FP]),
    "dust_mpf_set_particles": (C.c_int, [VP, FP]),
    "dust_mpf_get_prior": (C.c_int, [VP, FP, FP]),
    "dust_set_skid_steer": (C.c_int, [VP, C.POINTER(SkidConfig)]),
    "dust_mpf_set_prior_bw": (C.c_int, [VP, FP, C.c_int]),
    "dust_mpf_get_prior_bw": (C.c_int, [VP, FP]),
    "dust_mpf_stats": (C.c_int, [VP, C.POINTER(C.c_longlong)]),
    "dust_mpf_prior_sample": (C.c_int, [VP, C.c_int, C.c_uint64, FP]),
    "dust_mpf_prior_log_prob": (C.c_int, [VP, C.c_int, FP, FP]),
    "dust_mpf_silverman": (C.c_int, [VP, FP]),
    "dust_dual_tick": (C.c_int, [VP, VP, FP, FP, C.c_int, C.c_int, C.c_float, C.c_uint64, FP, FP, FP]),
}

_lib = None


def _share_torch_hip_runtime():
    """One HIP runtime per process.  torch wheels bundle their own libamdhip64.so.7 (same SONAME as /opt/rocm's): whichever copy is
    loaded first serves both, and torch on top of a runtime it was not built against reports "No HIP GPUs are available" (measured:
    Context first, torch.cuda afterwards).  libdust_amd.so only uses the stable HIP API, so when torch is installed but not yet
    imported its copy is loaded first - without importing torch.  DUST_AMD_SYSTEM_HIP=1 skips this (a process that never uses
    torch.cuda)."""
    if "torch" in sys.modules or os.environ.get("DUST_AMD_SYSTEM_HIP"):
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
            if os.path.exists(p):
                C.CDLL(p, mode=C.RTLD_GLOBAL)
    except Exception:  # no torch here: the system runtime is the only one
        pass


def load():
    """Load the HIP library; raises if it was never built (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is None:
        _share_torch_hip_runtime()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: the HIP extension is not built and dust_amd has NO CPU fallback. "
                "Build it with __graft_entry__.build() (hipcc --offload-arch=gfx950)." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the library is stale
            fn.restype = res
            fn.argtypes = args
        if lib.dust_abi_version() != ABI_VERSION:
            raise ImportError("libdust_amd.so ABI %d != binding ABI %d: rebuild" % (lib.dust_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def check(status):
    if status != OK:
        raise DustError(status, load().dust_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    st = load().dust_device_count(C.byref(n))
    return n.value if st == OK else 0
