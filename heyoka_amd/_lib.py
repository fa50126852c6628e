"""ctypes loader for libheyoka_amd.so (the C ABI declared in include/heyoka_amd.h).

The library is the product: there is no Python/CPU fallback. If it is missing the import fails
loudly; if it loads but no MI355X is visible, every operation that needs the device raises.
"""

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libheyoka_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "heyoka_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C heyoka_amd/csrc` (there is no CPU fallback for the MI355X path)." % LIB_PATH
    )

lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)


def tree_build_id():
    """Build id of the SOURCES next to this file: first 16 hex digits of the SHA-256 over heyoka_amd/csrc/*.{cpp,hpp} (sorted)
    and include/heyoka_amd.h - what the Makefile bakes into the library (hy_build_id())."""
    import glob
    import hashlib

    csrc = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.cpp")) + glob.glob(os.path.join(csrc, "*.hpp")), key=os.path.basename)
    files.append(os.path.join(os.path.dirname(_HERE), "include", "heyoka_amd.h"))
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _check_build_id():
    """A prebuilt library which does not match the sources it ships with is a stale build: hard error (the GPU box never
    rebuilds the library - it travels with the snapshot -, so nothing else would notice).
    HEYOKA_AMD_SKIP_BUILD_ID_CHECK=1 switches the check off (installed copies without the sources skip it by themselves)."""
    if os.environ.get("HEYOKA_AMD_SKIP_BUILD_ID_CHECK") == "1" or not os.path.isdir(os.path.join(_HERE, "csrc")):
        return
    try:
        lib.hy_build_id.restype = ctypes.c_char_p
        have = lib.hy_build_id().decode()
    except AttributeError:
        have = "(no build id: a library built before round 6)"
    want = tree_build_id()
    if have != want:
        raise ImportError(
            "heyoka_amd: %s was built from other sources than the ones in this tree (library build id %s, tree %s). Rebuild it "
            "with `make -C heyoka_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`)." % (LIB_PATH, have, want))


_check_build_id()

c_void_p = ctypes.c_void_p
c_char_p = ctypes.c_char_p
c_double = ctypes.c_double
c_int = ctypes.c_int
c_size_t = ctypes.c_size_t
c_uint32 = ctypes.c_uint32
c_uint64 = ctypes.c_uint64
c_int64 = ctypes.c_int64
dptr = ctypes.POINTER(c_double)


class TabConfig(ctypes.Structure):
    _fields_ = [
        ("tol", c_double),
        ("high_accuracy", c_int),
        ("compact_mode", c_int),
        ("parallel_mode", c_int),
        ("pars", c_void_p),
        ("n_pars", c_size_t),
        ("time", c_void_p),
        ("n_time", c_size_t),
        ("device", c_int),
        ("emitter", c_int),
        ("cluster_kernel", c_int),
        ("exact_division", c_int),
        ("events_on_cluster", c_int),
        ("sum_order", c_int),
        ("batch_semantics", c_int),
    ]


STEP_CALLBACK = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p)
STEP_PRE_HOOK = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p)


class StepCallbackDesc(ctypes.Structure):
    """hy_step_callback_desc (include/heyoka_amd.h)."""

    _fields_ = [("call", STEP_CALLBACK), ("pre_hook", STEP_PRE_HOOK), ("user_data", c_void_p)]
NT_EVENT_CB = ctypes.CFUNCTYPE(None, c_void_p, c_double, c_int, c_uint32, c_void_p)
T_EVENT_CB = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, c_uint32, c_void_p)


class NtEvent(ctypes.Structure):
    """hy_nt_event (include/heyoka_amd.h)."""

    _fields_ = [("eq", c_void_p), ("cb", NT_EVENT_CB), ("user", c_void_p), ("direction", c_int)]


class TEvent(ctypes.Structure):
    """hy_t_event (include/heyoka_amd.h)."""

    _fields_ = [("eq", c_void_p), ("cb", T_EVENT_CB), ("user", c_void_p), ("direction", c_int), ("cooldown", c_double)]

ENSEMBLE_GEN = ctypes.CFUNCTYPE(c_int, c_void_p, c_size_t, c_void_p)

# (name, restype, argtypes): every symbol declared in include/heyoka_amd.h.
SIGNATURES = [
    ("hy_last_error", c_char_p, []),
    ("hy_last_error_code", c_int, []),
    ("hy_free_str", None, [c_void_p]),
    ("hy_version", c_void_p, []),
    ("hy_build_id", c_char_p, []),
    ("hy_set_logger_level", c_int, [c_int]),
    ("hy_get_logger_level", c_int, []),
    ("hy_set_log_callback", None, [c_void_p, c_void_p]),
    ("hy_device_count", c_int, []),
    ("hy_expr_var", c_void_p, [c_char_p]),
    ("hy_expr_num", c_void_p, [c_double]),
    ("hy_expr_par", c_void_p, [c_uint32]),
    ("hy_expr_time", c_void_p, []),
    ("hy_expr_neg", c_void_p, [c_void_p]),
    ("hy_expr_add", c_void_p, [c_void_p, c_void_p]),
    ("hy_expr_sub", c_void_p, [c_void_p, c_void_p]),
    ("hy_expr_mul", c_void_p, [c_void_p, c_void_p]),
    ("hy_expr_div", c_void_p, [c_void_p, c_void_p]),
    ("hy_expr_pow", c_void_p, [c_void_p, c_void_p]),
    ("hy_expr_sqrt", c_void_p, [c_void_p]),
    ("hy_expr_sin", c_void_p, [c_void_p]),
    ("hy_expr_cos", c_void_p, [c_void_p]),
    ("hy_expr_exp", c_void_p, [c_void_p]),
    ("hy_expr_log", c_void_p, [c_void_p]),
    ("hy_expr_tan", c_void_p, [c_void_p]),
    ("hy_expr_tanh", c_void_p, [c_void_p]),
    ("hy_expr_sinh", c_void_p, [c_void_p]),
    ("hy_expr_cosh", c_void_p, [c_void_p]),
    ("hy_expr_asin", c_void_p, [c_void_p]),
    ("hy_expr_acos", c_void_p, [c_void_p]),
    ("hy_expr_atan", c_void_p, [c_void_p]),
    ("hy_expr_asinh", c_void_p, [c_void_p]),
    ("hy_expr_acosh", c_void_p, [c_void_p]),
    ("hy_expr_atanh", c_void_p, [c_void_p]),
    ("hy_expr_erf", c_void_p, [c_void_p]),
    ("hy_expr_sigmoid", c_void_p, [c_void_p]),
    ("hy_expr_atan2", c_void_p, [c_void_p, c_void_p]),
    ("hy_expr_kepE", c_void_p, [c_void_p, c_void_p]),
    ("hy_expr_kepF", c_void_p, [c_void_p, c_void_p, c_void_p]),
    ("hy_expr_kepDE", c_void_p, [c_void_p, c_void_p, c_void_p]),
    ("hy_expr_custom", c_void_p, [c_char_p, c_void_p, c_size_t]),
    ("hy_expr_pi", c_void_p, []),
    ("hy_node_rule_register", c_int, [c_void_p]),
    ("hy_expr_relu", c_void_p, [c_void_p, c_double]),
    ("hy_expr_relup", c_void_p, [c_void_p, c_double]),
    ("hy_expr_select", c_void_p, [c_void_p, c_void_p, c_void_p]),
    ("hy_expr_logical", c_void_p, [c_int, c_void_p, c_size_t]),
    ("hy_expr_rel", c_void_p, [c_int, c_void_p, c_void_p]),
    ("hy_expr_sum", c_void_p, [c_void_p, c_size_t]),
    ("hy_expr_prod", c_void_p, [c_void_p, c_size_t]),
    ("hy_expr_free", None, [c_void_p]),
    ("hy_expr_str", c_void_p, [c_void_p]),
    ("hy_sys_new", c_void_p, []),
    ("hy_sys_add", c_int, [c_void_p, c_void_p, c_void_p]),
    ("hy_sys_size", c_size_t, [c_void_p]),
    ("hy_sys_free", None, [c_void_p]),
    ("hy_model_nbody", c_void_p, [c_uint32, c_void_p, c_size_t, c_double]),
    ("hy_model_pendulum", c_void_p, [c_double, c_double]),
    ("hy_sys_decomposition_str", c_void_p, [c_void_p]),
    ("hy_tab_create", c_void_p, [c_void_p, c_void_p, c_size_t, c_uint32, c_void_p]),
    ("hy_tab_create_with_events", c_void_p,
     [c_void_p, c_void_p, c_size_t, c_uint32, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]),
    ("hy_tab_with_events", c_int, [c_void_p]),
    ("hy_event_counter_nt", None, [c_void_p, ctypes.c_double, c_int, ctypes.c_uint32, c_void_p]),
    ("hy_event_counter_t", c_int, [c_void_p, c_int, ctypes.c_uint32, c_void_p]),
    ("hy_tab_set_event_timing", c_int, [c_void_p, c_int]),
    ("hy_tab_get_event_stats", c_int, [c_void_p, c_void_p]),
    ("hy_tab_reset_cooldowns", c_int, [c_void_p, ctypes.c_int64]),
    ("hy_tab_get_te_cooldowns", c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    ("hy_tab_copy", c_void_p, [c_void_p]),
    ("hy_tab_free", None, [c_void_p]),
    ("hy_tab_get_batch_size", c_uint32, [c_void_p]),
    ("hy_tab_get_order", c_uint32, [c_void_p]),
    ("hy_tab_get_dim", c_uint32, [c_void_p]),
    ("hy_tab_get_n_pars", c_uint32, [c_void_p]),
    ("hy_tab_get_n_uvars", c_uint32, [c_void_p]),
    ("hy_tab_get_tol", c_double, [c_void_p]),
    ("hy_tab_get_high_accuracy", c_int, [c_void_p]),
    ("hy_tab_get_compact_mode", c_int, [c_void_p]),
    ("hy_tab_get_event_detection_failures", ctypes.c_ulonglong, [c_void_p]),
    ("hy_tab_get_compile_seconds", c_double, [c_void_p]),
    ("hy_tab_get_hip_source", c_void_p, [c_void_p]),
    ("hy_tab_get_internal_program", c_void_p, [c_void_p]),
    ("hy_tab_get_decomposition_str", c_void_p, [c_void_p]),
    ("hy_tab_get_codegen_info", c_void_p, [c_void_p]),
    ("hy_tab_get_code_object", c_int, [c_void_p, c_void_p, c_void_p]),
    ("hy_hiprtc_compile", c_int, [c_char_p, c_void_p, c_void_p]),
    ("hy_tab_get_state", c_int, [c_void_p, c_void_p]),
    ("hy_tab_set_state", c_int, [c_void_p, c_void_p]),
    ("hy_tab_get_state_data", c_void_p, [c_void_p]),
    ("hy_tab_get_pars_data", c_void_p, [c_void_p]),
    ("hy_tab_get_pars", c_int, [c_void_p, c_void_p]),
    ("hy_tab_set_pars", c_int, [c_void_p, c_void_p]),
    ("hy_tab_get_dtime", c_int, [c_void_p, c_void_p, c_void_p]),
    ("hy_tab_set_time", c_int, [c_void_p, c_void_p, c_size_t]),
    ("hy_tab_set_dtime", c_int, [c_void_p, c_void_p, c_void_p, c_size_t]),
    ("hy_tab_get_tc", c_int, [c_void_p, c_void_p]),
    ("hy_tab_get_last_h", c_int, [c_void_p, c_void_p]),
    ("hy_tab_update_d_output", c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    ("hy_tab_step", c_int, [c_void_p, c_int]),
    ("hy_tab_step_backward", c_int, [c_void_p, c_int]),
    ("hy_tab_step_limited", c_int, [c_void_p, c_void_p, c_size_t, c_int]),
    ("hy_tab_get_step_res", c_int, [c_void_p, c_void_p, c_void_p]),
    (
        "hy_tab_propagate_until",
        c_int,
        [c_void_p, c_void_p, c_size_t, c_uint64, c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_int],
    ),
    (
        "hy_tab_propagate_for",
        c_int,
        [c_void_p, c_void_p, c_size_t, c_uint64, c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_int],
    ),
    (
        "hy_tab_propagate_until_cbs",
        c_int,
        [c_void_p, c_void_p, c_size_t, c_uint64, c_void_p, c_size_t, c_void_p, c_size_t, c_int, c_int],
    ),
    (
        "hy_tab_propagate_for_cbs",
        c_int,
        [c_void_p, c_void_p, c_size_t, c_uint64, c_void_p, c_size_t, c_void_p, c_size_t, c_int, c_int],
    ),
    (
        "hy_tab_propagate_grid_cbs",
        c_int,
        [c_void_p, c_void_p, c_size_t, c_uint64, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p],
    ),
    (
        "hy_tab_propagate_grid",
        c_int,
        [c_void_p, c_void_p, c_size_t, c_uint64, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p],
    ),
    ("hy_tab_propagate_grid_device", c_int, [c_void_p, c_void_p, c_size_t, c_int, c_uint64, c_void_p, c_size_t, c_void_p]),
    ("hy_tab_get_propagate_res", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("hy_model_nbody_ex", c_void_p, [c_uint32, c_void_p, c_size_t, c_void_p]),
    ("hy_model_nbody_energy", c_void_p, [c_uint32, c_void_p, c_size_t, c_void_p]),
    ("hy_model_nbody_potential", c_void_p, [c_uint32, c_void_p, c_size_t, c_void_p]),
    ("hy_model_pendulum_energy", c_void_p, [c_double, c_double]),
    ("hy_model_np1body", c_void_p, [c_uint32, c_void_p, c_size_t, c_void_p]),
    ("hy_model_np1body_energy", c_void_p, [c_uint32, c_void_p, c_size_t, c_void_p]),
    ("hy_model_np1body_potential", c_void_p, [c_uint32, c_void_p, c_size_t, c_void_p]),
    ("hy_model_cr3bp", c_void_p, [c_void_p]),
    ("hy_model_cr3bp_jacobi", c_void_p, [c_void_p]),
    ("hy_model_fixed_centres", c_void_p, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]),
    ("hy_model_fixed_centres_energy", c_void_p, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]),
    ("hy_model_fixed_centres_potential", c_void_p, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]),
    ("hy_model_rotating", c_void_p, [c_void_p, c_size_t]),
    ("hy_model_rotating_energy", c_void_p, [c_void_p, c_size_t]),
    ("hy_model_rotating_potential", c_void_p, [c_void_p, c_size_t]),
    ("hy_model_mascon", c_void_p, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t]),
    ("hy_model_mascon_energy", c_void_p, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t]),
    ("hy_model_mascon_potential", c_void_p, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t]),
    ("hy_sys_get_vars", c_int, [c_void_p, c_void_p]),
    ("hy_compile_aux_kernels", c_int, [c_uint32, c_uint32, c_int]),
    ("hy_cfunc_new", c_void_p, [c_void_p, c_size_t, c_void_p, c_size_t, c_int]),
    ("hy_cfunc_free", None, [c_void_p]),
    ("hy_cfunc_get_nparams", c_uint32, [c_void_p]),
    ("hy_cfunc_get_nvars", c_uint32, [c_void_p]),
    ("hy_cfunc_get_nouts", c_uint32, [c_void_p]),
    ("hy_cfunc_is_time_dependent", c_int, [c_void_p]),
    ("hy_cfunc_decomposition_str", c_void_p, [c_void_p]),
    ("hy_cfunc_get_hip_source", c_void_p, [c_void_p]),
    ("hy_cfunc_set_stream", c_int, [c_void_p, c_void_p]),
    ("hy_cfunc_eval", c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t]),
    ("hy_cfunc_eval_device", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64]),
    ("hy_tab_take_c_output", c_int, [c_void_p, c_void_p]),
    ("hy_cout_free", None, [c_void_p]),
    ("hy_cout_clone", c_void_p, [c_void_p]),
    ("hy_cout_eval", c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    ("hy_cout_eval_device", c_int, [c_void_p, c_void_p, c_void_p]),
    ("hy_cout_get_batch_size", c_uint32, [c_void_p]),
    ("hy_cout_get_dim", c_uint32, [c_void_p]),
    ("hy_cout_get_order", c_uint32, [c_void_p]),
    ("hy_cout_get_n_steps", c_int, [c_void_p, c_void_p]),
    ("hy_cout_get_bounds", c_int, [c_void_p, c_void_p, c_void_p]),
    ("hy_cout_get_times", c_int, [c_void_p, c_void_p, c_void_p]),
    ("hy_cout_get_tcs", c_int, [c_void_p, c_void_p]),
    ("hy_cout_to_string", c_void_p, [c_void_p]),
    ("hy_tab_device_ptr", c_void_p, [c_void_p, c_int]),
    ("hy_tab_mark_device_modified", c_int, [c_void_p]),
    ("hy_tab_set_stream", c_int, [c_void_p, c_void_p]),
    ("hy_tab_synchronize", c_int, [c_void_p]),
    ("hy_tab_get_last_total_steps", c_uint64, [c_void_p]),
    ("hy_tab_get_kernel_ms_history", c_size_t, [c_void_p, c_void_p, c_size_t]),
    ("hy_tab_raw_step", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64]),
    ("hy_tab_raw_step_tape", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64]),
    ("hy_tab_raw_step_e", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64]),
    ("hy_tab_raw_step_e_tape", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64]),
    ("hy_tab_raw_d_out_f", c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint64]),
    ("hy_tab_tape_size_align", c_int, [c_void_p, c_uint64, c_void_p, c_void_p]),
    (
        "hy_ensemble_propagate_until_batch",
        c_int,
        [c_void_p, c_double, c_size_t, c_void_p, c_void_p, c_uint64, c_int, c_void_p],
    ),
    (
        "hy_ensemble_propagate_for_batch",
        c_int,
        [c_void_p, c_double, c_size_t, c_void_p, c_void_p, c_uint64, c_int, c_void_p],
    ),
    ("hy_ensemble_gather_states", c_int, [c_void_p, c_size_t, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    ("hy_ensemble_gather_results", c_int, [c_void_p, c_size_t, c_int, c_void_p, c_size_t, c_int, c_void_p]),
]

for _name, _res, _args in SIGNATURES:
    _f = getattr(lib, _name)  # raises AttributeError if a declared symbol is not exported
    _f.restype = _res
    _f.argtypes = _args


def take_str(ptr):
    """Convert a malloc'd C string returned by the library into str and free it."""
    if not ptr:
        return None
    s = ctypes.cast(ptr, c_char_p).value.decode()
    lib.hy_free_str(ptr)
    return s


class NotImplementedErrorHY(NotImplementedError):
    """heyoka::not_implemented_error."""


def last_error():
    return lib.hy_last_error().decode()


def raise_for(code):
    """Map a HY_ERR_* code to the Python exception matching the reference's C++ exception."""
    if code == 0:
        return
    msg = last_error()
    if code == 1:
        raise ValueError(msg)  # std::invalid_argument
    if code == 2:
        raise OverflowError(msg)  # std::overflow_error
    if code == 3:
        raise NotImplementedErrorHY(msg)
    raise RuntimeError(msg)


def check_handle(h):
    if not h:
        raise_for(lib.hy_last_error_code() or 4)
    return h
