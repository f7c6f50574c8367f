"""ctypes binding of libgaddpg.so (include/gaddpg.h).  There is NO CPU fallback: if the library is
missing or an entry point fails, this module raises.  torch is used only to own device memory and
to name the current stream."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GAD_LIB_PATH") or os.path.join(_HERE, "libgaddpg.so")   # env override: kernel experiments
MAX_GROUPS = 3
STAT_REPLICAS = 4

_i32, _f32, _f64, _vp = C.c_int32, C.c_float, C.c_double, C.c_void_p


class GemmFwdArgs(C.Structure):
    _fields_ = [("n_rows_dev", _vp), ("n_rows", _i32), ("row_w", _vp), ("mode", _i32),
                ("zin", _vp), ("zin_pitch", _i32), ("c_in", _i32), ("scale", _vp), ("shift", _vp),
                ("relu", _i32), ("extra", _vp), ("ones_col", _i32),
                ("src_xyz", _vp), ("ctr_xyz", _vp), ("feat", _vp), ("feat_c", _i32),
                ("action", _vp), ("act_c", _i32), ("grp_per_sample", _i32),
                ("row_pt", _vp), ("row_grp", _vp),
                ("n_groups", _i32), ("zin_off", _i32 * MAX_GROUPS), ("w_off", _i32 * MAX_GROUPS),
                ("out_off", _i32 * MAX_GROUPS), ("n_out", _i32 * MAX_GROUPS),
                ("W", _vp), ("Kp", _i32), ("zout", _vp), ("zout_pitch", _i32),
                ("stat_sum", _vp), ("stat_sq", _vp), ("stat_stride", _i32),
                ("pool_key", _vp), ("pool_row_grp", _vp), ("pool_gamma", _vp),
                ("in_stat_sum", _vp), ("in_stat_sq", _vp), ("in_stat_stride", _i32), ("in_count", _f64),
                ("in_gamma", _vp), ("in_beta", _vp), ("in_eps", _f32), ("in_momentum", _f32),
                ("in_running_mean", _vp), ("in_running_var", _vp), ("in_mean", _vp), ("in_istd", _vp),
                ("pre_W", _vp), ("pre_Kp", _i32),
                ("W_split", _vp), ("W_split_pitch", _i32), ("W_split_plane", _i32)]


class DzSrc(C.Structure):
    _fields_ = [("z", _vp), ("z_pitch", _i32), ("scale", _vp), ("shift", _vp), ("relu", _i32),
                ("coefP", _vp), ("coefQ", _vp), ("coefS", _vp), ("row_w", _vp), ("gmode", _i32),
                ("G", _vp), ("g_pitch", _i32), ("argmax", _vp), ("dout", _vp), ("row_grp", _vp),
                ("c", _i32), ("premasked", _i32),
                ("bn_dbeta", _vp), ("bn_dgamma", _vp), ("bn_stride", _i32), ("bn_count", _f64), ("bn_mean", _vp), ("bn_istd", _vp),
                ("gacc_gamma", _vp), ("gacc_beta", _vp)]


class GemmDxArgs(C.Structure):
    _fields_ = [("n_rows_dev", _vp), ("n_rows", _i32), ("dz", DzSrc), ("n_groups", _i32),
                ("dz_off", _i32 * MAX_GROUPS), ("w_off", _i32 * MAX_GROUPS), ("n_out", _i32 * MAX_GROUPS),
                ("gout_off", _i32 * MAX_GROUPS), ("accumulate", _i32), ("W", _vp), ("Kp", _i32),
                ("k_valid", _i32), ("epilogue", _i32), ("gout", _vp), ("gout_pitch", _i32),
                ("zprev", _vp), ("zprev_pitch", _i32), ("prev_scale", _vp), ("prev_shift", _vp),
                ("prev_mean", _vp), ("prev_istd", _vp), ("prev_dbeta", _vp), ("prev_dgamma", _vp),
                ("stat_stride", _i32), ("store_masked", _i32), ("dfeat", _vp), ("feat_c", _i32), ("row_pt", _vp), ("row_grp", _vp),
                ("daction", _vp), ("act_c", _i32), ("grp_per_sample", _i32),
                ("W_split_t", _vp), ("W_split_t_pitch", _i32), ("W_split_t_plane", _i32)]


class GemmDwArgs(C.Structure):
    _fields_ = [("inp", GemmFwdArgs), ("dz", DzSrc), ("dz_off", _i32 * MAX_GROUPS), ("gacc", _vp),
                ("row_splits", _i32), ("partial", _vp), ("partial_elems", C.c_int64)]


class ReplayGatherArgs(C.Structure):
    _fields_ = [("B", _i32), ("cloud_elems", _i32), ("idx", _vp), ("nxt", _vp), ("end", _vp), ("point_state", _vp),
                ("action", _vp), ("expert_action", _vp), ("goal", _vp), ("reward", _vp), ("returns", _vp),
                ("terminal", _vp), ("timestep", _vp), ("expert_flags", _vp), ("perturb_flags", _vp),
                ("out_point", _vp), ("out_next_point", _vp), ("out_action", _vp), ("out_expert_action", _vp),
                ("out_goal", _vp), ("out_reward", _vp), ("out_return", _vp), ("out_mask", _vp), ("out_time", _vp),
                ("out_time_m1", _vp), ("out_expert_flag", _vp), ("out_perturb_flag", _vp)]


class OptimJob(C.Structure):
    _fields_ = [("n", _i32), ("p", _vp), ("grad", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("active", _vp), ("m2p", _vp),
                ("packed", _vp), ("gacc", _vp), ("accumulate", _i32), ("hyper", _vp), ("clip_sumsq", _vp), ("clip_max", _f32),
                ("target", _vp), ("target_sel", _vp), ("target_m2p", _vp), ("target_packed", _vp), ("tau", _f32),
                ("hard_enable", _i32), ("absmax_p", _vp), ("absmax_grad", _vp), ("counter", _vp), ("counter_n", _i32),
                ("counter_add", _i32)]


class CopySeg(C.Structure):
    _fields_ = [("dst", _vp), ("src", _vp), ("bytes", C.c_longlong), ("add", _f32), ("reserved_", _i32)]


COPY_MAX_SEGS = 16


class SplitLayer(C.Structure):
    _fields_ = [("w_off", _i32), ("n_out", _i32), ("Kp", _i32), ("Ks", _i32), ("fwd_off", C.c_int64), ("t_off", C.c_int64)]


MAX_SPLIT_LAYERS = 16
# option "mfma_split": family mask (include/gaddpg.h GAD_SPLIT_*); 1 = every family that has the split-bf16 form
SPLIT_ALL, SPLIT_FWD_STREAM, SPLIT_FWD_WIDE, SPLIT_DX_WIDE, SPLIT_DW_WIDE, SPLIT_BWD_STREAM, SPLIT_DW_STREAM = 1, 2, 4, 8, 16, 32, 64

_lib = None


def lib():
    """Load libgaddpg.so (raises with the build hint if it is absent -- by design no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libgaddpg.so not found at %s -- build it with "
                               "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                               "The update-step path has no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.gad_last_error.restype = C.c_char_p
        L.gad_abi_version.restype = C.c_int
        L.gad_last_kernel.restype = C.c_char_p
        L.gad_plan_entry_name.restype = C.c_char_p
        _lib = L
        for name, v in OPTION_DEFAULTS.items():   # what THIS package runs with where the library's own default differs (explicit opt-in)
            check(L.gad_set_option(name.encode(), int(v)), "gad_set_option(%s)" % name)
        for k, v in os.environ.items():          # GAD_OPT_<name>=<int>: kernel-selection switches for A/B diagnostics
            if k.startswith("GAD_OPT_"):
                check(L.gad_set_option(k[8:].encode(), int(v)), "gad_set_option(%s)" % k[8:])
    return _lib


EXPORTS = (
    "gad_abi_version", "gad_last_kernel", "gad_last_error", "gad_set_option", "gad_timing_slot", "gad_stream_priority", "gad_wall_clock_khz", "gad_grid_rows_hint", "gad_bn_running_update", "gad_replay_gather", "gad_zero_buffers", "gad_furthest_point_sampling", "gad_gather_points",
    "gad_gather_points_grad", "gad_ball_query", "gad_group_points", "gad_group_points_grad",
    "gad_query_and_group", "gad_prep_points", "gad_rows_from_ball_query", "gad_rows_group_all",
    "gad_gemm_fwd", "gad_bn_finalize", "gad_bn_eval_affine", "gad_segment_pool", "gad_pool_finalize", "gad_affine_act", "gad_transpose_batched",
    "gad_pool_bwd_stats", "gad_bn_bwd_coef", "gad_gemm_dx", "gad_gemm_dw", "gad_gemm_bwd", "gad_gemm_dw_reduce", "gad_critic_loss",
    "gad_policy_outputs", "gad_policy_sample", "gad_actor_loss", "gad_actor_critic_loss", "gad_mask_counts", "gad_target_noise",
    "gad_grad_from_arena", "gad_grad_from_arena_sumsq", "gad_optim_jobs", "gad_sumsq", "gad_absmax_segments", "gad_adam_step", "gad_polyak",
    "gad_pack_params", "gad_split_weights", "gad_copy_buffers",
    "gad_plan_create", "gad_plan_destroy", "gad_plan_size", "gad_plan_add_call", "gad_plan_add_wait", "gad_plan_add_record",
    "gad_plan_add_wait_event", "gad_plan_add_memset", "gad_plan_add_memcpy", "gad_plan_patch", "gad_plan_arm_timing", "gad_plan_run",
    "gad_plan_entry_count", "gad_plan_entry_name")


class Ptr(int):
    """a raw device address (crosses the ABI as void*, never as a 32-bit int)"""


def ptr(t, offset_bytes=0):
    """device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    if isinstance(t, int):
        return Ptr(t + offset_bytes)
    return Ptr(t.data_ptr() + offset_bytes)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(status, what):
    if status != 0:
        raise RuntimeError("%s failed (status %d): %s" % (what, status, lib().gad_last_error().decode()))


def _args(*a):
    out = []
    for x in a:
        if x is None:
            out.append(C.c_void_p(None))
        elif torch.is_tensor(x):
            out.append(C.c_void_p(x.data_ptr()))
        elif isinstance(x, Ptr):
            out.append(C.c_void_p(int(x)))
        elif isinstance(x, float):
            out.append(C.c_float(x))
        elif isinstance(x, Dbl):
            out.append(C.c_double(x.v))
        elif isinstance(x, bool):
            out.append(C.c_int(int(x)))
        elif isinstance(x, int):
            out.append(C.c_int(x))
        else:
            out.append(x)
    return out


class Dbl(object):
    """marks a Python float that must cross the ABI as a C double"""
    def __init__(self, v):
        self.v = float(v)


def call(name, *a):
    """Call an entry point: tensors -> device pointers, ints -> int, floats -> float, Dbl -> double,
    None -> NULL; the current torch stream is appended as the trailing `stream` argument."""
    f = getattr(lib(), name)
    check(f(*(_args(*a) + [stream()])), name)


def call_struct(name, s):
    f = getattr(lib(), name)
    check(f(C.byref(s), stream()), name)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and (not t.is_cuda):
            raise RuntimeError("CPU tensor passed to a libgaddpg operator (CPU not supported)")
        if t is not None and not t.is_contiguous():
            raise RuntimeError("libgaddpg operators need contiguous tensors")


# options this package sets when it loads the library (the library's own default of "mfma_split" is 0 = the f32 MFMA: a plain C
# caller opts in itself); also what a test restores
OPTION_DEFAULTS = {"mfma_split": 1}


def get_option_default(name):
    """the value an option has when nothing set it: the library default, or the GAD_OPT_<name> environment override"""
    return int(os.environ.get("GAD_OPT_" + name, OPTION_DEFAULTS.get(name, 1)))


_options = {}
ROUTES = {}               # engine.Plan: tag -> kernel family of the tagged launches (valid for the current option values)


def set_option(name, value):
    """kernel-selection switch for A/B diagnostics (include/gaddpg.h: gad_set_option)"""
    check(lib().gad_set_option(name.encode(), int(value)), "gad_set_option")
    _options[name] = int(value)
    ROUTES.clear()


def get_option(name):
    """the value last set through set_option (else the default / environment override)"""
    return _options.get(name, get_option_default(name))
