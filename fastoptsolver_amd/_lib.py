"""ctypes binding of libfos_hip.so (include/fos.h).  Loading never builds and never falls back:
if the shared library is missing the import of any solver raises, loudly."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfos_hip.so")

FOS_F32, FOS_BF16 = 0, 1
MODE_FISTA, MODE_DELTA, MODE_ISTA = 0, 1, 2
PROX_L1, PROX_ENET = 0, 1
STOP_NONE, STOP_STEP, STOP_RATIO, STOP_GRAD, STOP_LS_STALL = 0, 1, 2, 3, 4
PLAN_NO_RESIDENT, PLAN_NO_TALL, PLAN_NO_WIDE, PLAN_NO_COLBLOCK, PLAN_CLUSTER, PLAN_INTERLEAVE, PLAN_NO_INTERLEAVE, PLAN_NO_CLUSTER, PLAN_FUSED_MFMA, PLAN_CHIP_RESIDENT, PLAN_NO_CHIP_RESIDENT = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024


class FistaParams(C.Structure):
    _fields_ = [("tau", C.c_double), ("alpha1", C.c_double), ("alpha2", C.c_double), ("delta", C.c_double),
                ("restart_threshold", C.c_double), ("tol_step", C.c_double), ("tol_ratio", C.c_double),
                ("tol_grad", C.c_double), ("mode", C.c_int32), ("prox_kind", C.c_int32), ("adaptive_restart", C.c_int32),
                ("reserved", C.c_int32)]


class LineSearchState(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("f0", "d0", "dtest", "width", "width1", "lo", "flo", "dlo", "hi", "fhi", "dhi",
                                           "smin", "smax")] + [(k, C.c_int32) for k in ("stage1", "brackt", "status", "reserved")]


class LbfgsResult(C.Structure):
    _fields_ = [("f", C.c_double), ("gmax", C.c_double), ("nit", C.c_int32), ("nfev", C.c_int32), ("task", C.c_int32),
                ("reserved", C.c_int32)]


LS_FG, LS_CONVERGENCE, LS_WARNING, LS_ERROR = 0, 1, 2, 3


class FistaStatus(C.Structure):
    _fields_ = [("t_prev", C.c_double), ("beta", C.c_double), ("this_step", C.c_double), ("prev_step", C.c_double),
                ("ratio", C.c_double), ("rr", C.c_double), ("gnorm2", C.c_double), ("xnorm1", C.c_double),
                ("xnorm2", C.c_double), ("rr_x", C.c_double), ("tau", C.c_double), ("k", C.c_int64), ("stopped", C.c_int32),
                ("restarts", C.c_int32)]


_vp, _i64, _i32, _f32, _f64 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_double

# name -> (restype, argtypes).  Every symbol include/fos.h declares must be listed here
# (tests/test_abi.py checks the header against this table and against the .so).
SIGNATURES = {
    "fos_last_error": (C.c_char_p, []),
    "fos_abi_version": (_i32, []),
    "fos_problem_create": (_i32, [C.POINTER(_vp), _vp, _i64, _i64, _i64, _i32, _vp, _vp]),
    "fos_problem_destroy": (_i32, [_vp]),
    "fos_problem_plan": (_i32, [_vp, C.POINTER(C.c_int32)]),
    "fos_problem_tune": (_i32, [_vp, _i32, _i32, _i32, _i32]),
    "fos_problem_tune_dd": (_i32, [_vp, _i32]),
    "fos_problem_set_gbuf": (_i32, [_vp, _vp]),
    "fos_problem_set_stream": (_i32, [_vp, _vp]),
    "fos_problem_replan": (_i32, [_vp, C.c_uint]),
    "fos_comm_unique_id": (_i32, [C.c_char_p]),
    "fos_comm_create": (_i32, [C.POINTER(_vp), C.c_char_p, _i32, _i32]),
    "fos_comm_destroy": (_i32, [_vp]),
    "fos_comm_mesh_create": (_i32, [C.POINTER(_vp), _i32, _i32, _i64, C.c_char_p]),
    "fos_comm_mesh_connect": (_i32, [_vp, C.c_char_p]),
    "fos_comm_check": (_i32, [_vp, _vp]),
    "fos_comm_info": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "fos_comm_mesh_info": (_i32, [_vp, C.POINTER(_i32), C.POINTER(C.c_int64)]),
    "fos_comm_transport": (C.c_char_p, []),
    "fos_comm_allreduce": (_i32, [_vp, _vp, _i64, _i32, _vp]),
    "fos_problem_set_comm": (_i32, [_vp, _vp]),
    "fos_problem_set_comm_cols": (_i32, [_vp, _vp]),
    "fos_problem_profile": (_i32, [_vp, _i32]),
    "fos_problem_profile_read": (_i32, [_vp, C.POINTER(_f64), C.POINTER(_i64)]),
    "fos_fista_run_chip": (_i32, [_vp, _i32]),
    "fos_problem_set_fused_stamps": (_i32, [_vp, _vp]),
    "fos_stream_read_probe": (_i32, [_vp, C.c_size_t, _i32, _vp, C.POINTER(_f64), C.POINTER(_f64)]),
    "fos_gemv_pair": (_i32, [_vp, _vp, _f32, _vp, _vp]),
    "fos_gemv_pair_f64": (_i32, [_vp, _vp, _f64, _vp, _vp]),
    "fos_gemv_pair_dd": (_i32, [_vp, _vp, _f64, _vp]),
    "fos_residual_objective": (_i32, [_vp, _vp, _vp]),
    "fos_residual_batch": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "fos_power_iter": (_i32, [_vp, _vp, _i32, _f64, C.POINTER(_f64), C.POINTER(_i32)]),
    "fos_prox_l1": (_i32, [_vp, _f32, _vp, _i64, _vp]),
    "fos_prox_l1_vec": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "fos_prox_elastic_net": (_i32, [_vp, _f32, _f32, _f32, _vp, _i64, _vp]),
    "fos_prox_elastic_net_vec": (_i32, [_vp, _vp, _f32, _f32, _vp, _i64, _vp]),
    "fos_fista_create": (_i32, [_vp, C.POINTER(_vp)]),
    "fos_fista_destroy": (_i32, [_vp]),
    "fos_fista_reset": (_i32, [_vp, C.POINTER(FistaParams), _vp]),
    "fos_fista_set_tau": (_i32, [_vp, _f64]),
    "fos_fista_set_precise": (_i32, [_vp, _i32]),
    "fos_fista_set_gbuf64": (_i32, [_vp, _vp]),
    "fos_fista_run": (_i32, [_vp, _i32]),
    "fos_fista_history_workspace": (_i64, [_vp, _i32]),
    "fos_fista_run_history": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "fos_fista_run_resident": (_i32, [_vp, _i32, _i32, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fos_fista_run_multi": (_i32, [C.POINTER(_vp), _i32, _i32]),
    "fos_fista_run_fused": (_i32, [_vp, _i32]),
    "fos_fista_grad": (_i32, [_vp]),
    "fos_fista_grad_dual": (_i32, [_vp]),
    "fos_fista_update": (_i32, [_vp]),
    "fos_fista_trial": (_i32, [_vp, _f64, _i32, C.POINTER(_f64)]),   # out8
    "fos_fista_trial_batch": (_i32, [_vp, _f64, _f64, _i32, C.POINTER(_f64)]),
    "fos_fista_run_backtracking": (_i32, [_vp, _i32, _f64, _f64, _f64, _vp, _vp]),
    "fos_fista_run_recorded": (_i32, [_vp, _i32, _i32, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _vp]),
    "fos_fista_resume_after_stall": (_i32, [_vp, C.POINTER(_f64)]),
    "fos_fista_status_get": (_i32, [_vp, C.POINTER(FistaStatus)]),
    "fos_fista_get_x": (_i32, [_vp, _vp]),
    "fos_fista_x": (_vp, [_vp]),
    "fos_fista_gbuf": (_vp, [_vp]),
    "fos_lbfgs_two_loop": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp]),
    "fos_vec_stats": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "fos_vec_axpby": (_i32, [_f64, _vp, _f64, _vp, _vp, _i64, _vp]),
    "fos_vec_stats_f64": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "fos_vec_axpby_f64": (_i32, [_f64, _vp, _f64, _vp, _vp, _i64, _vp]),
    "fos_lbfgs_two_loop_dd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp]),
    "fos_lbfgs_direction_work": (_i64, [_i64]),
    "fos_lbfgs_direction_dd": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp, _vp, _i64, _vp]),
    "fos_lbfgs_direction_cols": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i64]),
    "fos_vec_stats_dd": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "fos_vec_axpby_dd": (_i32, [_f64, _vp, _f64, _vp, _vp, _i64, _vp]),
    "fos_linesearch_begin": (_f64, [C.POINTER(LineSearchState), _f64, _f64, _f64]),
    "fos_linesearch_step": (_f64, [C.POINTER(LineSearchState), _f64, _f64, _f64]),
    "fos_lbfgs_minimize": (_i32, [_vp, _f64, _i32, _f64, _vp, C.POINTER(_f64), _vp, C.POINTER(C.c_float), _i32,
                                  C.POINTER(LbfgsResult)]),
}

_lib = None


class FosError(RuntimeError):
    pass


def load():
    """Return the loaded CDLL (cached).  Raises FosError if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FosError(
            f"{LIB_PATH} is missing: the HIP extension is the only compute path of fastoptsolver_amd.\n"
            "Build it with:  python -m fastoptsolver_amd.build   (needs hipcc; cross-compiles for gfx950)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().fos_last_error().decode("utf-8", "replace")
        raise FosError(f"{what} failed with code {rc}: {msg}")
