"""ctypes mirror of include/aslr_to_amd.h and the loader of the HIP shared library.

The product path has NO CPU fallback: if ``csrc/libaslr_to_hip.so`` is missing or cannot be
loaded, importing the compute path raises.  (The CPU restatement under ``oracle/`` is test
infrastructure and is never imported from here.)
"""
import ctypes as C
import os

MAX_NJ = 7
MAX_NX = 28
MAX_NU = 14
MAX_COSTS = 6
MAX_MODELS = 4
NALPHA = 10
ABI_VERSION = 2

OK, E_INVALID, E_HIP, E_NODEVICE, E_WORKSPACE = 0, -1, -2, -3, -4
DAM_SEA, DAM_VSA = 0, 1
COST_FRAME_PLACEMENT, COST_STATE, COST_CONTROL, COST_PENDULUM, COST_STIFFNESS = range(5)
SOLVER_DDP, SOLVER_FDDP, SOLVER_BOXDDP = 0, 1, 2
ST_CONVERGED, ST_REG_MAX, ST_BACKWARD_ERR, ST_FORWARD_ERR = 1, 2, 4, 8

(R_XS, R_US, R_XNEXT, R_COST, R_DERIV, R_GAPS, R_KGAIN, R_KFF, R_QU, R_VX, R_VXX, R_XS_TRY,
 R_US_TRY, R_TRAJ_F, R_TRAJ_I, R_X0, R_FRAME_REF, R_VXXF, R_DESC, R_NODE_MODEL, R_COST_TRY, R_DYN, R_POOL_SAVE,
 R_COUNT) = range(24)

(TF_COST, TF_STOP, TF_XREG, TF_D1, TF_D2, TF_STEP, TF_DV, TF_DVEXP, TF_DG, TF_DQ,
 TF_COST_TRY0) = range(11)
TF_DVTRY0 = TF_COST_TRY0 + NALPHA
TF_COUNT = TF_DVTRY0 + NALPHA
(TI_ITER, TI_STATUS, TI_FEASIBLE, TI_WAS_FEASIBLE, TI_RECALC, TI_ACCEPTED, TI_DONE, TI_NTRIALS,
 TI_GAPFLAG, TI_TRYFAIL0) = range(10)
TI_COUNT = TI_TRYFAIL0 + NALPHA

# rows of the per-iteration log (aslr_set_iteration_log)
(LOG_COST, LOG_STOP, LOG_XREG, LOG_STEP, LOG_D1, LOG_D2, LOG_DV, LOG_DVEXP, LOG_ACCEPTED, LOG_STATUS, LOG_FEASIBLE,
 LOG_COUNT) = range(12)

_d = C.c_double
_i = C.c_int32


class Chain(C.Structure):
    _fields_ = [("nj", _i), ("_pad0", _i), ("gravity", _d * 3),
                ("joint_R", (_d * 9) * MAX_NJ), ("joint_p", (_d * 3) * MAX_NJ),
                ("axis", (_d * 3) * MAX_NJ), ("mass", _d * MAX_NJ), ("com", (_d * 3) * MAX_NJ),
                ("inertia", (_d * 9) * MAX_NJ)]


class Cost(C.Structure):
    _fields_ = [("type", _i), ("frame_joint", _i), ("weight", _d), ("act_w", _d * MAX_NX),
                ("ref", _d * MAX_NX), ("frame_R", _d * 9), ("frame_p", _d * 3), ("lambda_", _d)]


class Model(C.Structure):
    _fields_ = [("dam", _i), ("nu", _i), ("ncosts", _i), ("has_u_limits", _i), ("dt", _d),
                ("K", _d * (MAX_NJ * MAX_NJ)), ("B", _d * (MAX_NJ * MAX_NJ)),
                ("S", _d * (MAX_NJ * MAX_NJ)), ("u_lb", _d * MAX_NU), ("u_ub", _d * MAX_NU),
                ("costs", Cost * MAX_COSTS)]


class ProblemDesc(C.Structure):
    _fields_ = [("B", _i), ("T", _i), ("nmodels", _i), ("_pad0", _i), ("chain", Chain),
                ("models", Model * MAX_MODELS), ("node_model", C.POINTER(_i)),
                ("x0", C.POINTER(_d)), ("frame_ref", C.POINTER(_d))]


class SolverParams(C.Structure):
    _fields_ = [("solver", _i), ("maxiter", _i), ("is_feasible", _i), ("fixed_iterations", _i),
                ("reg_init", _d), ("th_stop", _d), ("th_grad", _d), ("th_gaptol", _d),
                ("th_stepdec", _d), ("th_stepinc", _d), ("th_acceptstep", _d),
                ("th_acceptnegstep", _d), ("reg_min", _d), ("reg_max", _d),
                ("reg_incfactor", _d), ("reg_decfactor", _d), ("boxqp_maxiter", _i),
                ("_pad0", _i), ("boxqp_th_acceptstep", _d), ("boxqp_th_grad", _d),
                ("boxqp_reg", _d)]


class Pool(C.Structure):
    """aslr_pool_t: device pointers of a pool solve (aslr_solve_pool)."""
    _fields_ = [("P", _i), ("_pad0", _i), ("x0", C.c_void_p), ("frame_ref", C.c_void_p), ("xs_out", C.c_void_p),
                ("us_out", C.c_void_p), ("stat_f", C.c_void_p), ("stat_i", C.c_void_p), ("slot_problem", C.c_void_p),
                ("counters", C.c_void_p), ("xs_init", C.c_void_p), ("us_init", C.c_void_p)]


class Region(C.Structure):
    _fields_ = [("offset", C.c_int64), ("bytes", C.c_int64)]


def default_solver_params(solver=SOLVER_DDP):
    """Defaults of crocoddyl.SolverDDP / BoxQP; pure host logic, mirrors
    aslr_solver_params_default() of the library (tests check they agree)."""
    sp = SolverParams()
    sp.solver = solver
    sp.maxiter = 100
    sp.is_feasible = 0
    sp.fixed_iterations = 0
    sp.reg_init = float("nan")
    sp.th_stop = 1e-9
    sp.th_grad = 1e-12
    sp.th_gaptol = 1e-16
    sp.th_stepdec = 0.5
    sp.th_stepinc = 0.01
    sp.th_acceptstep = 0.1
    sp.th_acceptnegstep = 2.0
    sp.reg_min = 1e-9
    sp.reg_max = 1e9
    sp.reg_incfactor = 10.0
    sp.reg_decfactor = 10.0
    sp.boxqp_maxiter = 100
    sp.boxqp_th_acceptstep = 0.1
    sp.boxqp_th_grad = 1e-5
    sp.boxqp_reg = 0.0
    return sp


def record_len(nx, nu):
    n = 2 * nx * nx + 2 * nx * nu + nu * nu + nx + nu
    return (n + 15) // 16 * 16


def record_offsets(nx, nu):
    """Offsets (in doubles) of the blocks of one DERIV record."""
    o = {}
    o["Fx"] = 0
    o["Fu"] = o["Fx"] + nx * nx
    o["Lxx"] = o["Fu"] + nx * nu
    o["Lxu"] = o["Lxx"] + nx * nx
    o["Luu"] = o["Lxu"] + nx * nu
    o["Lx"] = o["Luu"] + nu * nu
    o["Lu"] = o["Lx"] + nx
    return o


LIB_NAME = "libaslr_to_hip.so"
_lib = None


def lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", LIB_NAME)


# every symbol include/aslr_to_amd.h declares
EXPORTED_SYMBOLS = [
    "aslr_abi_version", "aslr_sizeof", "aslr_record_len", "aslr_solver_params_default",
    "aslr_workspace_bytes", "aslr_problem_create", "aslr_problem_destroy", "aslr_problem_region",
    "aslr_calc", "aslr_calc_diff", "aslr_backward_pass", "aslr_forward_pass", "aslr_solve",
    "aslr_iterate", "aslr_iterate_timed", "aslr_finalize", "aslr_count_active", "aslr_dam_eval", "aslr_quasi_static", "aslr_last_error",
    "aslr_dam_residuals", "aslr_residual_len", "aslr_frame_placement", "aslr_set_iteration_log",
    "aslr_iterate_n", "aslr_set_subshards", "aslr_solve_pool",
]


def load_library():
    """Load the HIP C-ABI library; raise loudly when it is absent (no fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(
            "aslr_to_amd: the HIP extension %s is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback." % path)
    # torch first: its bundled HIP runtime must be the one (and only) libamdhip64 in the process, or
    # device pointers of torch tensors would belong to a different runtime than our kernel launches
    import torch  # noqa: F401
    lib = C.CDLL(path)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.aslr_abi_version.restype = C.c_int
    lib.aslr_sizeof.restype = i64
    lib.aslr_sizeof.argtypes = [C.c_int]
    lib.aslr_record_len.restype = i32
    lib.aslr_record_len.argtypes = [i32, i32]
    lib.aslr_solver_params_default.restype = None
    lib.aslr_solver_params_default.argtypes = [C.POINTER(SolverParams), i32]
    lib.aslr_workspace_bytes.restype = i64
    lib.aslr_workspace_bytes.argtypes = [C.POINTER(ProblemDesc)]
    lib.aslr_problem_create.restype = C.c_int
    lib.aslr_problem_create.argtypes = [C.POINTER(ProblemDesc), vp, i64, vp, C.POINTER(vp)]
    lib.aslr_problem_destroy.restype = C.c_int
    lib.aslr_problem_destroy.argtypes = [vp]
    lib.aslr_problem_region.restype = C.c_int
    lib.aslr_problem_region.argtypes = [vp, i32, C.POINTER(Region)]
    for name in ("aslr_calc", "aslr_calc_diff", "aslr_finalize"):
        f = getattr(lib, name)
        f.restype = C.c_int
        f.argtypes = [vp, vp]
    for name in ("aslr_backward_pass", "aslr_forward_pass"):
        f = getattr(lib, name)
        f.restype = C.c_int
        f.argtypes = [vp, C.POINTER(SolverParams), vp]
    lib.aslr_solve.restype = C.c_int
    lib.aslr_solve.argtypes = [vp, C.POINTER(SolverParams), i32, vp, C.POINTER(i32)]
    lib.aslr_iterate.restype = C.c_int
    lib.aslr_iterate.argtypes = [vp, C.POINTER(SolverParams), i32, vp]
    lib.aslr_iterate_timed.restype = C.c_int
    lib.aslr_iterate_timed.argtypes = [vp, C.POINTER(SolverParams), i32, vp, C.POINTER(C.c_float)]
    lib.aslr_count_active.restype = C.c_int
    lib.aslr_count_active.argtypes = [vp, vp, C.POINTER(i32)]
    lib.aslr_dam_eval.restype = C.c_int
    lib.aslr_dam_eval.argtypes = [vp, i32, i32] + [vp] * 11 + [vp]
    lib.aslr_quasi_static.restype = C.c_int
    lib.aslr_quasi_static.argtypes = [vp, i32, C.c_double, vp, vp]
    lib.aslr_last_error.restype = C.c_char_p
    lib.aslr_dam_residuals.restype = C.c_int
    lib.aslr_dam_residuals.argtypes = [vp, i32, i32, vp, vp, vp, vp]
    lib.aslr_residual_len.restype = i32
    lib.aslr_residual_len.argtypes = [C.POINTER(Model), i32]
    lib.aslr_frame_placement.restype = C.c_int
    lib.aslr_frame_placement.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_double), i32, vp, i64, vp, vp]
    lib.aslr_iterate_n.restype = C.c_int
    lib.aslr_iterate_n.argtypes = [vp, C.POINTER(SolverParams), i32, i32, vp]
    lib.aslr_solve_pool.restype = C.c_int
    lib.aslr_solve_pool.argtypes = [vp, C.POINTER(SolverParams), C.POINTER(Pool), i32, i32, vp, C.POINTER(i32)]
    lib.aslr_set_subshards.restype = C.c_int
    lib.aslr_set_subshards.argtypes = [vp, i32]
    lib.aslr_set_iteration_log.restype = C.c_int
    lib.aslr_set_iteration_log.argtypes = [vp, vp, i32]
    if lib.aslr_abi_version() != ABI_VERSION:
        raise ImportError("aslr_to_amd: ABI version mismatch between %s and the Python layer" % path)
    for which, st in enumerate((Chain, Cost, Model, ProblemDesc, SolverParams, Region, Pool)):
        if lib.aslr_sizeof(which) != C.sizeof(st):
            raise ImportError("aslr_to_amd: struct %s size mismatch (C %d, Python %d)"
                              % (st.__name__, lib.aslr_sizeof(which), C.sizeof(st)))
    _lib = lib
    return lib


class AslrError(RuntimeError):
    pass


def check(rc, what):
    if rc != OK:
        msg = {E_INVALID: "invalid argument", E_HIP: "HIP error", E_NODEVICE: "no GPU device",
               E_WORKSPACE: "workspace too small or misaligned"}.get(rc, "error %d" % rc)
        detail = ""
        if _lib is not None:
            e = _lib.aslr_last_error()
            if e:
                detail = ": " + e.decode()
        raise AslrError("%s failed: %s%s" % (what, msg, detail))
