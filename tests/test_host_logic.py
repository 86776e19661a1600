"""Host logic and C-ABI checks that need no GPU: struct mirrors, exported symbols, lowering of the
reference-shaped Python objects to the POD description, sharding, and the loud failure of the product
path without a device.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import aslr_to_amd as aslr_to
from aslr_to_amd import _abi, crocoddyl, example_robot_data, pinocchio, scenarios
from aslr_to_amd.lowering import lower_problem, shard_rows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _abi.load_library()
    hdr = open(os.path.join(ROOT, "include", "aslr_to_amd.h")).read()
    import re
    declared = set(re.findall(r"\b(aslr_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_abi.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    nm = subprocess.check_output(["nm", "-D", "--defined-only", _abi.lib_path()]).decode()
    for name in declared:
        assert (" T " + name + "\n") in nm, name


def test_struct_mirrors_match_the_compiled_sizes_and_defaults():
    lib = _abi.load_library()
    for which, st in enumerate((_abi.Chain, _abi.Cost, _abi.Model, _abi.ProblemDesc, _abi.SolverParams, _abi.Region)):
        assert lib.aslr_sizeof(which) == C.sizeof(st)
    assert lib.aslr_sizeof(99) == -1
    for nx, nu in ((8, 2), (8, 4), (28, 7)):
        assert lib.aslr_record_len(nx, nu) == _abi.record_len(nx, nu)
    assert _abi.record_len(8, 4) == 224 and _abi.record_len(8, 2) == 176 and _abi.record_len(28, 7) == 2048
    for solver in (_abi.SOLVER_DDP, _abi.SOLVER_FDDP, _abi.SOLVER_BOXDDP):
        a, b = _abi.SolverParams(), _abi.default_solver_params(solver)
        lib.aslr_solver_params_default(C.byref(a), solver)
        for name, _ in _abi.SolverParams._fields_:
            va, vb = getattr(a, name), getattr(b, name)
            assert (va == vb) or (np.isnan(va) and np.isnan(vb)), name
    # crocoddyl.SolverDDP defaults
    d = _abi.default_solver_params()
    assert (d.th_stop, d.th_grad, d.th_stepdec, d.th_stepinc, d.th_acceptstep) == (1e-9, 1e-12, 0.5, 0.01, 0.1)
    assert (d.reg_min, d.reg_max, d.reg_incfactor, d.reg_decfactor) == (1e-9, 1e9, 10.0, 10.0)


def test_workspace_size_is_the_sum_of_the_documented_regions():
    lib = _abi.load_library()
    sc = scenarios.two_dof_vsa_boxddp(B=70, T=9)
    low = scenarios.lower(sc)
    n = lib.aslr_workspace_bytes(C.byref(low.desc))
    B, T, nx, nu, rec = 70, 9, 8, 4, 224
    doubles = ((T + 1) * B * nx * 5 + T * B * nu * 3 + (T + 1) * B + (T + 1) * B * rec + T * B * nu * nx
               + (T + 1) * B * nx * nx + 10 * (T + 1) * B * nx + 10 * T * B * nu + _abi.TF_COUNT * B + B * nx + B * 12)
    doubles += 10 * (T + 1) * B  # COST_TRY
    doubles += B * (nx + 12)     # POOL_SAVE
    assert doubles * 8 <= n <= doubles * 8 + 64 * 1024
    bad = scenarios.lower(sc)
    bad.desc.B = 0
    assert lib.aslr_workspace_bytes(C.byref(bad.desc)) < 0


def test_lowering_of_the_vsa_boxddp_example():
    sc = scenarios.two_dof_vsa_boxddp(B=5, T=12)
    low = scenarios.lower(sc)
    d = low.desc
    assert (d.B, d.T, d.nmodels, low.nx, low.nu, low.dam) == (5, 12, 2, 8, 4, _abi.DAM_VSA)
    assert list(low.node_model) == [0] * 12 + [1]
    run, term = d.models[0], d.models[1]
    assert run.dt == 1e-2 and term.dt == 0.0 and run.nu == 4
    assert run.has_u_limits == 1 and term.has_u_limits == 0
    assert list(run.u_lb)[:4] == [-100, -100, 0, 0] and list(run.u_ub)[:4] == [100, 100, 100, 100]
    assert [run.costs[i].type for i in range(run.ncosts)] == [_abi.COST_FRAME_PLACEMENT, _abi.COST_STATE, _abi.COST_CONTROL]
    assert [run.costs[i].weight for i in range(run.ncosts)] == [1.0, 1e-1, 1e-1]
    assert term.ncosts == 1 and term.costs[0].weight == 4e4
    fc = run.costs[0]
    assert fc.frame_joint == 1 and list(fc.ref)[9:12] == [.01, .2, .18] and list(fc.act_w)[:6] == [1.0] * 6
    assert abs(run.B[0] - 1e-3) < 1e-18 and run.B[1] == 0.0
    assert list(d.chain.gravity) == [9.81, 0.0, 0.0] and d.chain.nj == 2
    # batch inputs: trajectory 0 is the script's nominal problem
    assert not sc["x0"][0].any() and list(sc["frame_refs"][0][9:]) == [.01, .2, .18]
    assert (sc["x0"][:, :2] == sc["x0"][:, 2:4]).all() and not sc["x0"][:, 4:].any()


def test_lowering_rejects_what_the_kernels_do_not_support():
    model = example_robot_data.load("asr_twodof").model
    state = aslr_to.StateMultibodyASR(model)
    act = aslr_to.ASRActuation(state)
    costs = crocoddyl.CostModelSum(state, act.nu)
    w_bad = crocoddyl.ActivationModelWeightedQuad(np.ones(3))
    costs.addCost("x", crocoddyl.CostModelResidual(state, w_bad, crocoddyl.ResidualModelState(state, state.zero(), act.nu)), 1.0)
    iam = aslr_to.IntegratedActionModelEulerASR(aslr_to.DifferentialFreeASRFwdDynamicsModel(state, act, costs), 1e-2)
    with pytest.raises(ValueError):
        lower_problem(np.zeros(8), [iam], iam)
    with pytest.raises(ValueError):
        lower_problem(np.zeros(7), [], iam)
    other = aslr_to.StateMultibodyASR(example_robot_data.load("asr_twodof").model)
    costs2 = crocoddyl.CostModelSum(other, 2)
    iam2 = aslr_to.IntegratedActionModelEulerASR(
        aslr_to.DifferentialFreeASRFwdDynamicsModel(other, aslr_to.ASRActuation(other), costs2), 1e-2)
    good = crocoddyl.CostModelSum(state, 2)
    iam_ok = aslr_to.IntegratedActionModelEulerASR(aslr_to.DifferentialFreeASRFwdDynamicsModel(state, act, good), 1e-2)
    with pytest.raises(ValueError):
        lower_problem(np.zeros(8), [iam_ok], iam2)  # two different robot models in one problem


def test_state_and_actuation_classes_follow_the_reference():
    model = example_robot_data.load("talos_arm").model
    state = aslr_to.StateMultibodyASR(model)
    assert (state.nx, state.ndx, state.nq, state.nv) == (28, 28, 14, 14)  # statemultibody_aslr.py:15
    x0, x1 = state.rand(), state.rand()
    np.testing.assert_allclose(state.integrate(x0, state.diff(x0, x1)), x1)
    J1, J2 = state.Jdiff(x0, x1)
    assert np.array_equal(J1, -np.eye(28)) and np.array_equal(J2, np.eye(28))
    assert aslr_to.ASRActuation(state).nu == 7 and aslr_to.VSAASRActuation(state).nu == 7
    vsa = aslr_to.DifferentialFreeFwdDynamicsModelVSA(state, aslr_to.VSAASRActuation(state), crocoddyl.CostModelSum(state, 14))
    assert vsa.nu == 14 and list(vsa._default_u()) == [0.0] * 7 + [3.0] * 7  # free_fwddyn_vsa.py:8,21-23
    sea = aslr_to.DifferentialFreeASRFwdDynamicsModel(state, aslr_to.ASRActuation(state), crocoddyl.CostModelSum(state, 7))
    assert np.allclose(sea.K, 0.1 * np.eye(7)) and np.allclose(sea.B, 1e-3 * np.eye(7))  # free_fwddyn_asr.py:12-19
    pend = aslr_to.ActuationModelDoublePendulum(aslr_to.StateMultibodyASR(example_robot_data.load("double_pendulum").model), 0, 2)
    assert np.array_equal(pend.motor_matrix(), [[1, 0], [0, 0]])  # __init__.py:284-289
    M = pinocchio.SE3(np.eye(3), [1, 2, 3]) * pinocchio.SE3(np.eye(3), [1, 2, 3]).inverse()
    assert np.allclose(M.translation, 0)


def test_u_squared_matches_the_reference_helper():
    class Log(object):
        us = [np.array([1.0, 2.0]), np.array([3.0, -1.0])]
    np.testing.assert_allclose(aslr_to.u_squared(Log()), [10.0, 5.0])


@pytest.mark.parametrize("B,world", [(4096, 1), (32768, 8), (10, 3), (7, 8)])
def test_shard_rows_partition_the_batch_contiguously(B, world):
    rows = [shard_rows(B, r, world) for r in range(world)]
    assert rows[0][0] == 0 and rows[-1][1] == B
    assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
    sizes = [hi - lo for lo, hi in rows]
    assert max(sizes) - min(sizes) <= 1
    if (B, world) == (32768, 8):
        assert rows[3] == (4096 * 3, 4096 * 4)  # shard r gets rows [4096 r, 4096 (r+1))


def test_sharded_problem_lowers_to_slices_of_the_full_batch():
    sc = scenarios.two_dof_vsa_boxddp(B=10, T=4)
    full = scenarios.lower(sc)
    for r in range(3):
        p = crocoddyl.ShootingProblem(sc["x0"], sc["running"], sc["terminal"], frame_refs=sc["frame_refs"], rank=r, world_size=3)
        lo, hi = p.rows
        np.testing.assert_array_equal(p.lowered.x0, full.x0[lo:hi])
        np.testing.assert_array_equal(p.lowered.frame_ref, full.frame_ref[lo:hi])
        assert p.batch == hi - lo and p.T == 4


def test_product_path_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from aslr_to_amd.engine import Engine
    sc = scenarios.two_dof_sea(B=2, T=3)
    with pytest.raises(_abi.AslrError):
        Engine(scenarios.lower(sc))
    problem = crocoddyl.ShootingProblem(sc["x0"], sc["running"], sc["terminal"], frame_refs=sc["frame_refs"])
    with pytest.raises(_abi.AslrError):
        crocoddyl.SolverDDP(problem).solve([], [], 5)
    data = sc["running"][0].createData()
    with pytest.raises(_abi.AslrError):
        sc["running"][0].calc(data, np.zeros(8), np.zeros(2))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "aslr_to_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in src and "aslr_cpu_" not in src and "aslr_oracle" not in src, f


def test_cost_sum_orders_its_residuals_by_name_like_crocoddyl():
    """data.r of a CostModelSum stacks the residual vectors in the order of the cost NAMES (Crocoddyl keeps its cost
    items in a std::map), not in the order addCost was called -- which is the order the kernels evaluate them in."""
    from aslr_to_amd import scenarios
    dam = scenarios.two_dof_vsa_modified(B=1, T=1)["running"][0].differential
    # insertion order: gripperPose (6), xReg (8), uReg (4), vsa (2)
    raw = np.arange(20, dtype=float)
    got = dam.costs.order_residuals(raw, 8, 4)
    np.testing.assert_array_equal(got, np.concatenate([raw[0:6], raw[14:18], raw[18:20], raw[6:14]]))
    assert dam.costs.nr == 20
    with pytest.raises(ValueError):
        dam.costs.order_residuals(raw[:-1], 8, 4)


def test_parity_report_names_the_first_differing_decision():
    """tests/_parity.py on hand-made logs: identical logs give no exception; a different accepted step length, an earlier
    exit and a silent cost drift are each named at the right iteration."""
    import _parity
    from aslr_to_amd import _abi as A
    n, B = 6, 4
    lg = np.full((n, A.LOG_COUNT, B), np.nan)
    for k in range(5):
        lg[k, :, :] = 0.0
        lg[k, A.LOG_COST] = 100.0 - k
        lg[k, A.LOG_XREG] = 1e-9
    lr = lg.copy()
    assert all(_parity.first_decision_flip(lg, lr, b) is None for b in range(B))
    lr[3, A.LOG_ACCEPTED, 1] = 2.0                      # trajectory 1: another step length at iteration 3
    lr[4, :, 2] = np.nan                                # trajectory 2: the oracle stopped after iteration 3
    lr[2:, A.LOG_COST, 3] += 1e-5                       # trajectory 3: costs drift apart from iteration 2 on
    f1, f2 = _parity.first_decision_flip(lg, lr, 1), _parity.first_decision_flip(lg, lr, 2)
    assert f1["kind"] == "line search" and f1["iteration"] == 3
    assert f2["kind"] == "exit" and f2["iteration"] == 4
    assert _parity.first_decision_flip(lg, lr, 3) is None and _parity.first_cost_drift(lg, lr, 3)[0] == 2
    assert _parity.first_cost_drift(lg, lr, 0) is None
    # status words: decision bits exactly, the overflow note within its allowance
    _parity.assert_status_words_match(np.array([1, 9, 1, 2]), np.array([1, 1, 1, 2]))
    with pytest.raises(AssertionError):
        _parity.assert_status_words_match(np.array([1, 1, 1, 2]), np.array([1, 1, 3, 2]))
    with pytest.raises(AssertionError):
        _parity.assert_status_words_match(np.array([9, 9, 1, 2]), np.array([1, 1, 1, 2]))


def test_one_motor_command_pendulum_is_lowered_with_padded_controls():
    """ActuationModelDoublePendulum(state, actLink=0, nu=1) (python/aslr_to/__init__.py:279-281): the model keeps nu = 1
    for its user; the lowered description carries nu = nj = 2 with a zero second column of S, a unit weight on the padded
    command in the control cost, and the same first column / weight as the nu = 2 lowering of the same problem."""
    lo1, lo2 = scenarios.lower(scenarios.double_pendulum_nu1(T=5)), scenarios.lower(scenarios.double_pendulum(T=5))
    assert (lo1.nu_user, lo1.nu, lo2.nu_user, lo2.nu) == (1, 2, 2, 2)
    m1, m2 = lo1.desc.models[0], lo2.desc.models[0]
    assert m1.nu == 2 and list(m1.S[:4]) == [1.0, 0.0, 0.0, 0.0] and list(m2.S[:4]) == [1.0, 0.0, 0.0, 0.0]
    c1 = [c for c in m1.costs[:m1.ncosts] if c.type == _abi.COST_CONTROL][0]
    c2 = [c for c in m2.costs[:m2.ncosts] if c.type == _abi.COST_CONTROL][0]
    assert list(c1.act_w[:2]) == [1.0, 1.0] and list(c2.act_w[:2]) == [1.0, 0.0] and c1.weight == c2.weight
    sc = scenarios.double_pendulum_nu1(T=5)
    assert sc["running"][0].nu == 1 and sc["running"][0].differential.nu_dev == 2
