"""Golden fixtures (tests/golden/golden_v1.npz, made by tests/golden/make_golden.py from the CPU oracle):
the oracle must reproduce them (CPU), and the HIP path must match them without the oracle in the loop
(GPU).  They freeze the synthetic robot tables, per-knot records and converged solver outputs.
"""
import os

import numpy as np
import pytest

from aslr_to_amd import _abi, example_robot_data, scenarios

HERE = os.path.dirname(os.path.abspath(__file__))
sys_path_golden = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def golden():
    g = np.load(os.path.join(sys_path_golden, "golden_v%d.npz" % example_robot_data.TABLE_VERSION))
    assert int(g["table_version"]) == example_robot_data.TABLE_VERSION
    return g


def _cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(sys_path_golden, "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


MG = _cases()


@pytest.mark.parametrize("name,kw", MG.KNOT_CASES)
def test_oracle_reproduces_golden_knot_records(oracle, golden, name, kw):
    low = scenarios.lower(scenarios.SCENARIOS[name](**kw))
    xnext, cost, deriv = oracle.calc_diff(low, golden["knot/%s/xs" % name], golden["knot/%s/us" % name])
    np.testing.assert_allclose(xnext, golden["knot/%s/xnext" % name], rtol=0, atol=1e-13)
    np.testing.assert_allclose(cost, golden["knot/%s/cost" % name], rtol=1e-13)
    np.testing.assert_allclose(deriv, golden["knot/%s/deriv" % name], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("name,kw,solver", MG.SOLVE_CASES)
def test_oracle_reproduces_golden_solutions(oracle, golden, name, kw, solver):
    sc = scenarios.SCENARIOS[name](**kw)
    r = oracle.solve(scenarios.lower(sc), scenarios.solver_params(sc, solver=solver))
    key = "solve/%s/%s" % (name, solver)
    np.testing.assert_array_equal(r["traj_i"][_abi.TI_ITER], golden[key + "/iters"])
    np.testing.assert_allclose(r["xs"], golden[key + "/xs"], atol=1e-9)
    np.testing.assert_allclose(r["us"], golden[key + "/us"], atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", MG.KNOT_CASES)
def test_hip_calcdiff_matches_golden_records(golden, name, kw):
    import torch
    from aslr_to_amd.engine import Engine
    e = Engine(scenarios.lower(scenarios.SCENARIOS[name](**kw)))
    e.region(_abi.R_XS).copy_(torch.as_tensor(golden["knot/%s/xs" % name]))
    e.region(_abi.R_US).copy_(torch.as_tensor(golden["knot/%s/us" % name]))
    e.calc_diff()
    torch.cuda.synchronize()
    for rid, key in ((_abi.R_XNEXT, "xnext"), (_abi.R_COST, "cost"), (_abi.R_DERIV, "deriv")):
        got, ref = e.region(rid).cpu().numpy(), golden["knot/%s/%s" % (name, key)]
        assert np.max(np.abs(got - ref) / (1 + np.abs(ref))) < 1e-9, key


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,solver", MG.SOLVE_CASES)
def test_hip_solve_matches_golden_solutions(golden, name, kw, solver):
    """north_star tolerances: 1e-6 on xs / us, 1e-4 on the final cost."""
    import torch
    from aslr_to_amd.engine import Engine
    sc = scenarios.SCENARIOS[name](**kw)
    e = Engine(scenarios.lower(sc))
    e.set_candidate(None, None)
    e.solve(scenarios.solver_params(sc, solver=solver))
    torch.cuda.synchronize()
    key = "solve/%s/%s" % (name, solver)
    conv = (golden[key + "/status"] & _abi.ST_CONVERGED) != 0
    assert conv.any()
    X, U = e.region(_abi.R_XS).cpu().numpy(), e.region(_abi.R_US).cpu().numpy()
    assert np.abs(X - golden[key + "/xs"])[:, conv].max() < 1e-6
    assert np.abs(U - golden[key + "/us"])[:, conv].max() < 1e-6
    assert np.abs(e.traj_f(_abi.TF_COST).cpu().numpy() - golden[key + "/cost"])[conv].max() < 1e-4
    assert (e.traj_i(_abi.TI_ITER).cpu().numpy() == golden[key + "/iters"])[conv].all()
