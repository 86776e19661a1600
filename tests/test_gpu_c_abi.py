"""The C ABI used from plain C: tests/c_abi/solve_from_c.c (HIP runtime + include/aslr_to_amd.h, no Python and no torch in
its process) is built with gcc against libaslr_to_hip.so and libamdhip64, given a problem description as a binary blob, and must return
the bits the Python layer gets for the same problem."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from aslr_to_amd import _abi, scenarios

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_plain_c_caller_gets_the_same_bits(tmp_path):
    import torch
    from aslr_to_amd.engine import Engine
    gcc, rocm = shutil.which("gcc"), os.environ.get("ROCM_PATH", "/opt/rocm")
    if gcc is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("no C compiler / ROCm headers on this box")
    lib_dir = os.path.join(ROOT, "aslr_to_amd", "csrc")
    exe = str(tmp_path / "solve_from_c")
    # plain C, gcc: the HIP runtime API header and libamdhip64 are all it needs next to include/aslr_to_amd.h
    subprocess.check_call([gcc, "-std=c11", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"), "-I",
                           os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "solve_from_c.c"),
                           "-L", lib_dir, "-laslr_to_hip", "-L", os.path.join(rocm, "lib"), "-lamdhip64",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", exe])
    sc = scenarios.two_dof_vsa_boxddp(B=37, T=30, seed=2)
    low = scenarios.lower(sc)
    sp = scenarios.solver_params(sc, maxiter=50)
    blob = str(tmp_path / "problem.bin")
    with open(blob, "wb") as f:
        f.write(bytes(memoryview(low.desc)))
        f.write(bytes(memoryview(sp)))
        f.write(low.node_model.astype(np.int32).tobytes())
        f.write(np.ascontiguousarray(low.x0).tobytes())
        f.write(np.int32(1).tobytes())
        f.write(np.ascontiguousarray(low.frame_ref).tobytes())
    out = str(tmp_path / "out.bin")
    r = subprocess.run([exe, blob, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    B, T, nx, nu = low.B, low.T, low.nx, low.nu
    raw = open(out, "rb").read()
    o = 0

    def take(n, dt):
        nonlocal o
        a = np.frombuffer(raw, dtype=dt, count=n, offset=o)
        o += a.nbytes
        return a
    xs = take((T + 1) * B * nx, np.float64).reshape(T + 1, B, nx)
    us = take(T * B * nu, np.float64).reshape(T, B, nu)
    cost, iters, status = take(B, np.float64), take(B, np.int32), take(B, np.int32)
    batch_iters = int(take(1, np.int32)[0])
    e = Engine(low)
    e.set_candidate(None, None)
    it = e.solve(sp, poll_every=4)
    torch.cuda.synchronize()
    assert batch_iters == it
    np.testing.assert_array_equal(xs, e.region(_abi.R_XS).cpu().numpy())
    np.testing.assert_array_equal(us, e.region(_abi.R_US).cpu().numpy())
    np.testing.assert_array_equal(cost, e.traj_f(_abi.TF_COST).cpu().numpy())
    np.testing.assert_array_equal(iters, e.traj_i(_abi.TI_ITER).cpu().numpy())
    np.testing.assert_array_equal(status, e.traj_i(_abi.TI_STATUS).cpu().numpy())
    assert C.sizeof(_abi.ProblemDesc) + C.sizeof(_abi.SolverParams) < os.path.getsize(blob)
