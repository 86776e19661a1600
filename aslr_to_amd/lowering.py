"""Lower a (batched) shooting problem to the POD `aslr_problem_desc_t` of include/aslr_to_amd.h.

Pure host logic (numpy + ctypes): no GPU, no torch.  Used by the engine and, in tests, to feed the
same description to the CPU oracle.
"""
import ctypes as C

import numpy as np

from . import _abi


class LoweredProblem(object):
    """Owns the ctypes description and the numpy buffers its pointers refer to."""

    def __init__(self, desc, node_model, x0, frame_ref, nj, nx, nu, dam, nu_user=None):
        self.desc, self.node_model, self.x0, self.frame_ref = desc, node_model, x0, frame_ref
        self.nj, self.nx, self.nu, self.dam = nj, nx, nu, dam
        # nu: control size on the device; nu_user: the models' own nu (smaller for a pendulum actuation with one motor
        # command: the lowered controls are padded, models._DifferentialBase.lower)
        self.nu_user = nu if nu_user is None else nu_user
        self.B, self.T = desc.B, desc.T
        self.rec = _abi.record_len(nx, nu)


def lower_problem(x0s, running_models, terminal_model, frame_refs=None):
    """x0s: [B, nx] (or [nx]); running_models: list of T IntegratedActionModelEulerASR;
    frame_refs: optional [B, 12] (row-major R, p) or list of SE3 overriding every frame-placement
    reference per trajectory."""
    x0 = np.ascontiguousarray(np.atleast_2d(np.asarray(x0s, dtype=np.float64)))
    B = x0.shape[0]
    T = len(running_models)
    if T < 1:
        raise ValueError("a shooting problem needs at least one running model")
    state = terminal_model.state
    chain_model = state.pinocchio
    nj, nx = chain_model.nv, state.ndx
    if x0.shape[1] != nx:
        raise ValueError("x0 must have %d entries" % nx)
    # distinct models by identity, in order of first appearance
    table, index = [], {}
    node_model = np.zeros(T + 1, dtype=np.int32)
    for t, m in enumerate(list(running_models) + [terminal_model]):
        if m.state.pinocchio is not chain_model:
            raise ValueError("all action models of a problem must share one robot model")
        key = id(m)
        if key not in index:
            if len(table) >= _abi.MAX_MODELS:
                raise ValueError("at most %d distinct action models per problem" % _abi.MAX_MODELS)
            index[key] = len(table)
            table.append(m)
        node_model[t] = index[key]
    nu_user = table[0].nu
    nu = table[0].differential.nu_dev
    dam = table[0].differential.dam
    desc = _abi.ProblemDesc()
    desc.B, desc.T, desc.nmodels = B, T, len(table)
    desc.chain = chain_model.to_struct()
    for i, m in enumerate(table):
        if m.nu != nu_user or m.differential.dam != dam:
            raise ValueError("all action models of a problem must share nu and the actuation kind")
        desc.models[i] = m.lower()
    fr = None
    if frame_refs is not None:
        if len(frame_refs) != B:
            raise ValueError("frame_refs needs one entry per trajectory")
        if hasattr(frame_refs[0], "as12"):
            fr = np.stack([f.as12() for f in frame_refs])
        else:
            fr = np.asarray(frame_refs, dtype=np.float64)
        fr = np.ascontiguousarray(fr.reshape(B, 12))
        desc.frame_ref = fr.ctypes.data_as(C.POINTER(C.c_double))
    desc.node_model = node_model.ctypes.data_as(C.POINTER(C.c_int32))
    desc.x0 = x0.ctypes.data_as(C.POINTER(C.c_double))
    return LoweredProblem(desc, node_model, x0, fr, nj, nx, nu, dam, nu_user)


def shard_rows(B, rank, world_size):
    """Contiguous block of the batch owned by `rank`: rows [lo, hi)."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, rem = divmod(B, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi
