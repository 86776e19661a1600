"""ctypes binding of the CPU oracle (oracle/libaslr_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by aslr_to_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from aslr_to_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libaslr_oracle.so")
_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = C.CDLL(_LIB)
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _arr(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
    if n is not None:
        assert a.size == n, (a.size, n)
    return a


# ---- rigid-body pieces ----
def rnea(chain, q, v, a):
    nj = chain.nj
    q, v, a = _arr(q, nj), _arr(v, nj), _arr(a, nj)
    tau = np.zeros(nj)
    lib().aslr_cpu_rnea(C.byref(chain), _d(q), _d(v), _d(a), _d(tau))
    return tau


def crba(chain, q):
    nj = chain.nj
    q = _arr(q, nj)
    M = np.zeros((nj, nj))
    lib().aslr_cpu_crba(C.byref(chain), _d(q), _d(M))
    return M


def nle(chain, q, v):
    nj = chain.nj
    q, v = _arr(q, nj), _arr(v, nj)
    out = np.zeros(nj)
    lib().aslr_cpu_nle(C.byref(chain), _d(q), _d(v), _d(out))
    return out


def rnea_derivatives(chain, q, v, a):
    nj = chain.nj
    q, v, a = _arr(q, nj), _arr(v, nj), _arr(a, nj)
    dq, dv = np.zeros((nj, nj)), np.zeros((nj, nj))
    lib().aslr_cpu_rnea_derivatives(C.byref(chain), _d(q), _d(v), _d(a), _d(dq), _d(dv))
    return dq, dv


def frame_placement(chain, q, joint, fR, fp):
    q, fR, fp = _arr(q, chain.nj), _arr(fR, 9), _arr(fp, 3)
    R, p = np.zeros((3, 3)), np.zeros(3)
    lib().aslr_cpu_frame_placement(C.byref(chain), _d(q), int(joint), _d(fR), _d(fp), _d(R), _d(p))
    return R, p


def frame_jacobian(chain, q, joint, fR, fp):
    q, fR, fp = _arr(q, chain.nj), _arr(fR, 9), _arr(fp, 3)
    J = np.zeros((6, chain.nj))
    lib().aslr_cpu_frame_jacobian(C.byref(chain), _d(q), int(joint), _d(fR), _d(fp), _d(J))
    return J


def log6(R, p):
    R, p = _arr(R, 9), _arr(p, 3)
    r = np.zeros(6)
    lib().aslr_cpu_log6(_d(R), _d(p), _d(r))
    return r


def exp6(r):
    r = _arr(r, 6)
    R, p = np.zeros((3, 3)), np.zeros(3)
    lib().aslr_cpu_exp6(_d(r), _d(R), _d(p))
    return R, p


def jlog6(R, p):
    R, p = _arr(R, 9), _arr(p, 3)
    J = np.zeros((6, 6))
    lib().aslr_cpu_jlog6(_d(R), _d(p), _d(J))
    return J


# ---- one knot ----
def dam(low, mi, x, u, frame_ref=None):
    """DAM calc + calcDiff -> dict(xout, cost, Fx, Fu, Lx, Lu, Lxx, Lxu, Luu)."""
    nx, nu, nv = low.nx, low.nu, low.nx // 2
    x = _arr(x, nx)
    up = None if u is None else _arr(u, nu)
    o = dict(xout=np.zeros(nv), cost=np.zeros(1), Fx=np.zeros((nv, nx)), Fu=np.zeros((nv, nu)), Lx=np.zeros(nx),
             Lu=np.zeros(nu), Lxx=np.zeros((nx, nx)), Lxu=np.zeros((nx, nu)), Luu=np.zeros((nu, nu)))
    fr = None if frame_ref is None else _arr(frame_ref, 12)
    lib().aslr_cpu_dam(C.byref(low.desc.chain), C.byref(low.desc.models[mi]), None if fr is None else _d(fr),
                       _d(x), None if up is None else _d(up), _d(o["xout"]), _d(o["cost"]), _d(o["Fx"]),
                       _d(o["Fu"]), _d(o["Lx"]), _d(o["Lu"]), _d(o["Lxx"]), _d(o["Lxu"]), _d(o["Luu"]))
    o["cost"] = float(o["cost"][0])
    return o


def knot(low, mi, x, u, frame_ref=None, diff=True):
    nx, nu = low.nx, low.nu
    x = _arr(x, nx)
    up = None if u is None else _arr(u, nu)
    xnext, cost, rec = np.zeros(nx), np.zeros(1), np.zeros(low.rec)
    fr = None if frame_ref is None else _arr(frame_ref, 12)
    lib().aslr_cpu_knot(C.byref(low.desc.chain), C.byref(low.desc.models[mi]), None if fr is None else _d(fr),
                        _d(x), None if up is None else _d(up), _d(xnext), _d(cost), _d(rec) if diff else None)
    out = dict(xnext=xnext, cost=float(cost[0]))
    if diff:
        o = _abi.record_offsets(nx, nu)
        shp = {"Fx": (nx, nx), "Fu": (nx, nu), "Lxx": (nx, nx), "Lxu": (nx, nu), "Luu": (nu, nu), "Lx": (nx,), "Lu": (nu,)}
        for k, s in shp.items():
            out[k] = rec[o[k]:o[k] + int(np.prod(s))].reshape(s).copy()
        out["rec"] = rec
    return out


# ---- batched, layouts of include/aslr_to_amd.h (time-major) ----
def calc_diff(low, xs, us, diff=True):
    """xs [T+1, B, nx], us [T, B, nu] -> xnext [T+1,B,nx], cost [T+1,B], deriv [T+1,B,rec]"""
    T, B, nx, nu = low.T, low.B, low.nx, low.nu
    xs, us = _arr(xs, (T + 1) * B * nx), _arr(us, T * B * nu)
    xnext, cost = np.zeros((T + 1, B, nx)), np.zeros((T + 1, B))
    deriv = np.zeros((T + 1, B, low.rec)) if diff else None
    rc = lib().aslr_cpu_calc_diff(C.byref(low.desc), _d(xs), _d(us), _d(xnext), _d(cost), _d(deriv) if diff else None)
    assert rc == 0
    return xnext, cost, deriv


def backward_pass(low, sp, deriv, gaps, us, xreg, feasible, kff0=None):
    T, B, nx, nu = low.T, low.B, low.nx, low.nu
    deriv, gaps, us = _arr(deriv), _arr(gaps, (T + 1) * B * nx), _arr(us, T * B * nu)
    xreg = _arr(np.broadcast_to(xreg, (B,)).copy(), B)
    feas = np.ascontiguousarray(np.broadcast_to(feasible, (B,)).astype(np.int32))
    K, k = np.zeros((T, B, nu, nx)), np.zeros((T, B, nu)) if kff0 is None else _arr(kff0).reshape(T, B, nu).copy()
    qu, vx, vxx = np.zeros((T, B, nu)), np.zeros((T + 1, B, nx)), np.zeros((T + 1, B, nx, nx))
    d1, d2, stop, fail = np.zeros(B), np.zeros(B), np.zeros(B), np.zeros(B, dtype=np.int32)
    rc = lib().aslr_cpu_backward_pass(C.byref(low.desc), C.byref(sp), _d(deriv), _d(gaps), _d(us), _d(xreg), _i(feas),
                                      _d(K), _d(k), _d(qu), _d(vx), _d(vxx), _d(d1), _d(d2), _d(stop), _i(fail))
    assert rc == 0
    return dict(K=K, k=k, Qu=qu, Vx=vx, Vxx=vxx, d1=d1, d2=d2, stop=stop, fail=fail)


def forward_pass(low, sp, alpha, xs, us, K, k, gaps=None, feasible=None):
    T, B, nx, nu = low.T, low.B, low.nx, low.nu
    xs, us, K, k = _arr(xs, (T + 1) * B * nx), _arr(us, T * B * nu), _arr(K, T * B * nu * nx), _arr(k, T * B * nu)
    g = None if gaps is None else _arr(gaps, (T + 1) * B * nx)
    f = None if feasible is None else np.ascontiguousarray(np.broadcast_to(feasible, (B,)).astype(np.int32))
    xs_try, us_try, cost_try = np.zeros((T + 1, B, nx)), np.zeros((T, B, nu)), np.zeros(B)
    fail = np.zeros(B, dtype=np.int32)
    rc = lib().aslr_cpu_forward_pass(C.byref(low.desc), C.byref(sp), C.c_double(alpha), _d(xs), _d(us), _d(K), _d(k),
                                     None if g is None else _d(g), None if f is None else _i(f), _d(xs_try),
                                     _d(us_try), _d(cost_try), _i(fail))
    assert rc == 0
    return xs_try, us_try, cost_try, fail


def dam_residuals(low, mi, x, u, frame_ref=None):
    """data.r of one point: the stacked cost residuals in the order of the model's cost list."""
    nr = sum({_abi.COST_FRAME_PLACEMENT: 6, _abi.COST_STATE: low.nx, _abi.COST_CONTROL: low.nu, _abi.COST_PENDULUM: 6,
              _abi.COST_STIFFNESS: low.nu // 2}[low.desc.models[mi].costs[c].type] for c in range(low.desc.models[mi].ncosts))
    x, u = _arr(x, low.nx), _arr(u, low.nu)
    r = np.zeros(nr)
    fr = None if frame_ref is None else _arr(frame_ref, 12)
    lib().aslr_cpu_dam_residuals(C.byref(low.desc.chain), C.byref(low.desc.models[mi]), None if fr is None else _d(fr),
                                 _d(x), _d(u), _d(r))
    return r


def solve(low, sp, xs=None, us=None, nthreads=1, log_cap=0):
    """-> dict(xs [T+1,B,nx], us [T,B,nu], traj_f [TF_COUNT,B], traj_i [TI_COUNT,B]); with log_cap > 0 also
    log [log_cap, LOG_COUNT, B], the per-iteration solver state (NaN where a trajectory had stopped)."""
    T, B, nx, nu = low.T, low.B, low.nx, low.nu
    xs = np.zeros((T + 1, B, nx)) if xs is None else _arr(xs, (T + 1) * B * nx).reshape(T + 1, B, nx).copy()
    us = np.zeros((T, B, nu)) if us is None else _arr(us, T * B * nu).reshape(T, B, nu).copy()
    tf = np.zeros((_abi.TF_COUNT, B))
    ti = np.zeros((_abi.TI_COUNT, B), dtype=np.int32)
    log = np.full((log_cap, _abi.LOG_COUNT, B), np.nan) if log_cap > 0 else None
    rc = lib().aslr_cpu_solve_log(C.byref(low.desc), C.byref(sp), _d(xs), _d(us), _d(tf), _i(ti), int(nthreads),
                                  None if log is None else _d(log), int(log_cap))
    assert rc == 0
    out = dict(xs=xs, us=us, traj_f=tf, traj_i=ti)
    if log is not None:
        out["log"] = log
    return out


def boxqp(H, q, lb, ub, xinit, maxiter=100, th_acceptstep=0.1, th_grad=1e-9, reg=1e-9):
    n = len(q)
    H, q, lb, ub = _arr(H, n * n), _arr(q, n), _arr(lb, n), _arr(ub, n)
    x = _arr(xinit, n).copy()
    Hff_inv = np.zeros(n * n)
    fidx, cidx = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
    nf, nc = C.c_int32(0), C.c_int32(0)
    it = lib().aslr_cpu_boxqp(n, _d(H), _d(q), _d(lb), _d(ub), _d(x), int(maxiter), C.c_double(th_acceptstep),
                              C.c_double(th_grad), C.c_double(reg), _d(Hff_inv), _i(fidx), C.byref(nf), _i(cidx),
                              C.byref(nc))
    nfv = nf.value
    return dict(x=x, iters=it, free=fidx[:nfv].copy(), clamped=cidx[:nc.value].copy(),
                Hff_inv=Hff_inv[:nfv * nfv].reshape(nfv, nfv).copy())


def quasi_static(low, mi, x, maxiter=100, tol=1e-9, frame_ref=None):
    x = _arr(x, low.nx)
    u = np.zeros(low.nu)
    fr = None if frame_ref is None else _arr(frame_ref, 12)
    it = lib().aslr_cpu_quasi_static(C.byref(low.desc), int(mi), None if fr is None else _d(fr), _d(x), _d(u),
                                     int(maxiter), C.c_double(tol))
    return u, it
