"""GPU engine: one shooting-problem shard on one MI355X, behind the C ABI of include/aslr_to_amd.h.

PyTorch-ROCm is used for storage and streams only: ONE torch uint8 tensor is the workspace the HIP
library carves into regions; every region is exposed as a zero-copy typed torch view.  All
arithmetic happens in the hand-written HIP kernels (csrc/aslr_*.hip: one translation unit per kernel family and size).  No CPU fallback: if the
library or a GPU is missing this module raises.
"""
import ctypes as C

import numpy as np

from . import _abi
from .lowering import lower_problem


def _torch():
    import torch
    return torch


class Engine(object):
    """ShootingProblem shard resident in HBM.  Region layouts: include/aslr_to_amd.h."""

    def __init__(self, lowered, device=None):
        torch = _torch()
        self.lib = _abi.load_library()
        if not torch.cuda.is_available():
            raise _abi.AslrError("aslr_to_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                                 "there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.low = lowered
        self.B, self.T, self.nx, self.nu, self.rec = lowered.B, lowered.T, lowered.nx, lowered.nu, lowered.rec
        self.nu_user = getattr(lowered, "nu_user", lowered.nu)  # < nu: controls are padded on the device (pad_u / cut_u)
        nbytes = self.lib.aslr_workspace_bytes(C.byref(lowered.desc))
        if nbytes <= 0:
            raise _abi.AslrError("invalid problem description (aslr_workspace_bytes = %d)" % nbytes)
        with torch.cuda.device(self.device):
            self.ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            off = (-self.ws.data_ptr()) % 256
            self.ws = self.ws[off:off + nbytes]
            self.handle = C.c_void_p()
            _abi.check(self.lib.aslr_problem_create(C.byref(lowered.desc), C.c_void_p(self.ws.data_ptr()),
                                                    nbytes, self._stream(), C.byref(self.handle)),
                       "aslr_problem_create")
        self._views = {}

    def _stream(self):
        torch = _torch()
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _call(self, name, *args):
        """One C-ABI call with this engine's device current (the ABI launches on the thread's current device, on a
        stream that belongs to self.device): an Engine on cuda:1 works whatever the caller's current device is."""
        torch = _torch()
        with torch.cuda.device(self.device):
            _abi.check(getattr(self.lib, name)(self.handle, *args), name)

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.aslr_problem_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- zero-copy views of the workspace regions ----
    def region(self, rid):
        torch = _torch()
        if rid in self._views:
            return self._views[rid]
        r = _abi.Region()
        _abi.check(self.lib.aslr_problem_region(self.handle, rid, C.byref(r)), "aslr_problem_region")
        raw = self.ws[r.offset:r.offset + r.bytes]
        B, T, nx, nu, rec = self.B, self.T, self.nx, self.nu, self.rec
        shapes = {
            _abi.R_XS: (T + 1, B, nx), _abi.R_US: (T, B, nu), _abi.R_XNEXT: (T + 1, B, nx),
            _abi.R_COST: (T + 1, B), _abi.R_DERIV: (T + 1, B, rec), _abi.R_GAPS: (T + 1, B, nx),
            _abi.R_KGAIN: (T, B, nu, nx), _abi.R_KFF: (T, B, nu), _abi.R_QU: (T, B, nu),
            _abi.R_VX: (T + 1, B, nx), _abi.R_VXX: (T + 1, B, nx, nx),
            _abi.R_XS_TRY: (_abi.NALPHA, T + 1, B, nx), _abi.R_US_TRY: (_abi.NALPHA, T, B, nu),
            _abi.R_TRAJ_F: (_abi.TF_COUNT, B), _abi.R_X0: (B, nx), _abi.R_FRAME_REF: (B, 12),
            _abi.R_VXXF: (T + 1, B, nx), _abi.R_COST_TRY: (_abi.NALPHA, T + 1, B),
        }
        if rid in (_abi.R_XS_TRY, _abi.R_US_TRY):
            # candidate slabs: for even widths <= 8 piece-interleaved in groups of 4 trajectories
            # ([ceil(B / 4)][w / 2][4][2], include/aslr_to_amd.h) -- returned as a de-interleaved COPY
            # [NALPHA][knots][B][w] (not cached: it would go stale)
            w, knots = (nx, T + 1) if rid == _abi.R_XS_TRY else (nu, T)
            f = raw.view(torch.float64)
            if w % 2 == 0 and w <= 8:
                G = (B + 3) // 4
                return (f.view(_abi.NALPHA, knots, G, w // 2, 4, 2).permute(0, 1, 2, 4, 3, 5)
                        .reshape(_abi.NALPHA, knots, G * 4, w)[:, :, :B].contiguous())
            return f.view(_abi.NALPHA, knots, B, w)
        if rid == _abi.R_TRAJ_I:
            v = raw.view(torch.int32).view(_abi.TI_COUNT, B)
        elif rid == _abi.R_NODE_MODEL:
            v = raw.view(torch.int32)
        elif rid in shapes:
            v = raw.view(torch.float64).view(*shapes[rid])
        else:
            v = raw
        self._views[rid] = v
        return v

    # batch-major views ([B, T, ...]) of the time-major storage
    @property
    def xs(self):
        return self.region(_abi.R_XS).permute(1, 0, 2)

    @property
    def us(self):
        return self.region(_abi.R_US).permute(1, 0, 2)

    def pad_u(self, u):
        """[..., nu_user] -> [..., nu] (zeros in the padded commands); tensors and arrays pass through otherwise."""
        if self.nu_user == self.nu or u.shape[-1] != self.nu_user:
            return u
        torch = _torch()
        if torch.is_tensor(u):
            return torch.cat([u, torch.zeros(u.shape[:-1] + (self.nu - self.nu_user,), dtype=u.dtype, device=u.device)], dim=-1)
        return np.concatenate([u, np.zeros(u.shape[:-1] + (self.nu - self.nu_user,))], axis=-1)

    def cut_u(self, v, axis=-1):
        """the models' own controls out of a device-sized array (axis = the control axis)"""
        if self.nu_user == self.nu:
            return v
        idx = [slice(None)] * v.ndim
        idx[axis] = slice(0, self.nu_user)
        return v[tuple(idx)]

    def deriv_block(self, name):
        """[T+1, B, rows, cols] view of one block of the DERIV records."""
        o = _abi.record_offsets(self.nx, self.nu)
        nx, nu = self.nx, self.nu
        shp = {"Fx": (nx, nx), "Fu": (nx, nu), "Lxx": (nx, nx), "Lxu": (nx, nu), "Luu": (nu, nu),
               "Lx": (nx,), "Lu": (nu,)}[name]
        n = int(np.prod(shp))
        d = self.region(_abi.R_DERIV)
        return d[:, :, o[name]:o[name] + n].reshape(self.T + 1, self.B, *shp)

    def set_candidate(self, xs=None, us=None):
        """xs: [B, T+1, nx] / [T+1, nx] / list of arrays; us likewise.  None -> zeros
        (crocoddyl setCandidate with empty lists: state.zero() and zero controls)."""
        torch = _torch()
        X, U = self.region(_abi.R_XS), self.region(_abi.R_US)
        if xs is None or (hasattr(xs, "__len__") and len(xs) == 0):
            X.zero_()
        else:
            x = torch.as_tensor(np.asarray(xs, dtype=np.float64) if not torch.is_tensor(xs) else xs,
                                dtype=torch.float64, device=self.device)
            if x.dim() == 2:
                x = x.unsqueeze(0).expand(self.B, -1, -1)
            if tuple(x.shape) != (self.B, self.T + 1, self.nx):
                raise ValueError("xs must have shape [B=%d, T+1=%d, nx=%d]" % (self.B, self.T + 1, self.nx))
            X.copy_(x.permute(1, 0, 2))
        if us is None or (hasattr(us, "__len__") and len(us) == 0):
            U.zero_()
        else:
            u = torch.as_tensor(np.asarray(us, dtype=np.float64) if not torch.is_tensor(us) else us,
                                dtype=torch.float64, device=self.device)
            if u.dim() == 2:
                u = u.unsqueeze(0).expand(self.B, -1, -1)
            u = self.pad_u(u)
            if tuple(u.shape) != (self.B, self.T, self.nu):
                raise ValueError("us must have shape [B=%d, T=%d, nu=%d]" % (self.B, self.T, self.nu))
            U.copy_(u.permute(1, 0, 2))

    # ---- the hot path ----
    def calc(self):
        self._call("aslr_calc", self._stream())

    def calc_diff(self):
        self._call("aslr_calc_diff", self._stream())

    def backward_pass(self, sp):
        self._call("aslr_backward_pass", C.byref(sp), self._stream())

    def forward_pass(self, sp):
        self._call("aslr_forward_pass", C.byref(sp), self._stream())

    def iterate(self, sp, first):
        self._call("aslr_iterate", C.byref(sp), 1 if first else 0, self._stream())

    def iterate_n(self, sp, first, n):
        """n lock-step iterations in one ABI call (sub-shards run them free of each other, see set_subshards)."""
        self._call("aslr_iterate_n", C.byref(sp), 1 if first else 0, int(n), self._stream())

    def set_subshards(self, n):
        """Iterate the shard as n (1..4) sub-shards on internal streams: same results, bit for bit; the serial sweeps
        of one sub-shard overlap the streaming kernels of the others (include/aslr_to_amd.h: aslr_set_subshards)."""
        self._call("aslr_set_subshards", int(n))

    def iterate_timed(self, sp, first=False):
        """-> (calc_ms, backward_ms, forward_ms) of one iteration, from HIP events on the launch stream."""
        ms = (C.c_float * 3)()
        self._call("aslr_iterate_timed", C.byref(sp), 1 if first else 0, self._stream(), ms)
        return tuple(float(v) for v in ms)

    def quasi_static(self, maxiter=100, tol=1e-9):
        """Fill US with the quasi-static controls of the states in XS; returns the [T, B] iteration counts."""
        torch = _torch()
        iters = torch.zeros((self.T, self.B), dtype=torch.int32, device=self.device)
        self._call("aslr_quasi_static", int(maxiter), float(tol), C.c_void_p(iters.data_ptr()), self._stream())
        return iters

    def finalize(self):
        self._call("aslr_finalize", self._stream())

    def count_active(self):
        n = C.c_int32(0)
        self._call("aslr_count_active", self._stream(), C.byref(n))
        return n.value

    def solve(self, sp, poll_every=4):
        it = C.c_int32(0)
        self._call("aslr_solve", C.byref(sp), poll_every, self._stream(), C.byref(it))
        return it.value

    # ---- per-iteration log, frame placements, residuals ----
    def enable_iteration_log(self, capacity):
        """Device-resident log of the per-iteration solver state ([capacity, LOG_COUNT, B], NaN = not written): the
        line-search kernel fills it, nothing crosses PCIe until `iteration_log()` is read.  capacity 0 / None: off."""
        torch = _torch()
        if not capacity:
            self._log = None
            self._call("aslr_set_iteration_log", C.c_void_p(0), 0)
            return None
        with torch.cuda.device(self.device):
            self._log = torch.full((int(capacity), _abi.LOG_COUNT, self.B), float("nan"), dtype=torch.float64,
                                   device=self.device)
        self._call("aslr_set_iteration_log", C.c_void_p(self._log.data_ptr()), int(capacity))
        return self._log

    def iteration_log(self):
        """The log tensor ([capacity, LOG_COUNT, B], on the device), or None when logging is off."""
        return getattr(self, "_log", None)

    def frame_placement(self, frame_joint, frame_R, frame_p, x, x_stride=None):
        """World placements of a frame (joint index + local placement) at the link positions of `x`
        (a device tensor [..., >= nj], contiguous): -> ([..., 3, 3], [..., 3]) device tensors."""
        torch = _torch()
        x = x.contiguous()
        stride = x.shape[-1] if x_stride is None else int(x_stride)
        n = x.numel() // x.shape[-1]
        out = torch.empty((n, 12), dtype=torch.float64, device=self.device)
        R = (C.c_double * 9)(*[float(v) for v in np.asarray(frame_R, dtype=np.float64).reshape(9)])
        pv = (C.c_double * 3)(*[float(v) for v in np.asarray(frame_p, dtype=np.float64).reshape(3)])
        self._call("aslr_frame_placement", int(frame_joint), R, pv, n, C.c_void_p(x.data_ptr()), stride,
                   C.c_void_p(out.data_ptr()), self._stream())
        lead = tuple(x.shape[:-1])
        return out[:, :9].reshape(lead + (3, 3)), out[:, 9:].reshape(lead + (3,))

    def dam_residuals(self, model_index, x, u):
        """Stacked cost residuals (data.r) of n points, in the order of the model's cost list: numpy [n, nr]."""
        torch = _torch()
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        u = self.pad_u(np.atleast_2d(np.asarray(u, dtype=np.float64)))
        nr = self.lib.aslr_residual_len(C.byref(self.low.desc.models[model_index]), self.nx // 4)
        if nr < 0:
            raise _abi.AslrError("aslr_residual_len failed")
        dx = torch.as_tensor(x, device=self.device).contiguous()
        du = torch.as_tensor(u, device=self.device).contiguous()
        r = torch.zeros((x.shape[0], max(nr, 1)), dtype=torch.float64, device=self.device)
        if nr > 0:
            self._call("aslr_dam_residuals", model_index, x.shape[0], C.c_void_p(dx.data_ptr()),
                       C.c_void_p(du.data_ptr()), C.c_void_p(r.data_ptr()), self._stream())
        return r[:, :nr].cpu().numpy()

    def solve_pool(self, x0s, frame_refs, sp, refill_every=4, poll_every=16, xs_init=None, us_init=None):
        """A pool of P problems of this engine's structure solved through its B slots, each to its own convergence
        (aslr_solve_pool: stopped slots are flushed and refilled on the device; every problem is cold-started).
        x0s [P, nx], frame_refs [P, 12] or None, xs_init [P, T+1, nx] / us_init [P, T, nu] or None = zeros (host or
        device).  -> dict of device tensors: xs [P, T+1, nx],
        us [P, T, nu], cost / stop / x_reg / step [P], iters / status [P] (int32), and batch_iters (int)."""
        torch = _torch()
        with torch.cuda.device(self.device):
            x0 = torch.as_tensor(x0s, dtype=torch.float64, device=self.device).contiguous()
            P = x0.shape[0]
            if tuple(x0.shape) != (P, self.nx):
                raise ValueError("x0s must have shape [P, nx=%d]" % self.nx)
            fr = None
            if frame_refs is not None:
                fr = torch.as_tensor(frame_refs, dtype=torch.float64, device=self.device).contiguous()
                if tuple(fr.shape) != (P, 12):
                    raise ValueError("frame_refs must have shape [P, 12]")
            xi = ui = None
            if xs_init is not None:
                xi = torch.as_tensor(xs_init, dtype=torch.float64, device=self.device).contiguous()
                if tuple(xi.shape) != (P, self.T + 1, self.nx):
                    raise ValueError("xs_init must have shape [P, T+1, nx]")
            if us_init is not None:
                ui = self.pad_u(torch.as_tensor(us_init, dtype=torch.float64, device=self.device)).contiguous()
                if tuple(ui.shape) != (P, self.T, self.nu):
                    raise ValueError("us_init must have shape [P, T, nu]")
            xs = torch.empty((P, self.T + 1, self.nx), dtype=torch.float64, device=self.device)
            us = torch.empty((P, self.T, self.nu), dtype=torch.float64, device=self.device)
            sf = torch.zeros((P, 4), dtype=torch.float64, device=self.device)
            si = torch.zeros((P, 2), dtype=torch.int32, device=self.device)
            slot = torch.empty((self.B,), dtype=torch.int32, device=self.device)
            cnt = torch.zeros((2,), dtype=torch.int32, device=self.device)
        pool = _abi.Pool()
        pool.P = P
        pool.x0, pool.frame_ref = x0.data_ptr(), (fr.data_ptr() if fr is not None else None)
        pool.xs_out, pool.us_out, pool.stat_f, pool.stat_i = xs.data_ptr(), us.data_ptr(), sf.data_ptr(), si.data_ptr()
        pool.slot_problem, pool.counters = slot.data_ptr(), cnt.data_ptr()
        pool.xs_init = xi.data_ptr() if xi is not None else None
        pool.us_init = ui.data_ptr() if ui is not None else None
        it = C.c_int32(0)
        self._call("aslr_solve_pool", C.byref(sp), C.byref(pool), int(refill_every), int(poll_every), self._stream(),
                   C.byref(it))
        torch.cuda.synchronize(self.device)
        return dict(xs=xs, us=self.cut_u(us), cost=sf[:, 0], stop=sf[:, 1], x_reg=sf[:, 2], step=sf[:, 3], iters=si[:, 0],
                    status=si[:, 1], batch_iters=it.value)

    def traj_f(self, row):
        return self.region(_abi.R_TRAJ_F)[row]

    def traj_i(self, row):
        return self.region(_abi.R_TRAJ_I)[row]

    def dam_eval(self, model_index, x, u):
        """DAM-level calc + calcDiff on n points (numpy in / numpy out)."""
        torch = _torch()
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        u = self.pad_u(np.atleast_2d(np.asarray(u, dtype=np.float64)))
        n, nx, nu, nv = x.shape[0], self.nx, self.nu, self.nx // 2
        dx = torch.as_tensor(x, device=self.device).contiguous()
        du = torch.as_tensor(u, device=self.device).contiguous()
        outs = {"xout": (n, nv), "cost": (n,), "Fx": (n, nv, nx), "Fu": (n, nv, nu), "Lx": (n, nx), "Lu": (n, nu),
                "Lxx": (n, nx, nx), "Lxu": (n, nx, nu), "Luu": (n, nu, nu)}
        t = {k: torch.empty(s, dtype=torch.float64, device=self.device) for k, s in outs.items()}
        p = lambda k: C.c_void_p(t[k].data_ptr())
        self._call("aslr_dam_eval", model_index, n, C.c_void_p(dx.data_ptr()), C.c_void_p(du.data_ptr()), p("xout"),
                   p("cost"), p("Fx"), p("Fu"), p("Lx"), p("Lu"), p("Lxx"), p("Lxu"), p("Luu"), self._stream())
        out = {k: v.cpu().numpy() for k, v in t.items()}
        if self.nu_user != self.nu:
            out["Fu"], out["Lu"], out["Lxu"] = self.cut_u(out["Fu"]), self.cut_u(out["Lu"]), self.cut_u(out["Lxu"])
            out["Luu"] = self.cut_u(self.cut_u(out["Luu"]), axis=-2)
        return out


class _PointEvaluator(object):
    """B = 1, T = 1 problem around one action model: backs `model.calc(data, x, u)`."""

    def __init__(self, model):
        from .models import IntegratedActionModelEulerASR
        if isinstance(model, IntegratedActionModelEulerASR):
            self.iam = model
        else:
            self.iam = IntegratedActionModelEulerASR(model, 0.0)
        self._sig = None
        self.engine = None

    def _ensure(self):
        # re-lower when the user mutated the model (weights, bounds, dt ...) since the last call
        low = lower_problem(np.zeros(self.iam.state.nx), [self.iam], self.iam)
        sig = bytes(memoryview(low.desc.models[0])) + bytes(memoryview(low.desc.chain))
        if sig != self._sig:
            if self.engine is not None:
                self.engine.close()
            self.engine = Engine(low)
            self._sig = sig
        return self.engine

    def dam(self, data, x, u, diff):
        e = self._ensure()
        r = e.dam_eval(0, x, u)
        data.xout[:] = r["xout"][0]
        data.cost = float(r["cost"][0])
        self._side_data(data, x, u)
        if diff:
            for k in ("Fx", "Fu", "Lx", "Lu", "Lxx", "Lxu", "Luu"):
                getattr(data, k)[...] = r[k][0]

    def _side_data(self, ddata, x, u):
        """data.r (the stacked cost residuals, costs.shareMemory: free_fwddyn_asr.py:128-129) and
        data.multibody.pinocchio.oMf of the frames the model knows, from the GPU."""
        e = self.engine
        dam = self.iam.differential
        r = e.dam_residuals(0, x, u)[0]
        ddata.r = dam.costs.order_residuals(r, dam.state.ndx, dam.nu, dam.nu_dev)
        ddata.multibody.pinocchio.oMf = _FrameMap(e, dam.state.pinocchio, np.asarray(x, dtype=np.float64))

    def integrated(self, data, x, u, diff):
        torch = _torch()
        e = self._ensure()
        x = np.asarray(x, dtype=np.float64)
        if u is None:
            u = self.iam.differential._default_u()
        xs = np.stack([x, x])[None]
        e.set_candidate(xs, np.asarray(u, dtype=np.float64)[None, None])
        (e.calc_diff if diff else e.calc)()
        torch.cuda.synchronize(e.device)
        data.xnext[:] = e.region(_abi.R_XNEXT)[0, 0].cpu().numpy()
        data.cost = float(e.region(_abi.R_COST)[0, 0].item())
        data.dx[:] = data.xnext - x
        self._side_data(data.differential, x, u)
        data.differential.cost = data.cost
        if self.iam.withCostResiduals:
            data.r = data.differential.r  # integrated_action.py:17-18
        if diff:
            for k in ("Fx", "Fu", "Lx", "Lu", "Lxx", "Lxu", "Luu"):
                blk = e.deriv_block(k)[0, 0].cpu().numpy()
                if k in ("Fu", "Lu", "Lxu", "Luu"):
                    blk = e.cut_u(blk)
                if k == "Luu":
                    blk = e.cut_u(blk, axis=-2)
                getattr(data, k)[...] = blk


class _SE3View(object):
    """What the scripts read from pinocchio.SE3: .rotation, .translation (numpy)."""

    def __init__(self, R, p):
        self.rotation, self.translation = R, p

    def __repr__(self):
        return "SE3(R=%r, p=%r)" % (self.rotation, self.translation)


class _FrameMap(object):
    """data.pinocchio.oMf: frame id -> placement, evaluated on the GPU (aslr_frame_placement) on first access."""

    def __init__(self, engine, pin_model, x):
        self._e, self._m, self._x, self._cache = engine, pin_model, x, {}

    def __getitem__(self, fid):
        if fid not in self._cache:
            torch = _torch()
            fr = self._m.frames[fid]
            if fr.parent < 0:  # a frame of the universe does not move
                self._cache[fid] = _SE3View(fr.placement.rotation.copy(), fr.placement.translation.copy())
                return self._cache[fid]
            xt = torch.as_tensor(self._x, dtype=torch.float64, device=self._e.device).reshape(1, -1)
            R, p = self._e.frame_placement(fr.parent, fr.placement.rotation, fr.placement.translation, xt)
            self._cache[fid] = _SE3View(R[0].cpu().numpy(), p[0].cpu().numpy())
        return self._cache[fid]

    def __len__(self):
        return len(self._m.frames)


def point_evaluator(model):
    if getattr(model, "_evaluator", None) is None:
        model._evaluator = _PointEvaluator(model)
    return model._evaluator
