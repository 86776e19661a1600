"""Shared by tests/test_gpu_headline.py and tools/parity_headline.py: trajectory-by-trajectory comparison of a GPU
solve with the CPU oracle's, using the per-iteration logs of both sides (aslr_set_iteration_log / aslr_cpu_solve_log).

For every trajectory it finds the FIRST iteration at which a discrete solver decision differs -- the accepted
line-search index, the regularisation after the update, a status bit, or whether the trajectory stopped -- and reports
the two values that were compared on either side of that decision (dV against th_acceptstep * dVexp for a line-search
accept, stop against th_stop for the exit).  A trajectory with no such iteration took the same decisions on both sides
throughout; its iterates then differ by accumulated rounding only.
"""
import numpy as np

from aslr_to_amd import _abi as A


def first_decision_flip(lg, lr, b):
    """lg, lr: logs [n, LOG_COUNT, B] (NaN = not written).  -> None or dict describing the first differing decision."""
    n = max(lg.shape[0], lr.shape[0])
    for k in range(n):
        g = lg[k, :, b] if k < lg.shape[0] else np.full(A.LOG_COUNT, np.nan)
        r = lr[k, :, b] if k < lr.shape[0] else np.full(A.LOG_COUNT, np.nan)
        gon, ron = not np.isnan(g[A.LOG_COST]), not np.isnan(r[A.LOG_COST])
        if not gon and not ron:
            return None
        kind = None
        if gon != ron:
            kind = "exit"           # one side stopped after iteration k-1, the other went on
        elif g[A.LOG_ACCEPTED] != r[A.LOG_ACCEPTED]:
            kind = "line search"
        elif g[A.LOG_XREG] != r[A.LOG_XREG]:
            kind = "regularisation"
        elif g[A.LOG_STATUS] != r[A.LOG_STATUS]:
            kind = "status"
        if kind:
            prev_g = lg[k - 1, :, b] if k > 0 else g
            prev_r = lr[k - 1, :, b] if k > 0 else r
            return dict(iteration=k, kind=kind, gpu=g.copy(), oracle=r.copy(), prev_gpu=prev_g.copy(),
                        prev_oracle=prev_r.copy())
    return None


def first_cost_drift(lg, lr, b, rel=1e-9):
    """First iteration at which the logged costs of both sides differ by more than `rel` (relative): where a difference
    that is not a logged decision -- a BoxQP active set, an ill-conditioned step -- first shows.  -> (k, cost_g, cost_r) or None"""
    n = min(lg.shape[0], lr.shape[0])
    for k in range(n):
        a, c = lg[k, A.LOG_COST, b], lr[k, A.LOG_COST, b]
        if np.isnan(a) or np.isnan(c):
            return None
        if abs(a - c) > rel * max(1.0, abs(c)):
            return k, float(a), float(c)
    return None


def compare(gpu, ref, sp):
    """gpu / ref: dict(xs [T+1,B,nx], us, traj_f, traj_i, log).  -> dict of per-trajectory arrays and the list of
    exceptions (trajectories whose iteration count, status, or converged iterates beyond 1e-6 / 1e-4 differ)."""
    it_g, it_r = gpu["traj_i"][A.TI_ITER], ref["traj_i"][A.TI_ITER]
    st_g, st_r = gpu["traj_i"][A.TI_STATUS], ref["traj_i"][A.TI_STATUS]
    conv_g, conv_r = (st_g & A.ST_CONVERGED) != 0, (st_r & A.ST_CONVERGED) != 0
    dx = np.abs(gpu["xs"] - ref["xs"]).max(axis=(0, 2))
    du = np.abs(gpu["us"] - ref["us"]).max(axis=(0, 2))
    dc = np.abs(gpu["traj_f"][A.TF_COST] - ref["traj_f"][A.TF_COST])
    both = conv_g & conv_r
    within = (dx < 1e-6) & (du < 1e-6) & (dc < 1e-4)
    odd = (it_g != it_r) | (st_g != st_r) | (both & ~within)
    rows = []
    for b in np.nonzero(odd)[0]:
        flip = first_decision_flip(gpu["log"], ref["log"], int(b))
        drift = first_cost_drift(gpu["log"], ref["log"], int(b))
        rows.append(dict(b=int(b), drift=drift, it_gpu=int(it_g[b]), it_oracle=int(it_r[b]), st_gpu=int(st_g[b]), st_oracle=int(st_r[b]),
                         dx=float(dx[b]), du=float(du[b]), dcost=float(dc[b]),
                         stop_gpu=float(gpu["traj_f"][A.TF_STOP][b]), stop_oracle=float(ref["traj_f"][A.TF_STOP][b]),
                         flip=flip))
    return dict(it_same=int((it_g == it_r).sum()), st_same=int((st_g == st_r).sum()), conv_both=int(both.sum()),
                conv_gpu=int(conv_g.sum()), conv_oracle=int(conv_r.sum()), within=int((both & within).sum()),
                max_dx=float(dx[both & within].max()) if (both & within).any() else 0.0,
                max_du=float(du[both & within].max()) if (both & within).any() else 0.0,
                max_dc=float(dc[both & within].max()) if (both & within).any() else 0.0,
                dx=dx, du=du, dc=dc, exceptions=rows)


def describe(row, sp):
    """One text line per exception: the decision that fell the other way and its margin."""
    f = row["flip"]
    head = ("traj %4d: iterations gpu %3d / oracle %3d, status %2d / %2d, |dx| %.2e |du| %.2e |dcost| %.2e, "
            "final stop %.3e / %.3e" % (row["b"], row["it_gpu"], row["it_oracle"], row["st_gpu"], row["st_oracle"],
                                        row["dx"], row["du"], row["dcost"], row["stop_gpu"], row["stop_oracle"]))
    d = row.get("drift")
    if d is not None and (f is None or d[0] < f["iteration"]):
        head += (" | costs part ways (> 1e-9 relative) at iteration %d: %.12e / %.12e, with the same logged decisions "
                 "up to there (a BoxQP active set or an ill-conditioned step)" % d)
    if f is None:
        return head + " | same decisions at every iteration (rounding drift only)"
    g, r, k = f["gpu"], f["oracle"], f["iteration"]
    if f["kind"] == "exit":
        pg, pr = f["prev_gpu"], f["prev_oracle"]
        return head + (" | first flip: EXIT test after iteration %d: stop %.6e (gpu) / %.6e (oracle) against th_stop %.1e"
                       % (k - 1, pg[A.LOG_STOP], pr[A.LOG_STOP], sp.th_stop))
    if f["kind"] == "line search":
        def margin(v):  # the accept test dV > th_acceptstep * dVexp at the step length that side stopped at
            return v[A.LOG_DV] - sp.th_acceptstep * v[A.LOG_DVEXP]
        return head + (" | first flip: LINE SEARCH at iteration %d: accepted index %d (gpu) / %d (oracle); "
                       "dV - %.1f dVexp at the stopping step = %.3e (gpu, dV %.6e) / %.3e (oracle, dV %.6e); cost before %.9e / %.9e"
                       % (k, int(g[A.LOG_ACCEPTED]), int(r[A.LOG_ACCEPTED]), sp.th_acceptstep, margin(g), g[A.LOG_DV],
                          margin(r), r[A.LOG_DV], f["prev_gpu"][A.LOG_COST], f["prev_oracle"][A.LOG_COST]))
    return head + (" | first flip: %s at iteration %d: xreg %.1e / %.1e, status %d / %d"
                   % (f["kind"].upper(), k, g[A.LOG_XREG], r[A.LOG_XREG], int(g[A.LOG_STATUS]), int(r[A.LOG_STATUS])))


def assert_status_words_match(st_gpu, st_ref, max_note_flips=None):
    """Status words: every bit that reflects a solver DECISION or OUTCOME (converged, regularisation at its maximum, a
    backward pass that had to be regularised) must agree exactly.  ASLR_ST_FORWARD_ERR is a NOTE that some rejected
    line-search trial overflowed (NaN / Inf / |x| or cost >= 1e30).  Such a trial is an unstable rollout that grows by
    ~2x per knot; it amplifies the 1e-13 difference between the two sides' gains by the same factor it amplifies the
    state, so by the time |x| nears 1e15 (cost 1e30) the two rollouts differ by orders of magnitude and one may cross
    the threshold within the horizon while the other does not -- both reject that step length either way
    (profiles/r02/forward_err_probe_traj1005.txt shows one case knot by knot: with the GPU's own gains the oracle
    overflows at the same knot; with its own, 1e-13 away, it stays below).  So that bit may differ on a few
    trajectories in a thousand; the caller checks separately that the iterates agree."""
    st_gpu, st_ref = np.asarray(st_gpu).astype(np.int64), np.asarray(st_ref).astype(np.int64)
    keep = ~np.int64(A.ST_FORWARD_ERR)
    np.testing.assert_array_equal(st_gpu & keep, st_ref & keep)
    flips = int(((st_gpu ^ st_ref) & A.ST_FORWARD_ERR != 0).sum())
    limit = max(1, st_gpu.size // 500) if max_note_flips is None else max_note_flips
    assert flips <= limit, "ASLR_ST_FORWARD_ERR differs on %d of %d trajectories" % (flips, st_gpu.size)
    return flips
