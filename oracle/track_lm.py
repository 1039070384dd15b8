"""oracle/track_lm.py -- TEST INFRASTRUCTURE (checker only; nothing under sage_slam_amd/ or bench.py's timed region imports it).

An independent restatement, in numpy fp32, of the reference's tracker Levenberg-Marquardt POLICY, written from the
reference text only -- it shares no code with the product's `sage_track_lm` (sage_slam_amd/csrc/host_math.cpp), which it is
the checker for:

  the loop                     core/system/camera_tracker.cpp:1156-1279 (TrackNewFrame), :1487-1620 (TrackFrame)
  UpdateVariables              core/system/camera_tracker.cpp:467-512   (se3_exp of core/mapping/mapping_utils.h:316-346,
                                                                        left update R <- dR R, t <- dR t + dt, s <- s + ds)
  LMConvergence                core/system/camera_tracker.cpp:527-573   (RotationToAngleAxis: mapping_utils.h:143-213)
  damped solve                 (AtA + damp * diag(AtA)).colPivHouseholderQr().solve(Atb), fp32 (:1182-1183)

The evaluation back-end (ComputeJacobianAndError / ComputeError) is a pair of callbacks, as in the reference's own split:
`lin(pose12, scale) -> (AtA, Atb, error)`, `err(pose12, scale) -> error`.  The damped solve here is LAPACK's column-pivoted
QR in fp32 (scipy.linalg.qr(pivoting=True)) -- the same factorisation family as Eigen's, not its code: solutions agree
with Eigen's to fp32 rounding amplified by the system's condition, which is what the trace comparison's tolerance on the
errors covers (damping values, accept / reject decisions and `update_jac` flags are compared exactly).

Parity pinning: the policy cannot be executed from the reference here (camera_tracker.cpp needs OpenCV / gtsam / TEASER++);
this file is the second, independent reading of it.  Its arithmetic pieces are pinned elsewhere: se3_exp by
tests/golden/diffba_retract.npz, the Eigen solve by tests/golden/colpiv_qr_eigen339.json.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg

F = np.float32


def default_config() -> dict:
    """the tracking_* flags the reference ships (system/sources/... slam_run.flags; SageLmConfig mirrors them)"""
    return dict(max_num_iters=40, min_grad_thresh=1.0e-4, min_param_inc_thresh=1.0e-2, init_damp=1.0e-4, min_damp=1.0e-6,
                max_damp=1.0e-2, damp_dec_factor=10.0, damp_inc_factor=100.0, jac_update_err_inc_threshold=1.0e-2,
                no_overlap_error=0.0)


# ---------------------------------------------------------------------------------------------------------------------
# mapping_utils.h:316-346  se3_exp<float>
# ---------------------------------------------------------------------------------------------------------------------
def _hat(n):
    return np.array([[0, -n[2], n[1]], [n[2], 0, -n[0]], [-n[1], n[0], 0]], F)


def se3_exp(omega, v):
    omega = np.asarray(omega, F); v = np.asarray(v, F)
    theta = F(np.sqrt(F(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2])))
    n = omega / theta if theta > 0 else np.array([1, 0, 0], F)      # "a casual rotation direction vector"
    theta = max(theta, F(1.0e-14))
    s, c = F(np.sin(theta)), F(np.cos(theta))
    K = _hat(n)
    K2 = (K @ K).astype(F)
    I = np.eye(3, dtype=F)
    R = (I + s * K + (F(1.0) - c) * K2).astype(F)
    V = (I + ((F(1.0) - c) / theta) * K + ((theta - s) / theta) * K2).astype(F)
    return R, (V @ v).astype(F)


def update_variables(sol, R, t, scale, dof):
    """camera_tracker.cpp:467-512: tangent = [v(3), omega(3), (ds)]; left multiplication"""
    dR, dt = se3_exp(sol[3:6], sol[0:3])
    Rn = (dR @ R).astype(F)
    tn = ((dR @ t.reshape(3, 1)).reshape(3) + dt).astype(F)
    return Rn, tn, (F(scale) + F(sol[6]) if dof == 7 else F(scale))


# ---------------------------------------------------------------------------------------------------------------------
# mapping_utils.h:143-213  RotationToAngleAxis = QuaternionToAngleAxis(RotationToQuaternion(R, eps)) -- AS WRITTEN:
# the denominator sums t0*mask_c1 + t1*mask_c1 + t2*mask_c2 + t3*mask_c3 (mask_c0 never enters: case c0 divides by 0)
# ---------------------------------------------------------------------------------------------------------------------
def rotation_to_angle_axis(R, eps=F(1.0e-6)):
    m = np.asarray(R, F).reshape(3, 3).T                     # rmat_t = rotation_matrix.permute({1, 0})
    d2 = m[2, 2] < eps
    d0_d1 = m[0, 0] > m[1, 1]
    d0_nd1 = m[0, 0] < -m[1, 1]
    t0 = F(1.0) + m[0, 0] - m[1, 1] - m[2, 2]
    t1 = F(1.0) - m[0, 0] + m[1, 1] - m[2, 2]
    t2 = F(1.0) - m[0, 0] - m[1, 1] + m[2, 2]
    t3 = F(1.0) + m[0, 0] + m[1, 1] + m[2, 2]
    q0 = np.array([m[1, 2] - m[2, 1], t0, m[0, 1] + m[1, 0], m[2, 0] + m[0, 2]], F)
    q1 = np.array([m[2, 0] - m[0, 2], m[0, 1] + m[1, 0], t1, m[1, 2] + m[2, 1]], F)
    q2 = np.array([m[0, 1] - m[1, 0], m[2, 0] + m[0, 2], m[1, 2] + m[2, 1], t2], F)
    q3 = np.array([t3, m[1, 2] - m[2, 1], m[2, 0] - m[0, 2], m[0, 1] - m[1, 0]], F)
    c0, c1 = F(d2 and d0_d1), F(d2 and not d0_d1)
    c2, c3 = F((not d2) and d0_nd1), F((not d2) and not d0_nd1)
    q = q0 * c0 + q1 * c1 + q2 * c2 + q3 * c3
    with np.errstate(divide="ignore", invalid="ignore"):
        q = (F(0.5) * q / np.sqrt(F(t0 * c1 + t1 * c1 + t2 * c2 + t3 * c3))).astype(F)
    sin_sq = F(q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    sin_t = F(np.sqrt(sin_sq))
    cos_t = q[0]
    two_theta = F(np.arctan2(-sin_t, -cos_t)) if cos_t < 0.0 else F(np.arctan2(sin_t, cos_t))
    with np.errstate(divide="ignore", invalid="ignore"):
        k = two_theta / sin_t if sin_sq > 0.0 else F(2.0)
    return (k * q[1:4]).astype(F)


def lm_convergence(cfg, R, t, scale, Atb, sol, dof):
    """camera_tracker.cpp:527-573: max |Atb| below min_grad_thresh, or the max (SIGNED, as written) relative parameter
    increment below min_param_inc_thresh.  A NaN anywhere makes torch::max NaN and both comparisons false."""
    rv = rotation_to_angle_axis(R)
    den = np.abs(np.concatenate([np.asarray(t, F).reshape(-1), rv] + ([np.array([scale], F)] if dof == 7 else []))).astype(F)
    max_grad = F(np.max(np.abs(np.asarray(Atb, F))))
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = (np.asarray(sol, F)[:dof] / (den + F(1.0e-8))).astype(F)
    max_inc = F(np.nan) if np.isnan(ratio).any() else F(np.max(ratio))
    return bool(max_grad < F(cfg["min_grad_thresh"]) or max_inc < F(cfg["min_param_inc_thresh"]))


def damped_solve(AtA, AtA_diag, Atb, damp):
    """(AtA + damp * diag(AtA)).colPivHouseholderQr().solve(Atb) in fp32 -- here through LAPACK sgeqp3"""
    A = (AtA + F(damp) * AtA_diag).astype(F)
    Q, Rm, piv = scipy.linalg.qr(A, pivoting=True)
    n = A.shape[0]
    # Eigen keeps every pivot that is not exactly negligible (nonzeroPivots()); sgeqp3 has no cut either: a triangular
    # solve over the non-zero diagonal entries, zeros for exactly-zero pivots
    y = (Q.T @ np.asarray(Atb, F)).astype(F)
    r = int(np.sum(np.abs(np.diag(Rm)) > 0))
    x = np.zeros(n, F)
    if r > 0:
        x[piv[:r]] = scipy.linalg.solve_triangular(Rm[:r, :r], y[:r]).astype(F)
    return x


def track_lm(cfg: dict, dof: int, lin, err, pose12, scale):
    """Returns (pose12, scale, final_error, iters, trace).  trace: one entry per outer iteration that reached the inner
    loop -- {damp (after the inner loop), error (curr_error), candidate_error, accepted, relinearized (update_jac),
    inner_evals} -- the quantities the product's SageLmTraceEntry records."""
    p = np.asarray(pose12, F).reshape(12)
    R, t = p[:9].reshape(3, 3).copy(), p[9:].copy()
    s = F(scale)
    prev_error, curr_error = F(0.0), F(1.0)                              # :1057-1058
    curr_iter, damp = 0, F(cfg["init_damp"])
    AtA = AtA_diag = Atb = None
    trace = []
    clamp = lambda d: F(min(max(F(cfg["min_damp"]), F(d)), F(cfg["max_damp"])))
    while True:
        with np.errstate(divide="ignore", invalid="ignore"):
            rel_change = F(abs(F(curr_error - prev_error))) / prev_error      # (1 - 0) / 0 = inf on the first pass
        if rel_change > F(cfg["jac_update_err_inc_threshold"]):              # :1159
            A, g, e = lin(np.concatenate([R.reshape(-1), t]).astype(F), float(s))
            AtA = np.asarray(A, F).reshape(dof, dof); Atb = np.asarray(g, F).reshape(dof)
            AtA_diag = np.diag(np.diag(AtA)).astype(F)
            if curr_iter == 0:                                                # update_error only on the first pass (:1166)
                curr_error = F(e)
            update_jac = True
        else:
            update_jac = False
        if cfg.get("no_overlap_error", 0.0) > 0 and curr_error >= F(cfg["no_overlap_error"]):      # :1515-1519
            return np.concatenate([R.reshape(-1), t]).astype(F), float(s), float(curr_error), curr_iter, trace, "no_overlap"
        curr_iter += 1
        sol = damped_solve(AtA, AtA_diag, Atb, damp)                          # :1182-1184
        if lm_convergence(cfg, R, t, s, Atb, sol, dof):                       # :1186
            break
        inner = 0
        while True:
            Rc, tc, sc = update_variables(sol, R, t, s, dof)                  # :1200
            cand_error = F(err(np.concatenate([Rc.reshape(-1), tc]).astype(F), float(sc)))
            inner += 1
            if cand_error < curr_error:                                       # :1218
                break
            elif damp < F(cfg["max_damp"]):                                   # :1223
                damp = clamp(damp * F(cfg["damp_inc_factor"]))
                sol = damped_solve(AtA, AtA_diag, Atb, damp)
            else:
                break
        accepted = bool(cand_error < curr_error)
        trace.append(dict(damp=float(damp), error=float(curr_error), candidate_error=float(cand_error),
                          accepted=int(accepted), relinearized=int(update_jac), inner_evals=inner))
        if cand_error >= curr_error and damp >= F(cfg["max_damp"]):          # :1255
            break
        R, t, s = Rc, tc, sc                                                  # :1262-1263
        if update_jac:
            prev_error = curr_error
        curr_error = cand_error
        damp = clamp(damp / F(cfg["damp_dec_factor"]))
        if curr_iter >= cfg["max_num_iters"]:                                 # :1273
            break
    return np.concatenate([R.reshape(-1), t]).astype(F), float(s), float(curr_error), curr_iter, trace, "ok"
