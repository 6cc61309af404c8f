"""oracle/twin.py — an INDEPENDENT second restatement of the BA arithmetic: visual, inertial and prior factors, the LM loop (TEST INFRASTRUCTURE ONLY).

Written from the reference's source lines and from Ceres Solver 2.2.0's published trust-region algorithm, WITHOUT going
through oracle/*.c: plain NumPy, generic in the scalar type — float64, numpy.longdouble (x87 80-bit) or mpmath `mpf`
(50+ digits, arrays of dtype=object). It pins the C oracle (and through it the HIP path) on the headline `localMapBA`
path, for which the reference holds no golden vector:

  * factor arithmetic:   ReprojectionErrCeres_pointxd_dx::Evaluate  cpp/include/isaeslam/optimizers/BundleAdjustmentCERESAnalytic.h:52-90
                         Camera::project                            cpp/src/data/sensors/Camera.cpp:84-139
                         AngularErrCeres_pointxd_dx::Evaluate       cpp/include/isaeslam/optimizers/AngularAdjustmentCERESAnalytic.h:55-111
                         PosePriordx::Evaluate                      cpp/include/isaeslam/optimizers/residuals.hpp:607-628
                         IMUPriordx / PoseToLandmarkFactor / Landmark3DPrior / LandmarkToLandmarkFactor   residuals.hpp:506-700
                         IMUFactor / IMUBiasFactor::Evaluate (the VIO window of localMapVIOptimization)      residuals.hpp:133-296
                         MarginalizationFactor::Evaluate            cpp/include/isaeslam/optimizers/marginalization.hpp:113-215
                         Marginalization::computeSchurComplement / rankReveallingDecomposition / computeJacobiansAndResiduals /
                         sparsifyVIO / sparsifyVO (the factor informations)   cpp/src/optimizers/marginalization.cpp:213-265,318-342,362-530
                         SO(3) helpers                              cpp/include/utilities/geometry.h:17-37,131-166
  * the problem of addResidualsLocalMap (BundleAdjustmentCERESAnalytic.cpp:197-314) on a flat window;
  * Ceres 2.2.0 TrustRegionMinimizer + LevenbergMarquardtStrategy (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc,
    corrector.cc, as published; solver options AOptimizer.cpp:315-323) on the UN-REDUCED normal equations
    (J^T J + D^2) y = J^T r — the reference's SPARSE_NORMAL_CHOLESKY solve — by a dense factorisation, no Schur complement.

tests/golden/make_golden.py generates the committed golden vectors from THIS module (float64, cross-checked against the
50-digit evaluation); tests/test_twin.py compares the C oracle with it. Nothing under sadvio_amd/ may import it.
"""
from __future__ import annotations

import numpy as np

try:
    import mpmath
except Exception:  # pragma: no cover
    mpmath = None


# ---------------------------------------------------------------------------------------------------------------
# scalar back ends
# ---------------------------------------------------------------------------------------------------------------
class Backend:
    """Scalar type + the handful of transcendental functions the factors need."""

    def __init__(self, kind="f64", digits=50):
        self.kind = kind
        if kind == "f64":
            self.dtype = np.float64
        elif kind == "ld":
            self.dtype = np.longdouble
        elif kind == "mp":
            if mpmath is None:
                raise RuntimeError("mpmath is not importable")
            mpmath.mp.dps = digits
            self.dtype = object
        else:
            raise ValueError(kind)

    def s(self, x):
        """scalar"""
        if self.kind == "mp":
            return x if isinstance(x, mpmath.mpf) else mpmath.mpf(float(x)) if not isinstance(x, (int, str)) else mpmath.mpf(x)
        return self.dtype(x)

    def a(self, x):
        """array"""
        x = np.asarray(x)
        if self.kind == "mp":
            out = np.empty(x.shape, dtype=object)
            for idx in np.ndindex(x.shape):
                out[idx] = self.s(x[idx])
            return out
        return x.astype(self.dtype)

    def zeros(self, shape):
        if self.kind == "mp":
            out = np.empty(shape, dtype=object)
            out.fill(mpmath.mpf(0))
            return out
        return np.zeros(shape, dtype=self.dtype)

    def eye(self, n):
        m = self.zeros((n, n))
        for i in range(n):
            m[i, i] = self.s(1)
        return m

    def _f(self, name, x):
        if self.kind == "mp":
            return getattr(mpmath, name)(x)
        return getattr(np, {"acos": "arccos"}.get(name, name))(x)

    def sqrt(self, x): return self._f("sqrt", x)
    def sin(self, x): return self._f("sin", x)
    def cos(self, x): return self._f("cos", x)
    def acos(self, x): return self._f("acos", x)

    def isfinite(self, x):
        if self.kind == "mp":
            return mpmath.isfinite(x)
        return bool(np.isfinite(x))

    def f(self, x):
        """to float64 (scalar or array)"""
        if isinstance(x, np.ndarray):
            return np.array([float(v) for v in x.ravel()], dtype=np.float64).reshape(x.shape)
        return float(x)


def norm(B, v):
    return B.sqrt(sum(x * x for x in v))


# ---------------------------------------------------------------------------------------------------------------
# geometry.h
# ---------------------------------------------------------------------------------------------------------------
def skew(B, w):                                                  # geometry.h:17-23
    z = B.s(0)
    return np.array([[z, -w[2], w[1]], [w[2], z, -w[0]], [-w[1], w[0], z]], dtype=B.dtype)


def vee(B, S):                                                   # FromskewMatrix, geometry.h:25-28
    return np.array([S[2, 1], S[0, 2], S[1, 0]], dtype=B.dtype)


def so3_right_jacobian(B, w):                                    # geometry.h:30-37
    n = norm(B, w)
    if n < 1e-5:
        return B.eye(3)
    S = skew(B, w)
    return B.eye(3) - ((1 - B.cos(n)) / (n * n)) * S + ((n - B.sin(n)) / (n * n * n)) * (S @ S)


def exp_so3(B, v):                                               # geometry.h:131-147
    angle = norm(B, v)
    if angle < 1e-9:
        return B.eye(3) + skew(B, v)
    S = skew(B, v / angle)
    return B.eye(3) + (1 - B.cos(angle)) * (S @ S) + B.sin(angle) * S


def log_so3(B, M):                                               # geometry.h:149-166
    c = (M[0, 0] + M[1, 1] + M[2, 2]) / 2 - B.s(1) / 2
    c = min(max(c, B.s(-1)), B.s(1))
    angle = B.acos(c)
    d = vee(B, M - M.T)
    if abs(B.sin(angle)) < 1e-9 or angle < 1e-9:
        return d / 2
    return (angle / (2 * B.sin(angle))) * d


def inv3(B, A):
    """3x3 inverse by the adjugate (what Eigen's fixed-size .inverse() computes)."""
    c = B.zeros((3, 3))
    c[0, 0] = A[1, 1] * A[2, 2] - A[1, 2] * A[2, 1]
    c[0, 1] = A[0, 2] * A[2, 1] - A[0, 1] * A[2, 2]
    c[0, 2] = A[0, 1] * A[1, 2] - A[0, 2] * A[1, 1]
    c[1, 0] = A[1, 2] * A[2, 0] - A[1, 0] * A[2, 2]
    c[1, 1] = A[0, 0] * A[2, 2] - A[0, 2] * A[2, 0]
    c[1, 2] = A[0, 2] * A[1, 0] - A[0, 0] * A[1, 2]
    c[2, 0] = A[1, 0] * A[2, 1] - A[1, 1] * A[2, 0]
    c[2, 1] = A[0, 1] * A[2, 0] - A[0, 0] * A[2, 1]
    c[2, 2] = A[0, 0] * A[1, 1] - A[0, 1] * A[1, 0]
    det = A[0, 0] * c[0, 0] + A[0, 1] * c[1, 0] + A[0, 2] * c[2, 0]
    return c / det


def split_T(B, T12):
    T = B.a(np.asarray(T12).reshape(12))
    return T[:9].reshape(3, 3), T[9:12].copy()


# ---------------------------------------------------------------------------------------------------------------
# factors
# ---------------------------------------------------------------------------------------------------------------
def pixel_factor(B, T_f_w0, K, T_s_f, p0, uv, sigma, dpose, dl):
    """ReprojectionErrCeres_pointxd_dx::Evaluate + Camera::project. Returns r[2], J_pose[2,6], J_lmk[2,3], valid."""
    R0, t0 = split_T(B, T_f_w0)
    Rs, ts = split_T(B, T_s_f)
    K = B.a(K); p0 = B.a(p0); uv = B.a(uv); dpose = B.a(dpose); dl = B.a(dl)
    w = 1 / B.s(sigma)                                           # info_sqrt_ = (1 / sigma) I   (…Analytic.h:46)
    dR = exp_so3(B, dpose[:3])
    R = R0 @ dR                                                  # T_f_w = T_f_w0 * (exp(w), t)  (…Analytic.h:54-55)
    t = R0 @ dpose[3:6] + t0
    pw = p0 + dl                                                 # T_w_lmk * (I, dl), point landmark (…Analytic.h:58)
    tc = Rs @ (R @ pw + t) + ts                                  # Camera.cpp:92-93
    fx, fy, cx, cy = K
    Kc = np.array([[fx, B.s(0), cx], [B.s(0), fy, cy], [B.s(0), B.s(0), B.s(1)]], dtype=B.dtype)
    pt = Kc @ tc                                                 # :96
    z = pt[2]
    Jh = np.array([[1 / z, B.s(0), -pt[0] / (z * z)], [B.s(0), 1 / z, -pt[1] / (z * z)]], dtype=B.dtype)   # :97-99
    p2d = np.array([pt[0] / z, pt[1] / z], dtype=B.dtype)        # :101-102
    Jint = B.zeros((3, 6))                                       # :104-110
    Jint[:, :3] = -(R @ skew(B, pw) @ so3_right_jacobian(B, log_so3(B, R)))
    Jint[:, 3:] = B.eye(3)
    Jframe = w * (Jh @ Kc @ Rs @ Jint)                           # :112-115
    Jlmk = w * (Jh @ Kc @ (Rs @ R))                              # :118-125
    valid = True                                                 # :127-137
    if tc[2] < 0.1:
        valid = False
    if p2d[0] < 0 or p2d[1] < 0 or p2d[0] > 2 * cx or p2d[1] > 2 * cy:
        valid = False
    if not (B.isfinite(p2d[0]) and B.isfinite(p2d[1])):
        valid = False
    r = w * (p2d - uv) if valid else B.zeros(2)                  # …Analytic.h:62-67
    Jl = B.zeros((6, 6))                                         # :70-77
    Jl[:3, :3] = inv3(B, so3_right_jacobian(B, log_so3(B, R))) @ so3_right_jacobian(B, dpose[:3])
    Jl[3:, 3:] = R0
    return r, Jframe @ Jl, Jlmk, valid


def angular_factor(B, T_f_w0, T_s_f, p0, bearing, sigma, dpose, dl):
    """AngularErrCeres_pointxd_dx::Evaluate. Returns r[2], J_pose[2,6], J_lmk[2,3]."""
    R0, t0 = split_T(B, T_f_w0)
    Rs, ts = split_T(B, T_s_f)
    p0 = B.a(p0); b = B.a(bearing); dpose = B.a(dpose); dl = B.a(dl)
    w = 1 / B.s(sigma)                                           # :60
    dR = exp_so3(B, dpose[:3])
    q = p0 + dl
    pf = R0 @ (dR @ q + dpose[3:6]) + t0                         # _T_f_w * dT * (_t_w_lmk + dt)   (:63)
    tsl = Rs @ pf + ts
    nrm = norm(B, tsl)
    bs = tsl / nrm                                               # :64-65
    ex = np.array([B.s(1), B.s(0), B.s(0)], dtype=B.dtype)
    ez = np.array([B.s(0), B.s(0), B.s(1)], dtype=B.dtype)
    cross = lambda u, v: np.array([u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]], dtype=B.dtype)
    b1 = cross(b, ex) if norm(B, b - ex) > 1e-5 else cross(b, ez)   # :68-75
    b1 = b1 / norm(B, b1)
    b2 = cross(b1, b)
    b2 = b2 / norm(B, b2)                                        # :77-78
    Pt = np.stack([b1, b2])                                      # :80-83
    r = w * (Pt @ (bs - b))                                      # :86
    Jel = Pt @ (B.eye(3) - np.outer(bs, bs)) @ Rs @ R0 / nrm     # :90-92
    Jbf = B.zeros((3, 6))                                        # :95-99
    Jbf[:, :3] = -(dR @ skew(B, q) @ so3_right_jacobian(B, log_so3(B, dR)))
    Jbf[:, 3:] = B.eye(3)
    return r, w * (Jel @ Jbf), w * (Jel @ dR)                    # :102, :107


def T4(B, R, t):
    T = B.eye(4)
    T[:3, :3] = R; T[:3, 3] = t
    return T


def line_pixel_factor(B, T_f_w0, K, T_s_f, T_w_l0, model6, uv4, sigma, dpose, dline):
    """ReprojectionErrCeres_linexd_dx::Evaluate (BundleAdjustmentCERESAnalytic.h:116-166), Jacobian branch: the two model points,
    carried by T_w_lmk = T_w_lmk0 * se3_doubleVec3dtoRT(parameters[1]) — the first THREE entries of the line's 6-vector read as a
    translation (:121) — projected with Camera::project against the two measured end points. Blocks as coded:
    J_frame = jac0 * blockdiag(Jr(log R_f_w)^-1 Jr(w), R_f_w0) (:145-151), J_lmk = jac1 * [-R_w_lmk [pt]x | I] (:154-161).
    Returns r[4], J_frame[4,6], J_line[4,6]."""
    Rl, tl = split_T(B, T_w_l0)
    model6 = B.a(model6); uv4 = B.a(uv4); dline = B.a(dline)
    t_w_l = tl + Rl @ dline[:3]                                  # T_w_lmk_ * (I, x[0..2])
    r = B.zeros(4); Jf = B.zeros((4, 6)); Jl = B.zeros((4, 6))
    for i in range(2):
        pt = model6[3 * i: 3 * i + 3]
        pw = Rl @ pt + t_w_l                                     # T_w_lmk * Tpt, translation part (:124-129)
        ri, Jpi, Jli, _ = pixel_factor(B, T_f_w0, K, T_s_f, pw, uv4[2 * i: 2 * i + 2], sigma, dpose, B.zeros(3))
        r[2 * i: 2 * i + 2] = ri                                 # invalid projection: zero residual, Jacobian kept (:137-141)
        Jf[2 * i: 2 * i + 2] = Jpi
        JP = B.zeros((3, 6))
        JP[:, :3] = -(Rl @ skew(B, pt)); JP[:, 3:] = B.eye(3)
        Jl[2 * i: 2 * i + 2] = Jli @ JP
    return r, Jf, Jl


def line_angular_factor(B, T_f_w0, T_s_f, T_w_l0, bearings6, sigma, dpose, dline):
    """AngularErrCeres_linexd_dx::Evaluate (AngularAdjustmentCERESAnalytic.h:378-459) with the helper definitions of
    utilities/geometry.h:328-342 AS CODED (J_norm(X) = X^T / |X|, J_normalization(X) = (I - X X^T) / |X| with X not normalised,
    J_AcrossX(A) = -[A]x). Returns r[2], J_frame[2,6], J_line[2,6]."""
    R0, t0 = split_T(B, T_f_w0); Rs, ts = split_T(B, T_s_f); Rl, tl = split_T(B, T_w_l0)
    dpose = B.a(dpose); dline = B.a(dline); bb = B.a(bearings6)
    cross = lambda u, v: np.array([u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]], dtype=B.dtype)
    Jnz = lambda X: (B.eye(3) - np.outer(X, X)) / norm(B, X)
    dT = T4(B, exp_so3(B, dpose[:3]), dpose[3:6]); dL = T4(B, exp_so3(B, dline[:3]), dline[3:6])
    Tsf, Tfw, Twl = T4(B, Rs, ts), T4(B, R0, t0), T4(B, Rl, tl)
    weight = 1 / (B.s(sigma) * B.s(sigma))                       # :382
    Tsl = Tsf @ Tfw @ dT @ Twl @ dL                              # :384
    Rsl, tsl = Tsl[:3, :3], Tsl[:3, 3]
    n_obs = cross(bb[:3], bb[3:6]); n_obs = n_obs / norm(B, n_obs)   # :388-389
    b_l = tsl / norm(B, tsl)                                     # :392
    ex = np.array([B.s(1), B.s(0), B.s(0)], dtype=B.dtype)
    d = Rsl @ ex; d = d / norm(B, d)                             # Rotation2directionVector, geometry.h:125-129
    n_l = cross(b_l, d); n_ln = n_l / norm(B, n_l)               # :393-394
    cx = cross(n_obs, n_ln)
    r = np.array([weight * norm(B, cx), weight * (n_obs @ b_l)], dtype=B.dtype)   # :400-401
    J_e0 = (cx / norm(B, cx))[None, :] @ (-skew(B, n_obs)) @ Jnz(n_l)             # :407-408
    J_e1 = n_obs[None, :] @ Jnz(tsl)                                              # :411
    Rsw = Rs @ R0
    dR = dT[:3, :3]; dRl = dL[:3, :3]
    Jr_dT = so3_right_jacobian(B, log_so3(B, dR)); Jr_dL = so3_right_jacobian(B, log_so3(B, dRl))
    Jt_dT = B.zeros((3, 6))                                      # :413-419
    Jt_dT[:, :3] = -(Rsw @ dR @ skew(B, Rl @ dL[:3, 3]) @ Jr_dT) - (Rsw @ dR @ skew(B, tl))
    Jt_dT[:, 3:] = Rsw
    JR_dT = B.zeros((3, 6))                                      # :421-425
    JR_dT[:, :3] = -(Rsw @ dR @ skew(B, Rl @ dRl @ ex) @ Jr_dT)
    Rsl0 = Rsw @ dR @ Rl                                         # (_T_s_f * _T_f_w * dT * _T_w_lmk).rotation()
    Jt_dL = B.zeros((3, 6)); Jt_dL[:, 3:] = Rsl0                 # :427-428
    JR_dL = B.zeros((3, 6)); JR_dL[:, :3] = -(Rsl0 @ dRl @ skew(B, ex) @ Jr_dL)   # :430-433
    out = []
    for Jt, JR in ((Jt_dT, JR_dT), (Jt_dL, JR_dL)):              # :439-458
        row0 = weight * (J_e0 @ (skew(B, Rsl @ ex).T @ (Jnz(tsl) @ Jt) + skew(B, n_ln) @ JR))
        row1 = weight * (J_e1 @ Jt)
        out.append(np.vstack([row0, row1]))
    return r, out[0], out[1]


def pose_prior_factor(B, T_f_w0, T_prior, inf_diag, dpose):
    """PosePriordx::Evaluate, sqrt_inf = diag(inf_diag) (BundleAdjustmentCERESAnalytic.cpp:226). Returns r[6], J[6,6]."""
    R0, t0 = split_T(B, T_f_w0)
    Rp, tp = split_T(B, T_prior)
    dpose = B.a(dpose); W = B.a(inf_diag)
    dR = exp_so3(B, dpose[:3])
    R = R0 @ dR                                                  # T = _T * dT   (:609)
    t = R0 @ dpose[3:6] + t0
    Rpi = Rp.T                                                   # _T_prior.inverse()
    tpi = -(Rpi @ tp)
    Re = R @ Rpi                                                 # T * T_prior^-1
    te = R @ tpi + t
    err = np.concatenate([log_so3(B, Re), te])                   # se3_RTtoVec6d   (:610)
    J = B.eye(6)                                                 # :613-623
    wv = log_so3(B, R @ Rp.T)
    J[:3, :3] = inv3(B, so3_right_jacobian(B, wv)) @ Rp @ so3_right_jacobian(B, dpose[:3])
    J[3:, :3] = R @ skew(B, Rp.T @ tp) @ so3_right_jacobian(B, dpose[:3])
    J[3:, 3:] = R0
    return W * err, W[:, None] * J


def relative_pose_factor(B, Ta, Tb, Tab_prior, W, da, db):
    """Relative6DPose::Evaluate (residuals.hpp:70-131): r[6], Ja[6,6], Jb[6,6]. Ta / Tb are the transforms the deltas are
    composed on (T_w_a, T_w_b in the reference's only use, BundleAdjustmentCERESAnalytic.cpp:787-790)."""
    Ra, ta = split_T(B, Ta); Rb, tb = split_T(B, Tb); Rp, tp = split_T(B, Tab_prior)
    da = B.a(da); db = B.a(db); W = B.a(np.asarray(W).reshape(6, 6))
    Rau = Ra @ exp_so3(B, da[:3]); tau = Ra @ da[3:6] + ta             # T_w_a_up  (:80)
    Rbu = Rb @ exp_so3(B, db[:3]); tbu = Rb @ db[3:6] + tb             # T_w_b_up  (:81)
    Rpi = Rp.T; tpi = -(Rpi @ tp)                                      # T_b_a_prior  (:82)
    Rai = Rau.T; tai = -(Rai @ tau)
    R1 = Rpi @ Rai; t1 = Rpi @ tai + tpi
    R = R1 @ Rbu; t = R1 @ tbu + t1                                     # T  (:83)
    w = log_so3(B, R)
    err = W @ np.concatenate([w, t])                                    # :84
    Jri = inv3(B, so3_right_jacobian(B, w))
    Ja = B.eye(6)
    Ja[:3, :3] = -(Jri @ Rbu.T @ Rau @ so3_right_jacobian(B, da[:3]))   # :98-99
    Ja[3:, :3] = Rpi @ Rau.T @ skew(B, tbu - tau) @ Rau @ so3_right_jacobian(B, da[:3])   # :102-104
    Ja[3:, 3:] = -Rpi                                                   # :107
    Jb = B.eye(6)
    Jb[:3, :3] = Jri @ so3_right_jacobian(B, db[:3])                    # :118
    Jb[3:, 3:] = Rpi @ Rau.T @ Rbu                                      # :121
    return err, W @ Ja, W @ Jb


# ---------------------------------------------------------------------------------------------------------------
# sparse (NFR) prior factors, residuals.hpp:506-700 — as coded
# ---------------------------------------------------------------------------------------------------------------
def imu_prior_factor(B, T_f_w0, v0, ba0, bg0, T_prior, v_prior, ba_prior, bg_prior, sqrt_inf, params15):
    """IMUPriordx::Evaluate (residuals.hpp:649-695): r[15], J[15,15] over [pose 6 | dv 3 | dba 3 | dbg 3]. As coded the pose
    block is sqrt_inf * J_pose (all 15 rows), while the v / ba / bg blocks are plain identities in their own rows — NOT
    multiplied by sqrt_inf although the residual is (:678, :684, :690)."""
    W = B.a(np.asarray(sqrt_inf).reshape(15, 15)); p = B.a(params15)
    R0, t0 = split_T(B, T_f_w0); Rp, tp = split_T(B, T_prior)
    dw = p[:3]
    dR = exp_so3(B, dw)
    R = R0 @ dR; t = R0 @ p[3:6] + t0                              # T = _T * dT   (:664)
    Rpi = Rp.T; tpi = -(Rpi @ tp)
    Re = R @ Rpi; te = R @ tpi + t
    err = B.zeros(15)
    err[:3] = log_so3(B, Re); err[3:6] = te                         # se3_RTtoVec6d(T * T_prior^-1)   (:665)
    err[6:9] = B.a(v0) + p[6:9] - B.a(v_prior)                      # :666-668
    err[9:12] = B.a(ba0) + p[9:12] - B.a(ba_prior)
    err[12:15] = B.a(bg0) + p[12:15] - B.a(bg_prior)
    r = W @ err                                                     # :669
    Jp = B.zeros((15, 6))                                           # :674-686
    wv = log_so3(B, R @ Rp.T)
    Jp[:3, :3] = inv3(B, so3_right_jacobian(B, wv)) @ Rp @ so3_right_jacobian(B, dw)
    Jp[3:6, :3] = R @ skew(B, Rp.T @ tp) @ so3_right_jacobian(B, dw)
    Jp[3:6, 3:6] = R0
    J = B.zeros((15, 15))
    J[:, :6] = W @ Jp
    for q in range(9):
        J[6 + q, 6 + q] = B.s(1)                                    # :690-707
    return r, J


def chol_lower(B, A):
    """L (lower) with A = L L^T — Eigen::LLT::matrixL()."""
    n = A.shape[0]
    L = B.zeros((n, n))
    for j in range(n):
        d = A[j, j] - sum(L[j, k] * L[j, k] for k in range(j))
        L[j, j] = B.sqrt(d)
        for i in range(j + 1, n):
            L[i, j] = (A[i, j] - sum(L[i, k] * L[j, k] for k in range(j))) / L[j, j]
    return L


GRAVITY = (0.0, 0.0, -9.81)                                       # `g`, cpp/include/isaeslam/data/sensors/IMU.h:8


def imu_factor(B, f, T_fi_w0, T_fj_w0, v_i0, v_j0, dpose_i, dpose_j, dv_i, dv_j, dba, dbg):
    """IMUFactor::Evaluate (residuals.hpp:137-241): r[9] and the six Jacobian blocks [pose_i 9x6, pose_j 9x6, v_i 9x3, v_j 9x3, ba_i 9x3,
    bg_i 9x3], whitened by inf_sqrt = LLT(cov^-1).matrixL()^T (:151-154). `f`: the pre-integration of frame j (delta_R / delta_v /
    delta_p, the five bias Jacobians, cov 9x9, dt). As coded: the pose_i translation block is the UNPERTURBED R_i0 (:185), the
    pose_i rotation block of r_dp uses p_j - v_i dt - g dt^2 / 2 without p_i (:181-184), pose_j's translation block is
    -R_i exp(w_j)^T (:198)."""
    g = B.a(GRAVITY)
    Ri0, ti0 = split_T(B, T_fi_w0); Rj0, tj0 = split_T(B, T_fj_w0)
    dpi, dpj = B.a(dpose_i), B.a(dpose_j)
    Ri = Ri0 @ exp_so3(B, dpi[:3]); ti = Ri0 @ dpi[3:6] + ti0       # T_fi_w = T0 * (exp w, t)   (:140-143)
    Rj = Rj0 @ exp_so3(B, dpj[:3]); tj = Rj0 @ dpj[3:6] + tj0
    vi = B.a(v_i0) + B.a(dv_i); vj = B.a(v_j0) + B.a(dv_j)          # :144-145
    dba, dbg = B.a(dba), B.a(dbg)
    dt = B.s(f["dt"])
    cov = B.a(np.asarray(f["cov"], dtype=np.float64).reshape(9, 9))
    W = chol_lower(B, _inverse(B, cov)).T                           # :151-154
    DR = B.a(np.asarray(f["delta_R"], dtype=np.float64).reshape(3, 3))
    Dv = B.a(np.asarray(f["delta_v"], dtype=np.float64)); Dp = B.a(np.asarray(f["delta_p"], dtype=np.float64))
    JRg = B.a(np.asarray(f["J_dR_bg"], dtype=np.float64).reshape(3, 3))
    Jva = B.a(np.asarray(f["J_dv_ba"], dtype=np.float64).reshape(3, 3)); Jvg = B.a(np.asarray(f["J_dv_bg"], dtype=np.float64).reshape(3, 3))
    Jpa = B.a(np.asarray(f["J_dp_ba"], dtype=np.float64).reshape(3, 3)); Jpg = B.a(np.asarray(f["J_dp_bg"], dtype=np.float64).reshape(3, 3))
    dR = (DR @ exp_so3(B, JRg @ dbg)).T @ Ri @ Rj.T                 # :157-158
    r_dr = log_so3(B, dR)
    pi = -(inv3(B, Ri) @ ti); pj = -(inv3(B, Rj) @ tj)              # T.inverse().translation() of an Eigen::Affine3d
    r_dv = Ri @ (vj - vi - g * dt) - (Dv + Jvg @ dbg + Jva @ dba)   # :160-161
    r_dp = Ri @ (pj - pi - vi * dt - g * (dt * dt) / 2) - (Dp + Jpg @ dbg + Jpa @ dba)   # :162-164
    r = W @ np.concatenate([r_dr, r_dv, r_dp])
    Jri = inv3(B, so3_right_jacobian(B, r_dr))
    Jwi = so3_right_jacobian(B, dpi[:3]); Jwj = so3_right_jacobian(B, dpj[:3])
    Ji = B.zeros((9, 6))                                             # :174-187
    Ji[0:3, 0:3] = Jri @ Rj @ Jwi
    Ji[3:6, 0:3] = -(Ri @ skew(B, vj - vi - g * dt) @ Jwi)
    Ji[6:9, 0:3] = -(Ri @ skew(B, pj - vi * dt - g * (dt * dt) / 2) @ Jwi)
    Ji[6:9, 3:6] = Ri0
    Jj = B.zeros((9, 6))                                             # :190-200
    Jj[0:3, 0:3] = -(Jri @ Rj @ Jwj)
    Jj[6:9, 0:3] = -(Ri @ Rj.T @ skew(B, tj) @ Rj @ Jwj)
    Jj[6:9, 3:6] = -(Ri @ exp_so3(B, dpj[:3]).T)
    Jvi = B.zeros((9, 3)); Jvi[3:6] = -Ri; Jvi[6:9] = -(Ri * dt)     # :203-209
    Jvj = B.zeros((9, 3)); Jvj[3:6] = Ri                             # :212-217
    Jba = B.zeros((9, 3)); Jba[3:6] = -Jva; Jba[6:9] = -Jpa          # :220-226
    Jbg = B.zeros((9, 3))                                            # :229-237
    Jbg[0:3] = -(Jri @ dR.T @ so3_right_jacobian(B, JRg @ dbg) @ JRg)
    Jbg[3:6] = -Jvg; Jbg[6:9] = -Jpg
    return r, [W @ Ji, W @ Jj, W @ Jvi, W @ Jvj, W @ Jba, W @ Jbg]


def imu_factor_init(B, f, T_fi_w, T_fj_w, v_i0, v_j0, r_wi2, dv_i, dv_j, dba, dbg, lam):
    """IMUFactorInit::Evaluate (residuals.hpp:302-410), the factor of AOptimizer::VIInit: r[9] and the Jacobian blocks [r_wi 9x2, v_i 9x3,
    v_j 9x3, ba 9x3, bg 9x3, lambda 9x1]. As coded the scale block has no exp(lambda) factor (:397-400)."""
    g = B.a(GRAVITY)
    Ri, ti = split_T(B, T_fi_w); Rj, tj = split_T(B, T_fj_w)
    w_wi = np.array([B.s(r_wi2[0]), B.s(r_wi2[1]), B.s(0)], dtype=B.dtype)      # :309
    Rwi = exp_so3(B, w_wi)
    vi = B.a(v_i0) + B.a(dv_i); vj = B.a(v_j0) + B.a(dv_j)
    dba, dbg = B.a(dba), B.a(dbg)
    lam = B.s(lam)
    el = mpmath.exp(lam) if B.kind == "mp" else np.exp(lam)
    dt = B.s(f["dt"])
    cov = B.a(np.asarray(f["cov"], dtype=np.float64).reshape(9, 9))
    W = chol_lower(B, _inverse(B, cov)).T
    DR = B.a(np.asarray(f["delta_R"], dtype=np.float64).reshape(3, 3))
    Dv = B.a(np.asarray(f["delta_v"], dtype=np.float64)); Dp = B.a(np.asarray(f["delta_p"], dtype=np.float64))
    JRg = B.a(np.asarray(f["J_dR_bg"], dtype=np.float64).reshape(3, 3))
    Jva = B.a(np.asarray(f["J_dv_ba"], dtype=np.float64).reshape(3, 3)); Jvg = B.a(np.asarray(f["J_dv_bg"], dtype=np.float64).reshape(3, 3))
    Jpa = B.a(np.asarray(f["J_dp_ba"], dtype=np.float64).reshape(3, 3)); Jpg = B.a(np.asarray(f["J_dp_bg"], dtype=np.float64).reshape(3, 3))
    dR = (DR @ exp_so3(B, JRg @ dbg)).T @ Ri @ Rj.T                 # :326-327
    r_dr = log_so3(B, dR)
    pi = -(inv3(B, Ri) @ ti); pj = -(inv3(B, Rj) @ tj)
    RR = Ri @ Rwi
    a = (vj - vi) - g * dt
    b = el * (pj - pi) - vi * dt - g * (dt * dt) / 2
    r_dv = RR @ a - (Dv + Jvg @ dbg + Jva @ dba)                    # :329-330
    r_dp = RR @ b - (Dp + Jpg @ dbg + Jpa @ dba)                    # :331-335
    r = W @ np.concatenate([r_dr, r_dv, r_dp])
    Jr2 = so3_right_jacobian(B, w_wi)[:, :2]
    Jw = B.zeros((9, 2)); Jw[3:6] = -(RR @ skew(B, a) @ Jr2); Jw[6:9] = -(RR @ skew(B, b) @ Jr2)     # :346-357
    Jvi = B.zeros((9, 3)); Jvi[3:6] = -RR; Jvi[6:9] = -(RR * dt)                                   # :360-366
    Jvj = B.zeros((9, 3)); Jvj[3:6] = RR                                                          # :369-374
    Jba = B.zeros((9, 3)); Jba[3:6] = -Jva; Jba[6:9] = -Jpa                                       # :377-383
    Jbg = B.zeros((9, 3))                                                                         # :386-394
    Jbg[0:3] = -(inv3(B, so3_right_jacobian(B, r_dr)) @ dR.T @ so3_right_jacobian(B, JRg @ dbg) @ JRg)
    Jbg[3:6] = -Jvg; Jbg[6:9] = -Jpg
    Jl = B.zeros((9, 1)); Jl[6:9, 0] = RR @ (pj - pi)                                             # :397-400 (no exp(lambda))
    return r, [W @ Jw, W @ Jvi, W @ Jvj, W @ Jba, W @ Jbg, W @ Jl]


def imu_bias_factor(B, f, ba_i0, bg_i0, ba_j0, bg_j0, dba_i, dbg_i, dba_j, dbg_j):
    """IMUBiasFactor::Evaluate (residuals.hpp:252-296): r[6] = [(ba_j + dba_j - ba_i - dba_i) / sqrt(dt s_ba^2); (bg ...) / sqrt(dt s_bg^2)],
    Jacobian blocks [dba_i, dbg_i, dba_j, dbg_j], each 6x3."""
    dt = B.s(f["dt"])
    sa = 1 / B.sqrt(dt * B.s(f["bacc_noise"]) * B.s(f["bacc_noise"]))
    sg = 1 / B.sqrt(dt * B.s(f["bgyr_noise"]) * B.s(f["bgyr_noise"]))
    r = np.concatenate([sa * (B.a(ba_j0) + B.a(dba_j) - B.a(ba_i0) - B.a(dba_i)), sg * (B.a(bg_j0) + B.a(dbg_j) - B.a(bg_i0) - B.a(dbg_i))])
    I3 = B.eye(3)
    Ja_i = B.zeros((6, 3)); Ja_i[0:3] = -sa * I3
    Jg_i = B.zeros((6, 3)); Jg_i[3:6] = -sg * I3
    Ja_j = B.zeros((6, 3)); Ja_j[0:3] = sa * I3
    Jg_j = B.zeros((6, 3)); Jg_j[3:6] = sg * I3
    return r, [Ja_i, Jg_i, Ja_j, Jg_j]


def pose_to_landmark_factor(B, T_f_w0, p0, delta, sqrt_inf, dpose, dl):
    """PoseToLandmarkFactor::Evaluate (residuals.hpp:570-595): r[3], J[3,9] over [pose 6 | landmark 3]."""
    W = B.a(np.asarray(sqrt_inf).reshape(-1)[:9].reshape(3, 3)); dpose = B.a(dpose)
    R0, t0 = split_T(B, T_f_w0)
    dR = exp_so3(B, dpose[:3])
    R = R0 @ dR; t = R0 @ dpose[3:6] + t0                           # :578
    pl = B.a(p0) + B.a(dl)                                          # :579
    r = W @ (R @ pl + t - B.a(delta))                               # :581
    J = B.zeros((3, 9))
    J[:, :3] = W @ R0 @ (-(dR @ skew(B, pl) @ so3_right_jacobian(B, dpose[:3])))   # :588-589
    J[:, 3:6] = W @ R0                                              # :590
    J[:, 6:9] = W @ R                                               # :595
    return r, J


def landmark_prior_factor(B, p0, prior, sqrt_inf, dl):
    """Landmark3DPrior::Evaluate (residuals.hpp:512-522): r[3], J[3,3]."""
    W = B.a(np.asarray(sqrt_inf).reshape(-1)[:9].reshape(3, 3))
    return W @ (B.a(p0) + B.a(dl) - B.a(prior)), W


def landmark_to_landmark_factor(B, p0, p1, delta, sqrt_inf, dl0, dl1):
    """LandmarkToLandmarkFactor::Evaluate (residuals.hpp:537-556): r[3], J[3,6] over [landmark 0 | landmark 1]."""
    W = B.a(np.asarray(sqrt_inf).reshape(-1)[:9].reshape(3, 3))
    r = W @ ((B.a(p0) + B.a(dl0)) - (B.a(p1) + B.a(dl1)) - B.a(delta))
    return r, np.concatenate([W, -W], axis=1)


# ---------------------------------------------------------------------------------------------------------------
# marginalisation algebra, marginalization.cpp:213-265, 318-342, 362-408, 491-514, 516-530 and the factor it feeds,
# MarginalizationFactor::Evaluate (marginalization.hpp:113-215)
# ---------------------------------------------------------------------------------------------------------------
def sym_eig(B, A):
    """(eigenvalues ascending, eigenvectors in columns) of a symmetric matrix: Eigen::SelfAdjointEigenSolver."""
    n = A.shape[0]
    if B.kind == "mp":
        E, Q = mpmath.eigsy(mpmath.matrix(A.tolist()))
        lam = np.array([E[i] for i in range(n)], dtype=object)
        V = np.array([[Q[i, j] for j in range(n)] for i in range(n)], dtype=object)
        order = sorted(range(n), key=lambda i: lam[i])
        return lam[order], V[:, order]
    lam, V = np.linalg.eigh(np.asarray(A, dtype=np.float64))
    return B.a(lam), B.a(V)


def schur_prior(B, A, b, m, cut=1e-12):
    """computeSchurComplement + rankReveallingDecomposition + computeJacobiansAndResiduals on A = sum J^T J, b = sum J^T r
    (first m columns marginalised). `cut`: the eigenvalue threshold — the reference's absolute _eps = 1e-12, or a callable
    lambda_max -> threshold for the documented relative floor of oracle/marg.c. Returns Ak, bk, U, Lambda, J, r0, n_full."""
    A = B.a(A); b = B.a(b)
    thr = (lambda lmax: cut(lmax)) if callable(cut) else (lambda lmax: cut)
    Amm = (A[:m, :m] + A[:m, :m].T) / 2                             # :232
    lam, V = sym_eig(B, Amm)
    t = thr(max(abs(x) for x in lam))
    inv = np.array([1 / x if x > t else B.s(0) for x in lam], dtype=B.dtype)   # :234-238
    Ammi = (V * inv[None, :]) @ V.T
    Arm = A[m:, :m]; Arr = A[m:, m:]
    Ak = Arr - Arm @ Ammi @ Arm.T                                   # :245
    bk = b[m:] - Arm @ Ammi @ b[:m]                                 # :246
    lk, Uf = sym_eig(B, (Ak + Ak.T) / 2)                            # rankReveallingDecomposition (:318-342)
    t = thr(max(abs(x) for x in lk))
    keep = [i for i in range(len(lk)) if lk[i] > t]
    U = Uf[:, keep]; Lam = lk[keep]
    sq = np.array([B.sqrt(x) for x in Lam], dtype=B.dtype)
    J = sq[:, None] * U.T                                           # :524   J = Lambda^1/2 U^T
    r0 = -((U.T @ bk) / sq)                                         # :525   r0 = -Lambda^-1/2 U^T bk
    return dict(Ak=Ak, bk=bk, U=U, Lambda=Lam, J=J, r0=r0, n_full=len(keep))


def marginalization_factor(B, J, r0, dx):
    """MarginalizationFactor::Evaluate (marginalization.hpp:113-215): r = r0 + J dx; the Jacobian of a parameter block is the
    matching column slice of J."""
    J = B.a(J)
    return B.a(r0) + J @ B.a(dx), J


def nfr_sqrt_information(B, Jf, U, Sigma, invert_covariance_eigenvalues, cut=1e-12):
    """The square-root information of one NFR factor with Jacobian Jf (rows x n) against the prior (U, Sigma = Lambda^-1).
    sparsifyVIO (marginalization.cpp:377-383, :399-404): inf = (J~ Sigma J~^T)^-1 inverted as a MATRIX first, then the symmetric
    square root of its eigen-decomposition (eigenvalues <= cut dropped). sparsifyVO (:491-498, :506-512): the eigen-decomposition
    of the covariance J~ Sigma J~^T itself, eigenvalues inverted (`invert_covariance_eigenvalues`), then the square root."""
    Jt = B.a(Jf) @ U
    cov = (Jt * B.a(Sigma)[None, :]) @ Jt.T
    if invert_covariance_eigenvalues:
        lam, V = sym_eig(B, (cov + cov.T) / 2)
        s = np.array([B.sqrt(1 / x) if x > cut else B.s(0) for x in lam], dtype=B.dtype)
    else:
        n = cov.shape[0]
        inf = inv3(B, cov) if n == 3 else _inverse(B, cov)
        lam, V = sym_eig(B, (inf + inf.T) / 2)
        s = np.array([B.sqrt(x) if x > cut else B.s(0) for x in lam], dtype=B.dtype)
    return (V * s[None, :]) @ V.T


def _inverse(B, A):
    """Dense inverse by Gauss-Jordan with partial pivoting (Eigen's MatrixXd::inverse() is a PartialPivLU)."""
    n = A.shape[0]
    M = np.concatenate([A.copy(), B.eye(n)], axis=1)
    for c in range(n):
        p = max(range(c, n), key=lambda i: abs(M[i, c]))
        if p != c:
            M[[c, p]] = M[[p, c]]
        M[c] = M[c] / M[c, c]
        for i in range(n):
            if i != c:
                M[i] = M[i] - M[i, c] * M[c]
    return M[:, n:]


def sparsify_vio_jacobians(B, T_f_w, kf_col, lmk_cols, n):
    """The factor Jacobians sparsifyVIO projects the prior on (marginalization.cpp:369-376, :393-398): one 3 x n per kept
    landmark (pose-to-landmark) and the 15 x n of the absolute factor of the kept frame, as coded."""
    R, t = split_T(B, T_f_w)
    Js = []
    for c in lmk_cols:
        J = B.zeros((3, n))
        J[:, c:c + 3] = R
        J[:, kf_col:kf_col + 3] = -(R @ skew(B, t))
        J[:, kf_col + 3:kf_col + 6] = R
        Js.append(J)
    Ja = B.zeros((15, n))
    for q in range(15):
        Ja[q, kf_col + q] = B.s(1)
    Ja[:3, kf_col:kf_col + 3] = R
    Ja[:3, kf_col + 3:kf_col + 6] = R
    Ja[3:6, kf_col + 3:kf_col + 6] = R
    return Js, Ja


def huber(B, s, a):
    """ceres::HuberLoss(a)::Evaluate -> (rho, rho'), loss_function.cc; s = |r|^2."""
    b = B.s(a) * B.s(a)
    if s > b:
        r = B.sqrt(s)
        return 2 * B.s(a) * r - b, max(B.s(a) / r, B.s(0) + np.finfo(np.float64).tiny)
    return s, B.s(1)


# ---------------------------------------------------------------------------------------------------------------
# the problem of addResidualsLocalMap on a flat window (duck-typed: attribute names of sadvio_flat_window)
# ---------------------------------------------------------------------------------------------------------------
class Problem:
    """Parameter blocks: free key-frames (6 each, window order) then free landmarks (3 each, window order). A residual
    block all of whose parameters are constant contributes to fixed_cost only (Ceres' reduced program)."""

    def __init__(self, B, w, huber_a=0.0):
        self.B, self.w, self.huber_a = B, w, huber_a
        self.n_kf = int(np.asarray(w.kf_T_f_w).reshape(-1, 12).shape[0])
        self.n_lmk = int(np.asarray(w.lmk_p).reshape(-1, 3).shape[0])
        kc = np.asarray(w.kf_const).astype(bool)
        lc = np.zeros(self.n_lmk, bool) if getattr(w, "lmk_const", None) is None else np.asarray(w.lmk_const).astype(bool)
        ptr = np.asarray(w.lmk_obs_ptr)
        self.ptr = ptr
        # a landmark is a parameter block only if some residual block uses it
        has_obs = (ptr[1:] - ptr[:-1]) > 0
        self.sparse = list(getattr(w, "sparse_priors", None) or [])
        for f in self.sparse:                                     # ... an NFR factor of the sparsified prior does
            for key in ("lmk0", "lmk1"):
                if int(f.get(key, -1)) >= 0:
                    has_obs[int(f[key])] = True
        dp0 = getattr(w, "dense_prior", None)
        if dp0 is not None:                                       # ... the marginalisation factor does
            for li, lc0 in zip(dp0["lmk_index"], dp0["lmk_col"]):
                if lc0 >= 0:
                    has_obs[int(li)] = True
        self.kf_col = np.full(self.n_kf, -1)
        self.lmk_col = np.full(self.n_lmk, -1)
        n = 0
        for k in range(self.n_kf):
            if not kc[k]:
                self.kf_col[k] = n
                n += 6
        self.n_pose = n
        for l in range(self.n_lmk):
            if not lc[l] and has_obs[l]:
                self.lmk_col[l] = n
                n += 3
        # VIO windows (localMapVIOptimization, AOptimizer.cpp:352-446): v, ba, bg of every free key-frame (3 each); constant with the frame
        self.has_imu = bool(getattr(w, "has_imu", 0))
        self.v_col = np.full(self.n_kf, -1); self.ba_col = np.full(self.n_kf, -1); self.bg_col = np.full(self.n_kf, -1)
        if self.has_imu:
            for k in range(self.n_kf):
                if not kc[k]:
                    self.v_col[k], self.ba_col[k], self.bg_col[k] = n, n + 3, n + 6
                    n += 9
            self.vel = np.asarray(w.kf_vel, dtype=np.float64).reshape(-1, 3)
            self.ba0 = np.asarray(w.kf_ba, dtype=np.float64).reshape(-1, 3)
            self.bg0 = np.asarray(w.kf_bg, dtype=np.float64).reshape(-1, 3)
            self.imu = list(getattr(w, "imu_factors", []))
        # linexd landmarks (PoseParametersBlock, 6 each): a parameter block iff free and observed
        self.lines = getattr(w, "lines", None)
        self.n_line = 0
        self.line_col = np.zeros(0, dtype=np.int64)
        if self.lines is not None:
            L = self.lines
            self.line_T = np.asarray(L["T_w_l"], dtype=np.float64).reshape(-1, 12)
            self.n_line = self.line_T.shape[0]
            self.line_model = np.asarray(L["model"], dtype=np.float64).reshape(-1, 6)
            self.line_ptr = np.asarray(L["obs_ptr"]); self.line_okf = np.asarray(L["obs_kf"]); self.line_ocam = np.asarray(L["obs_cam"])
            self.line_meas = np.asarray(L["obs_meas"], dtype=np.float64)
            lconst = np.zeros(self.n_line, bool) if L.get("const") is None else np.asarray(L["const"]).astype(bool)
            self.line_col = np.full(self.n_line, -1)
            for l in range(self.n_line):
                if not lconst[l] and self.line_ptr[l + 1] > self.line_ptr[l]:
                    self.line_col[l] = n
                    n += 6
        self.n = n
        self.T = np.asarray(w.kf_T_f_w, dtype=np.float64).reshape(-1, 12)
        self.K = np.asarray(w.cam_K, dtype=np.float64).reshape(-1, 4)
        self.Ts = np.asarray(w.cam_T_s_f, dtype=np.float64).reshape(-1, 12)
        self.sig = np.asarray(w.cam_sigma, dtype=np.float64)
        self.P = np.asarray(w.lmk_p, dtype=np.float64).reshape(-1, 3)
        self.meas = np.asarray(w.obs_meas, dtype=np.float64)
        self.okf, self.ocam = np.asarray(w.obs_kf), np.asarray(w.obs_cam)
        self.priors = list(getattr(w, "pose_priors", []))
        self.dense = getattr(w, "dense_prior", None)

    def split(self, x):
        """x (length n, scalar type of B) -> per key-frame 6-vectors, per landmark 3-vectors (zeros when constant)."""
        B = self.B
        xp = B.zeros((self.n_kf, 6)); xl = B.zeros((self.n_lmk, 3))
        for k in range(self.n_kf):
            if self.kf_col[k] >= 0:
                xp[k] = x[self.kf_col[k]: self.kf_col[k] + 6]
        for l in range(self.n_lmk):
            if self.lmk_col[l] >= 0:
                xl[l] = x[self.lmk_col[l]: self.lmk_col[l] + 3]
        return xp, xl

    def split_lines(self, x):
        xs = self.B.zeros((self.n_line, 6))
        for l in range(self.n_line):
            if self.line_col[l] >= 0:
                xs[l] = x[self.line_col[l]: self.line_col[l] + 6]
        return xs

    def split_vio(self, x):
        """x -> per key-frame dv, dba, dbg (zeros when constant / no IMU)."""
        B = self.B
        xv = B.zeros((self.n_kf, 3)); xa = B.zeros((self.n_kf, 3)); xg = B.zeros((self.n_kf, 3))
        for k in range(self.n_kf):
            if self.v_col[k] >= 0:
                xv[k] = x[self.v_col[k]: self.v_col[k] + 3]; xa[k] = x[self.ba_col[k]: self.ba_col[k] + 3]; xg[k] = x[self.bg_col[k]: self.bg_col[k] + 3]
        return xv, xa, xg

    def blocks(self, x, want_j=True):
        """Yield (r, [(col, J)], in_program) for every residual block at x, loss function already applied (Corrector)."""
        B = self.B
        xp, xl = self.split(x)
        pixel = int(self.w.factor_type) == 0
        for l in range(self.n_lmk):
            for o in range(self.ptr[l], self.ptr[l + 1]):
                k, c = int(self.okf[o]), int(self.ocam[o])
                if pixel:
                    r, Jp, Jl, _ = pixel_factor(B, self.T[k], self.K[c], self.Ts[c], self.P[l], self.meas[o][:2], self.sig[c], xp[k], xl[l])
                else:
                    r, Jp, Jl = angular_factor(B, self.T[k], self.Ts[c], self.P[l], self.meas[o][:3], self.sig[c], xp[k], xl[l])
                s = r[0] * r[0] + r[1] * r[1]
                rho = s
                if self.huber_a > 0:                              # corrector.cc: rho'' <= 0 -> scale r and J by sqrt(rho')
                    rho, d1 = huber(B, s, self.huber_a)
                    sc = B.sqrt(d1)
                    r, Jp, Jl = sc * r, sc * Jp, sc * Jl
                cols = []
                if self.kf_col[k] >= 0:
                    cols.append((self.kf_col[k], Jp))
                if self.lmk_col[l] >= 0:
                    cols.append((self.lmk_col[l], Jl))
                yield r, cols, bool(cols), rho
        if self.n_line:                                           # linexd blocks (…Analytic.cpp:273-311 / Angular….cpp:293-333), the caller's loss function
            xs = self.split_lines(x)
            for l in range(self.n_line):
                for o in range(self.line_ptr[l], self.line_ptr[l + 1]):
                    k, c = int(self.line_okf[o]), int(self.line_ocam[o])
                    if pixel:
                        r, Jf, Jl = line_pixel_factor(B, self.T[k], self.K[c], self.Ts[c], self.line_T[l], self.line_model[l], self.line_meas[o][:4], 1.0, xp[k], xs[l])
                    else:
                        r, Jf, Jl = line_angular_factor(B, self.T[k], self.Ts[c], self.line_T[l], self.line_meas[o][:6], 1.0, xp[k], xs[l])
                    s = sum(v * v for v in r)
                    rho = s
                    if self.huber_a > 0:
                        rho, d1 = huber(B, s, self.huber_a)
                        sc = B.sqrt(d1)
                        r, Jf, Jl = sc * r, sc * Jf, sc * Jl
                    cols = [(cc, Jb) for cc, Jb in ((self.kf_col[k], Jf), (self.line_col[l], Jl)) if cc >= 0]
                    yield r, cols, bool(cols), rho
        for (k, Tp, inf) in self.priors:                          # PosePriordx blocks (…Analytic.cpp:224-228)
            k = int(k)
            r, J = pose_prior_factor(B, self.T[k], Tp, inf, xp[k])
            cols = [(self.kf_col[k], J)] if self.kf_col[k] >= 0 else []
            yield r, cols, bool(cols), sum(v * v for v in r)
        if self.has_imu:                                          # addIMUResiduals (AOptimizer.cpp:22-96): IMUFactor + IMUBiasFactor per pair
            xv, xa, xg = self.split_vio(x)
            for f in self.imu:
                i, j = int(f["kf_i"]), int(f["kf_j"])
                r, Js = imu_factor(B, f, self.T[i], self.T[j], self.vel[i], self.vel[j], xp[i], xp[j], xv[i], xv[j], xa[i], xg[i])
                cols = [(c, Jb) for c, Jb in zip((self.kf_col[i], self.kf_col[j], self.v_col[i], self.v_col[j], self.ba_col[i], self.bg_col[i]), Js) if c >= 0]
                yield r, cols, bool(cols), sum(v * v for v in r)
                r, Js = imu_bias_factor(B, f, self.ba0[i], self.bg0[i], self.ba0[j], self.bg0[j], xa[i], xg[i], xa[j], xg[j])
                cols = [(c, Jb) for c, Jb in zip((self.ba_col[i], self.bg_col[i], self.ba_col[j], self.bg_col[j]), Js) if c >= 0]
                yield r, cols, bool(cols), sum(v * v for v in r)
        for f in self.sparse:                                     # sparse branch of addMarginalizationResiduals (…Analytic.cpp:363-426)
            t = int(f["type"])
            if t == 0:                                            # IMUPriordx on the kept frame
                k = int(f["kf"])
                xv, xa, xg = self.split_vio(x)
                r, J = imu_prior_factor(B, self.T[k], self.vel[k], self.ba0[k], self.bg0[k], f["T_prior"], f["v_prior"], f["ba_prior"], f["bg_prior"],
                                        f["sqrt_inf"], np.concatenate([xp[k], xv[k], xa[k], xg[k]]))
                cols = [(c, J[:, o: o + wdt]) for c, o, wdt in ((self.kf_col[k], 0, 6), (self.v_col[k], 6, 3), (self.ba_col[k], 9, 3), (self.bg_col[k], 12, 3)) if c >= 0]
            elif t == 1:                                          # PoseToLandmarkFactor
                k, l = int(f["kf"]), int(f["lmk0"])
                r, J = pose_to_landmark_factor(B, self.T[k], self.P[l], f["delta"], f["sqrt_inf"], xp[k], xl[l])
                cols = [(c, Jb) for c, Jb in ((self.kf_col[k], J[:, :6]), (self.lmk_col[l], J[:, 6:9])) if c >= 0]
            elif t == 2:                                          # Landmark3DPrior
                l = int(f["lmk0"])
                r, J = landmark_prior_factor(B, self.P[l], f["delta"], f["sqrt_inf"], xl[l])
                cols = [(self.lmk_col[l], J)] if self.lmk_col[l] >= 0 else []
            elif t == 3:                                          # LandmarkToLandmarkFactor
                l0, l1 = int(f["lmk0"]), int(f["lmk1"])
                r, J = landmark_to_landmark_factor(B, self.P[l0], self.P[l1], f["delta"], f["sqrt_inf"], xl[l0], xl[l1])
                cols = [(c, Jb) for c, Jb in ((self.lmk_col[l0], J[:, :3]), (self.lmk_col[l1], J[:, 3:6])) if c >= 0]
            elif t == 4:                                          # Relative6DPose between key-frames a and b (the slots hold T_w_a, T_w_b)
                a, b = int(f["kf"]), int(f["kf_b"])
                r, Ja, Jb = relative_pose_factor(B, self.T[a], self.T[b], f["T_prior"], f["sqrt_inf"], xp[a], xp[b])
                cols = [(c, Jx) for c, Jx in ((self.kf_col[a], Ja), (self.kf_col[b], Jb)) if c >= 0]
            else:
                raise NotImplementedError(f"twin: sparse factor type {t} inside a solve")
            yield r, cols, bool(cols), sum(v * v for v in r)
        if self.dense is not None:                                # MarginalizationFactor: r = r0 + J dx (marginalization.hpp:113-215), VO layout
            d = self.dense
            J = B.a(np.asarray(d["J"], dtype=np.float64)); r0 = B.a(np.asarray(d["r0"], dtype=np.float64))
            dx = B.zeros(J.shape[1]); cols = []
            kk = int(d.get("kf_keep", -1))
            if kk >= 0:                                           # the kept frame's blocks: pose 6 | v 3 | ba 3 | bg 3 (marginalization.hpp:121-136, 157-197)
                fc = int(d["kf_col"])
                xv, xa, xg = self.split_vio(x)
                dx[fc: fc + 6] = xp[kk]; dx[fc + 6: fc + 9] = xv[kk]; dx[fc + 9: fc + 12] = xa[kk]; dx[fc + 12: fc + 15] = xg[kk]
                for c, o, wdt in ((self.kf_col[kk], 0, 6), (self.v_col[kk], 6, 3), (self.ba_col[kk], 9, 3), (self.bg_col[kk], 12, 3)):
                    if c >= 0:
                        cols.append((c, J[:, fc + o: fc + o + wdt]))
            for li, lc in zip(d["lmk_index"], d["lmk_col"]):
                if lc < 0:
                    continue
                dx[lc: lc + 3] = xl[int(li)]
                if self.lmk_col[int(li)] >= 0:
                    cols.append((self.lmk_col[int(li)], J[:, lc: lc + 3]))
            r = r0 + J @ dx
            yield r, cols, bool(cols), sum(v * v for v in r)

    def evaluate(self, x, want_j=True):
        """cost (reduced program), fixed cost, residual vector, dense Jacobian [m, n] (None unless want_j)."""
        B = self.B
        rs, rows, cost, fixed = [], [], B.s(0), B.s(0)
        for r, cols, inprog, rho in self.blocks(x, want_j):
            if not inprog:
                fixed = fixed + rho / 2
                continue
            cost = cost + rho / 2
            rs.append(r)
            rows.append(cols)
        m = sum(len(r) for r in rs)
        res = np.concatenate(rs) if rs else B.zeros(0)
        if not want_j:
            return cost, fixed, res, None
        J = B.zeros((m, self.n))
        i = 0
        for r, cols in zip(rs, rows):
            for (c, Jb) in cols:
                J[i: i + len(r), c: c + Jb.shape[1]] = Jb
            i += len(r)
        return cost, fixed, res, J


# ---------------------------------------------------------------------------------------------------------------
# dense SPD solve, generic scalar type
# ---------------------------------------------------------------------------------------------------------------
def cholesky_solve(B, A, b):
    """x = A^-1 b for SPD A by an (unblocked, row-vectorised) Cholesky factorisation; returns None if not PD."""
    if B.kind == "f64":
        try:
            L = np.linalg.cholesky(A)
        except np.linalg.LinAlgError:
            return None
        y = np.linalg.solve(L, b)          # LAPACK, independent of the oracle's hand-written Schur + Cholesky
        return np.linalg.solve(L.T, y)
    n = A.shape[0]
    L = A.copy()
    for j in range(n):
        d = L[j, j] - (L[j, :j] * L[j, :j]).sum() if j else L[j, j]
        if not d > 0:
            return None
        d = B.sqrt(d)
        L[j, j] = d
        if j + 1 < n:
            L[j + 1:, j] = (L[j + 1:, j] - (L[j + 1:, :j] @ L[j, :j] if j else 0)) / d
    y = b.copy()
    for i in range(n):
        y[i] = (y[i] - (L[i, :i] * y[:i]).sum() if i else y[i]) / L[i, i]
    x = y.copy()
    for i in range(n - 1, -1, -1):
        x[i] = (x[i] - (L[i + 1:, i] * x[i + 1:]).sum() if i + 1 < n else x[i]) / L[i, i]
    return x


def kept_landmark_columns(P):
    """Columns of the landmarks a dense prior couples with each other (they cannot be eliminated block by block)."""
    dp = getattr(P, "dense", None)
    if dp is None:
        return []
    cols = []
    for li, lc in zip(np.asarray(dp.get("lmk_index", []), dtype=int), np.asarray(dp.get("lmk_col", []), dtype=int)):
        if lc >= 0 and P.lmk_col[li] >= 0:
            cols.extend(range(P.lmk_col[li], P.lmk_col[li] + 3))
    return cols


def chol3(B, M):
    """Lower Cholesky factor of a 3 x 3 SPD block and its inverse (forward substitution), in the backend's arithmetic."""
    L = B.zeros((3, 3))
    L[0, 0] = B.sqrt(M[0, 0])
    L[1, 0] = M[1, 0] / L[0, 0]; L[2, 0] = M[2, 0] / L[0, 0]
    L[1, 1] = B.sqrt(M[1, 1] - L[1, 0] * L[1, 0])
    L[2, 1] = (M[2, 1] - L[2, 0] * L[1, 0]) / L[1, 1]
    L[2, 2] = B.sqrt(M[2, 2] - L[2, 0] * L[2, 0] - L[2, 1] * L[2, 1])
    Li = B.zeros((3, 3))
    Li[0, 0] = 1 / L[0, 0]; Li[1, 1] = 1 / L[1, 1]; Li[2, 2] = 1 / L[2, 2]
    Li[1, 0] = -L[1, 0] * Li[0, 0] / L[1, 1]
    Li[2, 0] = -(L[2, 0] * Li[0, 0] + L[2, 1] * Li[1, 0]) / L[2, 2]
    Li[2, 1] = -L[2, 1] * Li[1, 1] / L[2, 2]
    return L, Li


def schur_solve(B, P, H, g, D2, elim="adjugate"):
    """(H + diag(D2)) y = g by eliminating the landmark blocks no other landmark is coupled with — only used for the long-double
    arbitration runs on windows whose un-reduced system is too large for a long-double dense factorisation (2 943 unknowns: > 4 h)
    and for the elimination-numerics study (scripts/elim_numerics.py).
    The landmarks a dense prior holds are coupled with each other through it: they stay in the reduced system beside the poses (round
    4: eliminating them block by block as well solved a different system, 9e-4 off every other implementation). Exact-arithmetic
    equivalent of cholesky_solve. elim = "adjugate": M^-1 by the adjugate (what the device's sym3_inverse and the oracle do) and
    S -= (E M^-1) E^T; "cholesky": the block step of a landmark-first Cholesky of the un-reduced system (what a sparse direct
    solver does with a fill-reducing ordering): M = L L^T, W = E L^-T, S -= W W^T, every landmark solve through L."""
    npz = P.n_pose
    A = H + np.diag(D2)
    kept = kept_landmark_columns(P)
    kept_set = set(kept)
    R = list(range(npz)) + kept
    S = A[np.ix_(R, R)].copy()
    gR = g[R].copy()
    Minv = {}
    for l in range(P.n_lmk):
        c = P.lmk_col[l]
        if c < 0 or c in kept_set:
            continue
        E = A[:npz, c: c + 3]
        if elim == "cholesky":
            _, Li = chol3(B, A[c: c + 3, c: c + 3])
            W = E @ Li.T
            Minv[l] = Li
            S[:npz, :npz] -= W @ W.T
            gR[:npz] -= W @ (Li @ g[c: c + 3])
            continue
        Mi = inv3(B, A[c: c + 3, c: c + 3])
        Minv[l] = Mi
        Y = E @ Mi
        S[:npz, :npz] -= Y @ E.T
        gR[:npz] -= Y @ g[c: c + 3]
    yR = cholesky_solve(B, S, gR)
    if yR is None:
        return None
    y = B.zeros(P.n)
    y[R] = yR
    yp = yR[:npz]
    for l, Mi in Minv.items():
        c = P.lmk_col[l]
        if elim == "cholesky":
            y[c: c + 3] = Mi.T @ (Mi @ (g[c: c + 3] - A[:npz, c: c + 3].T @ yp))
        else:
            y[c: c + 3] = Mi @ (g[c: c + 3] - A[:npz, c: c + 3].T @ yp)
    return y


def normal_matrix(B, J):
    """J^T J. Long double has no BLAS: a dense product of the (rows x n) Jacobian is 38 minutes per LM iteration at n = 2 943, so the
    rows are taken by their sparsity — a reprojection row touches 9 columns (outer-product update), the rows of a dense prior form one
    dense block over the prior's columns."""
    if B.kind != "ld":
        return J.T @ J
    n = J.shape[1]
    H = np.zeros((n, n), dtype=J.dtype)
    nz = J != 0
    cnt = nz.sum(axis=1)
    dense_rows = np.flatnonzero(cnt > 64)
    if len(dense_rows):
        cols = np.flatnonzero(nz[dense_rows].any(axis=0))
        Jb = J[np.ix_(dense_rows, cols)]
        H[np.ix_(cols, cols)] += Jb.T @ Jb
    for i in np.flatnonzero((cnt > 0) & (cnt <= 64)):
        idx = np.flatnonzero(nz[i])
        v = J[i, idx]
        H[np.ix_(idx, idx)] += np.outer(v, v)
    return H


# ---------------------------------------------------------------------------------------------------------------
# Ceres 2.2.0 trust-region loop (Levenberg-Marquardt strategy), as published
# ---------------------------------------------------------------------------------------------------------------
DEFAULTS = dict(max_num_iterations=20, jacobi_scaling=1, max_num_consecutive_invalid_steps=5, function_tolerance=1e-3,
                gradient_tolerance=1e-10, parameter_tolerance=1e-8, initial_trust_region_radius=1e4,
                max_trust_region_radius=1e16, min_trust_region_radius=1e-32, min_lm_diagonal=1e-6, max_lm_diagonal=1e32,
                min_relative_decrease=1e-3, huber_a=0.0)
TERM = {"NO_CONVERGENCE": 0, "FUNCTION_TOL": 1, "PARAMETER_TOL": 2, "GRADIENT_TOL": 3, "MIN_RADIUS": 4, "FAILURE": 5}


def options_dict(opts=None):
    o = dict(DEFAULTS)
    if opts is not None:
        for k in o:
            o[k] = getattr(opts, k) if not isinstance(opts, dict) else opts.get(k, o[k])
    return o


class ViInitProblem:
    """AOptimizer::VIInit's problem (AOptimizer.cpp:448-581): gravity direction r_wi (2), the scale exponent lambda (1, constant
    unless optim_scale), the window's bias deltas dba / dbg (3 + 3; constant in the reference, `optim_bias` frees them as the C ABI
    allows), one velocity delta per frame (3); IMUFactorInit between consecutive key-frames, Landmark3DPrior(0, 0, I / sigma) on dba, dbg."""

    def __init__(self, B, T_f_w, vel, factors, optim_scale=False, optim_bias=False, sigma_dba=1.0, sigma_dbg=1.0):
        self.B = B
        self.T = np.asarray(T_f_w, dtype=np.float64).reshape(-1, 12)
        self.vel = np.asarray(vel, dtype=np.float64).reshape(-1, 3)
        self.factors = list(factors)
        self.n_f = self.T.shape[0]
        n = 2
        self.c_lam = -1
        if optim_scale:
            self.c_lam = n; n += 1
        self.c_ba = self.c_bg = -1
        if optim_bias:
            self.c_ba, self.c_bg = n, n + 3; n += 6
        used = sorted({int(f["kf_i"]) for f in self.factors} | {int(f["kf_j"]) for f in self.factors})
        self.c_v = np.full(self.n_f, -1)
        for k in used:                                             # a parameter block exists only if a residual block uses it
            self.c_v[k] = n; n += 3
        self.n = n
        self.sa, self.sg = 1.0 / sigma_dba, 1.0 / sigma_dbg
        self.has_imu = False

    def unpack(self, x):
        B = self.B
        lam = x[self.c_lam] if self.c_lam >= 0 else B.s(0)
        dba = x[self.c_ba: self.c_ba + 3] if self.c_ba >= 0 else B.zeros(3)
        dbg = x[self.c_bg: self.c_bg + 3] if self.c_bg >= 0 else B.zeros(3)
        dv = B.zeros((self.n_f, 3))
        for k in range(self.n_f):
            if self.c_v[k] >= 0:
                dv[k] = x[self.c_v[k]: self.c_v[k] + 3]
        return x[0:2], lam, dba, dbg, dv

    def evaluate(self, x, want_j=True):
        B = self.B
        r_wi, lam, dba, dbg, dv = self.unpack(x)
        rs, rows, cost, fixed = [], [], B.s(0), B.s(0)
        for f in self.factors:
            i, j = int(f["kf_i"]), int(f["kf_j"])
            r, Js = imu_factor_init(B, f, self.T[i], self.T[j], self.vel[i], self.vel[j], r_wi, dv[i], dv[j], dba, dbg, lam)
            cols = [(c, Jb) for c, Jb in zip((0, self.c_v[i], self.c_v[j], self.c_ba, self.c_bg, self.c_lam), Js) if c >= 0]
            cost = cost + sum(v * v for v in r) / 2
            rs.append(r); rows.append(cols)
        for c, sc, val in ((self.c_ba, self.sa, dba), (self.c_bg, self.sg, dbg)):    # Landmark3DPrior(0, 0, I / sigma)
            r = B.s(sc) * val
            if c >= 0:
                cost = cost + sum(v * v for v in r) / 2
                rs.append(r); rows.append([(c, B.s(sc) * B.eye(3))])
            else:
                fixed = fixed + sum(v * v for v in r) / 2
        m = sum(len(r) for r in rs)
        res = np.concatenate(rs) if rs else B.zeros(0)
        if not want_j:
            return cost, fixed, res, None
        J = B.zeros((m, self.n))
        i = 0
        for r, cols in zip(rs, rows):
            for (c, Jb) in cols:
                J[i: i + len(r), c: c + Jb.shape[1]] = Jb
            i += len(r)
        return cost, fixed, res, J


def lm_solve(w, opts=None, kind="f64", digits=50, use_schur=False, max_iterations=None, problem=None, elim="adjugate"):
    """Minimise the window's cost with Ceres' trust-region / LM schedule. Returns a dict: pose[n_kf,6], lmk[n_lmk,3]
    (float64), summary fields and `log` (one row per iteration: cost, cost_change, radius, step_norm, relative_decrease,
    successful, gradient_max, model_cost_change — the layout of the C oracle's log)."""
    o = options_dict(opts)
    if max_iterations is not None:
        o["max_num_iterations"] = max_iterations
    B = Backend(kind, digits) if problem is None else problem.B
    P = Problem(B, w, huber_a=o["huber_a"]) if problem is None else problem
    n = P.n
    x = B.zeros(n)
    x_norm = B.s(0)
    cost, fixed, r, J = P.evaluate(x)
    g = J.T @ r                                                   # gradient of the UNSCALED problem
    scale = B.zeros(n)
    for i in range(n):                                            # jacobian_scaling = 1 / (1 + ||col||), iteration 0 only
        scale[i] = 1 / (1 + B.sqrt((J[:, i] * J[:, i]).sum())) if o["jacobi_scaling"] else B.s(1)
    J = J * scale[None, :]
    gmax = max([abs(v) for v in g]) if n else B.s(0)
    radius = B.s(o["initial_trust_region_radius"])
    decrease_factor = B.s(2)
    reuse_diagonal = False
    diagonal = None
    log = [[B.f(cost), 0.0, B.f(radius), 0.0, 0.0, 1.0, B.f(gmax), 0.0]]
    out = dict(initial_cost=B.f(cost), fixed_cost=B.f(fixed), n_success=0, n_unsuccess=0, termination=TERM["NO_CONVERGENCE"])
    it = 0
    n_invalid = 0
    step_successful = True
    if gmax <= o["gradient_tolerance"]:
        out["termination"] = TERM["GRADIENT_TOL"]
    else:
        while True:
            # FinalizeIterationAndCheckIfMinimizerCanContinue
            if it >= o["max_num_iterations"]:
                out["termination"] = TERM["NO_CONVERGENCE"]
                break
            if step_successful and gmax <= o["gradient_tolerance"]:
                out["termination"] = TERM["GRADIENT_TOL"]
                break
            if radius < o["min_trust_region_radius"]:
                out["termination"] = TERM["MIN_RADIUS"]
                break
            it += 1
            # LevenbergMarquardtStrategy::ComputeStep
            if not reuse_diagonal:
                diagonal = B.zeros(n)
                for i in range(n):
                    d = (J[:, i] * J[:, i]).sum()
                    diagonal[i] = min(max(d, B.s(o["min_lm_diagonal"])), B.s(o["max_lm_diagonal"]))
            D2 = diagonal / radius                                # lm_diagonal = sqrt(diagonal / radius); D^2 enters the normal equations
            H = normal_matrix(B, J)
            rhs = J.T @ r
            y = schur_solve(B, P, H, rhs, D2, elim) if use_schur else cholesky_solve(B, H + np.diag(D2), rhs)
            valid = y is not None and all(B.isfinite(v) for v in y)
            mcc = B.s(0)
            if valid:
                step = -y
                Jd = J @ step
                mcc = -(Jd * (r + Jd / 2)).sum()                  # model_cost_change
                valid = mcc > 0
            if not valid:                                         # HandleInvalidStep
                n_invalid += 1
                step_successful = False
                log.append([B.f(cost), 0.0, B.f(radius), 0.0, 0.0, 0.0, B.f(gmax), B.f(mcc)])
                if n_invalid >= o["max_num_consecutive_invalid_steps"]:
                    out["termination"] = TERM["FAILURE"]
                    break
                radius = radius / 2                               # StepIsInvalid
                reuse_diagonal = True
                continue
            n_invalid = 0
            delta = step * scale
            cand = x + delta
            cand_cost, _, _, _ = P.evaluate(cand, want_j=False)
            step_norm = B.sqrt((delta * delta).sum())
            if step_norm <= o["parameter_tolerance"] * (x_norm + o["parameter_tolerance"]):
                out["termination"] = TERM["PARAMETER_TOL"]
                log.append([B.f(cand_cost), B.f(cost - cand_cost), B.f(radius), B.f(step_norm), 0.0, 0.0, B.f(gmax), B.f(mcc)])
                break
            cost_change = cost - cand_cost
            if abs(cost_change) <= o["function_tolerance"] * cost:
                out["termination"] = TERM["FUNCTION_TOL"]
                log.append([B.f(cand_cost), B.f(cost_change), B.f(radius), B.f(step_norm), B.f(cost_change / mcc), 0.0, B.f(gmax), B.f(mcc)])
                break
            rho = cost_change / mcc
            if rho > o["min_relative_decrease"]:                  # HandleSuccessfulStep
                x = cand
                x_norm = B.sqrt((x * x).sum())
                cost, _, r, J = P.evaluate(x)
                g = J.T @ r
                J = J * scale[None, :]
                gmax = max(abs(v) for v in g)
                radius = radius / max(B.s(1) / 3, 1 - (2 * rho - 1) ** 3)   # StepAccepted
                radius = min(B.s(o["max_trust_region_radius"]), radius)
                decrease_factor = B.s(2)
                reuse_diagonal = False
                step_successful = True
                out["n_success"] += 1
                log.append([B.f(cost), B.f(cost_change), B.f(radius), B.f(step_norm), B.f(rho), 1.0, B.f(gmax), B.f(mcc)])
            else:                                                 # HandleUnsuccessfulStep / StepRejected
                radius = radius / decrease_factor
                decrease_factor = decrease_factor * 2
                reuse_diagonal = True
                step_successful = False
                out["n_unsuccess"] += 1
                log.append([B.f(cost), B.f(cost_change), B.f(radius), B.f(step_norm), B.f(rho), 0.0, B.f(gmax), B.f(mcc)])
    if problem is not None:                                       # a problem of its own (VIInit): raw solution, unpacked by the caller
        out.update(iterations=it, final_cost=B.f(cost), final_radius=B.f(radius), log=np.array(log), x_scalar=x, backend=B, problem=P)
        return out
    xp, xl = P.split(x)
    if P.n_line:
        out.update(line=B.f(P.split_lines(x)))
    if P.has_imu:
        xv, xa, xg = P.split_vio(x)
        out.update(dv=B.f(xv), dba=B.f(xa), dbg=B.f(xg))
    out.update(pose=B.f(xp), lmk=B.f(xl), iterations=it, final_cost=B.f(cost), final_radius=B.f(radius), log=np.array(log),
               x_scalar=x, backend=B, problem=P)
    return out


def first_iteration(w, opts=None, kind="f64", digits=50):
    """ONE full LM iteration at x = 0 on the un-reduced normal equations: the step (pose / landmark parts, float64), the
    model cost change, the candidate cost and the step quality."""
    res = lm_solve(w, opts, kind=kind, digits=digits, max_iterations=1)
    return res
