"""ctypes mirrors of the structs of include/sadvio_ba.h — the ORACLE's own copy (test infrastructure only).

Written from the C header, not shared with the product's binding (sadvio_amd/capi.py): a field-order slip in one of the two
mirrors shows up as a GPU-vs-oracle disagreement instead of cancelling out, and tests/test_struct_layout.py checks both
against the C compiler's sizeof / offsetof. Converters take duck-typed Python objects (attribute names = header names) and
copy BY NAME."""
import ctypes as C

import numpy as np

f64, i32, i64, u8 = C.c_double, C.c_int32, C.c_int64, C.c_uint8
P = C.POINTER


class flat_window(C.Structure):            # sadvio_flat_window
    _fields_ = [("n_kf", i32), ("n_cam", i32), ("n_lmk", i32), ("n_obs", i32), ("factor_type", i32), ("has_imu", i32),
                ("kf_id", P(i64)), ("kf_T_f_w", P(f64)), ("kf_const", P(u8)), ("kf_vel", P(f64)), ("kf_ba", P(f64)),
                ("kf_bg", P(f64)), ("cam_K", P(f64)), ("cam_T_s_f", P(f64)), ("cam_sigma", P(f64)),
                ("lmk_id", P(i64)), ("lmk_p", P(f64)), ("lmk_const", P(u8)), ("lmk_obs_ptr", P(i32)),
                ("obs_kf", P(i32)), ("obs_cam", P(i32)), ("obs_meas", P(f64))]


class imu_factor(C.Structure):             # sadvio_imu_factor
    _fields_ = [("kf_i", i32), ("kf_j", i32), ("dt", f64), ("delta_R", f64 * 9), ("delta_v", f64 * 3), ("delta_p", f64 * 3),
                ("J_dR_bg", f64 * 9), ("J_dv_ba", f64 * 9), ("J_dv_bg", f64 * 9), ("J_dp_ba", f64 * 9), ("J_dp_bg", f64 * 9),
                ("cov", f64 * 81), ("bacc_noise", f64), ("bgyr_noise", f64)]


class pose_prior(C.Structure):             # sadvio_pose_prior
    _fields_ = [("kf", i32), ("pad", i32), ("T_prior", f64 * 12), ("inf_diag", f64 * 6)]


class sparse_prior(C.Structure):           # sadvio_sparse_prior
    _fields_ = [("type", i32), ("kf", i32), ("lmk0", i32), ("lmk1", i32), ("T_prior", f64 * 12), ("v_prior", f64 * 3),
                ("ba_prior", f64 * 3), ("bg_prior", f64 * 3), ("delta", f64 * 3), ("sqrt_inf", f64 * 225), ("kf_b", i32), ("pad", i32)]


class line_set(C.Structure):               # sadvio_line_set
    _fields_ = [("n_line", i32), ("n_obs", i32), ("line_id", P(i64)), ("line_T_w_l", P(f64)), ("line_model", P(f64)),
                ("line_const", P(u8)), ("line_obs_ptr", P(i32)), ("obs_kf", P(i32)), ("obs_cam", P(i32)), ("obs_meas", P(f64))]


def lines_to_c(lines):
    """(line_set, keep-alive) from a dict with keys T_w_l [n,12], model [n,6], obs_ptr [n+1], obs_kf, obs_cam, obs_meas
    (+ optional id, const)."""
    T = np.ascontiguousarray(lines["T_w_l"], dtype=np.float64).reshape(-1, 12)
    n = T.shape[0]
    arrs = dict(line_id=np.ascontiguousarray(lines.get("id", np.arange(n)), dtype=np.int64), line_T_w_l=T,
                line_model=np.ascontiguousarray(lines["model"], dtype=np.float64).reshape(n, 6),
                line_obs_ptr=np.ascontiguousarray(lines["obs_ptr"], dtype=np.int32), obs_kf=np.ascontiguousarray(lines["obs_kf"], dtype=np.int32),
                obs_cam=np.ascontiguousarray(lines["obs_cam"], dtype=np.int32), obs_meas=np.ascontiguousarray(lines["obs_meas"], dtype=np.float64))
    c = line_set()
    c.n_line, c.n_obs = n, int(arrs["obs_kf"].size)
    ct = dict(line_id=i64, line_T_w_l=f64, line_model=f64, line_obs_ptr=i32, obs_kf=i32, obs_cam=i32, obs_meas=f64)
    for k, a in arrs.items():
        setattr(c, k, a.ctypes.data_as(P(ct[k])))
    keep = list(arrs.values())
    if lines.get("const") is not None:
        lc = np.ascontiguousarray(lines["const"], dtype=np.uint8)
        c.line_const = lc.ctypes.data_as(P(u8))
        keep.append(lc)
    return c, keep


class solve_options(C.Structure):          # sadvio_solve_options
    _fields_ = [("max_num_iterations", i32), ("jacobi_scaling", i32), ("max_num_consecutive_invalid_steps", i32),
                ("reserved", i32), ("function_tolerance", f64), ("gradient_tolerance", f64), ("parameter_tolerance", f64),
                ("initial_trust_region_radius", f64), ("max_trust_region_radius", f64), ("min_trust_region_radius", f64),
                ("min_lm_diagonal", f64), ("max_lm_diagonal", f64), ("min_relative_decrease", f64), ("huber_a", f64),
                ("max_solver_time_in_seconds", f64)]


class solve_summary(C.Structure):          # sadvio_solve_summary
    _fields_ = [("iterations", i32), ("num_successful_steps", i32), ("num_unsuccessful_steps", i32), ("termination", i32),
                ("initial_cost", f64), ("final_cost", f64), ("fixed_cost", f64), ("final_radius", f64)]


class viinit_problem(C.Structure):         # sadvio_viinit_problem
    _fields_ = [("n_frames", i32), ("n_factors", i32), ("T_f_w", P(f64)), ("vel", P(f64)), ("factors", P(imu_factor)),
                ("optim_scale", i32), ("optim_bias", i32), ("sigma_dba", f64), ("sigma_dbg", f64)]


class viinit_result(C.Structure):          # sadvio_viinit_result
    _fields_ = [("r_wi", f64 * 2), ("lambda_", f64), ("dba", f64 * 3), ("dbg", f64 * 3), ("R_w_i", f64 * 9), ("scale", f64)]


# Ceres Solver 2.2.0 defaults + the reference's hard-coded options (AOptimizer.cpp:315-323)
def reference_options():
    o = solve_options()
    o.max_num_iterations, o.jacobi_scaling, o.max_num_consecutive_invalid_steps = 20, 1, 5
    o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance = 1e-3, 1e-10, 1e-8
    o.initial_trust_region_radius, o.max_trust_region_radius, o.min_trust_region_radius = 1e4, 1e16, 1e-32
    o.min_lm_diagonal, o.max_lm_diagonal, o.min_relative_decrease = 1e-6, 1e32, 1e-3
    return o


def options_from(obj):
    """Copy any object with the option attributes (e.g. the product binding's struct) BY NAME."""
    if obj is None:
        return reference_options()
    o = solve_options()
    for name, _ in solve_options._fields_:
        setattr(o, name, getattr(obj, name))
    return o


_ARRAYS = [("kf_id", np.int64, i64), ("kf_T_f_w", np.float64, f64), ("kf_const", np.uint8, u8), ("kf_vel", np.float64, f64),
           ("kf_ba", np.float64, f64), ("kf_bg", np.float64, f64), ("cam_K", np.float64, f64), ("cam_T_s_f", np.float64, f64),
           ("cam_sigma", np.float64, f64), ("lmk_id", np.int64, i64), ("lmk_p", np.float64, f64), ("lmk_const", np.uint8, u8),
           ("lmk_obs_ptr", np.int32, i32), ("obs_kf", np.int32, i32), ("obs_cam", np.int32, i32), ("obs_meas", np.float64, f64)]


def window_to_c(w):
    """(flat_window, keep-alive list) from a window object with the header's field names as attributes."""
    n_kf, n_lmk = int(np.asarray(w.kf_T_f_w).reshape(-1, 12).shape[0]), int(np.asarray(w.lmk_p).reshape(-1, 3).shape[0])
    c = flat_window()
    c.n_kf, c.n_cam, c.n_lmk, c.n_obs = n_kf, int(np.asarray(w.cam_K).reshape(-1, 4).shape[0]), n_lmk, int(np.asarray(w.obs_kf).size)
    c.factor_type, c.has_imu = int(w.factor_type), int(getattr(w, "has_imu", 0))
    keep = []
    for name, dt, ct in _ARRAYS:
        a = getattr(w, name, None)
        if a is None and name == "kf_id":
            a = np.arange(n_kf)
        if a is None and name == "lmk_id":
            a = np.arange(n_lmk)
        if a is None:
            setattr(c, name, P(ct)())
            continue
        arr = np.ascontiguousarray(a, dtype=dt)
        keep.append(arr)
        setattr(c, name, arr.ctypes.data_as(P(ct)))
    return c, keep


def fill_imu(dst, f):
    dst.kf_i, dst.kf_j, dst.dt = int(f["kf_i"]), int(f["kf_j"]), float(f["dt"])
    for k in ("delta_R", "delta_v", "delta_p", "J_dR_bg", "J_dv_ba", "J_dv_bg", "J_dp_ba", "J_dp_bg", "cov"):
        getattr(dst, k)[:] = list(np.asarray(f[k], dtype=np.float64).ravel())
    dst.bacc_noise, dst.bgyr_noise = float(f["bacc_noise"]), float(f["bgyr_noise"])


def imus_to_c(factors):
    arr = (imu_factor * max(1, len(factors)))()
    for i, f in enumerate(factors):
        fill_imu(arr[i], f)
    return arr, len(factors)


def priors_to_c(priors):
    arr = (pose_prior * max(1, len(priors)))()
    for i, (kf, T, inf) in enumerate(priors):
        arr[i].kf = int(kf)
        arr[i].T_prior[:] = list(np.asarray(T, dtype=np.float64).ravel())
        arr[i].inf_diag[:] = list(np.asarray(inf, dtype=np.float64).ravel())
    return arr, len(priors)


def sparse_to_c(factors):
    arr = (sparse_prior * max(1, len(factors)))()
    for i, f in enumerate(factors):
        a = arr[i]
        a.type, a.kf, a.lmk0, a.lmk1 = int(f["type"]), int(f.get("kf", -1)), int(f.get("lmk0", -1)), int(f.get("lmk1", -1))
        a.kf_b = int(f.get("kf_b", -1))
        for k, n in (("T_prior", 12), ("v_prior", 3), ("ba_prior", 3), ("bg_prior", 3), ("delta", 3)):
            v = np.zeros(n) if f.get(k) is None else np.asarray(f[k], dtype=np.float64).ravel()
            getattr(a, k)[:] = list(v)
        W = np.asarray(f["sqrt_inf"], dtype=np.float64).ravel()
        buf = np.zeros(225)
        buf[: len(W)] = W
        a.sqrt_inf[:] = list(buf)
    return arr, len(factors)
