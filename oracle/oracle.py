"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY -- parity unpinned (see oracle/dpgo_oracle.h).  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by dpgo_ros_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Meas(C.Structure):
    _fields_ = [("r1", C.c_int), ("p1", C.c_int), ("r2", C.c_int), ("p2", C.c_int),
                ("R", C.c_double * 9), ("t", C.c_double * 3),
                ("kappa", C.c_double), ("tau", C.c_double), ("weight", C.c_double),
                ("fixed_weight", C.c_int), ("is_known_inlier", C.c_int)]


MEAS_DTYPE = np.dtype([("r1", "<i4"), ("p1", "<i4"), ("r2", "<i4"), ("p2", "<i4"),
                       ("R", "<f8", (9,)), ("t", "<f8", (3,)),
                       ("kappa", "<f8"), ("tau", "<f8"), ("weight", "<f8"),
                       ("fixed_weight", "<i4"), ("is_known_inlier", "<i4")], align=True)
assert MEAS_DTYPE.itemsize == C.sizeof(Meas)


class Params(C.Structure):
    _fields_ = [("d", C.c_int), ("r", C.c_int), ("num_robots", C.c_int), ("method", C.c_int),
                ("rgd_stepsize", C.c_double), ("rgd_use_preconditioner", C.c_int),
                ("rtr_iterations", C.c_int), ("rtr_tcg_iterations", C.c_int),
                ("gradnorm_tol", C.c_double), ("rtr_initial_radius", C.c_double),
                ("rtr_max_radius", C.c_double), ("precond_shift", C.c_double),
                ("acceleration", C.c_int), ("restart_interval", C.c_int),
                ("rel_change_tol", C.c_double), ("max_num_iters", C.c_int),
                ("robust_cost_type", C.c_int), ("gnc_barc", C.c_double),
                ("gnc_mu_step", C.c_double), ("gnc_init_mu", C.c_double),
                ("robust_opt_num_weight_updates", C.c_int), ("robust_opt_inner_iters", C.c_int),
                ("robust_opt_min_convergence_ratio", C.c_double), ("weights_as_float32", C.c_int),
                ("robust_opt_num_resets", C.c_int), ("precond_mode", C.c_int), ("status_every_iterate", C.c_int),
                ("rgd_line_search", C.c_int), ("rgd_ls_max_backoffs", C.c_int), ("rgd_ls_shrink", C.c_double),
                ("rgd_ls_sigma", C.c_double), ("tls_threshold", C.c_double), ("huber_threshold", C.c_double)]


class OptResult(C.Structure):
    _fields_ = [("success", C.c_int), ("f_init", C.c_double), ("f_opt", C.c_double),
                ("gradnorm_init", C.c_double), ("gradnorm_opt", C.c_double),
                ("rtr_outer_iters", C.c_int), ("tcg_iters_total", C.c_int),
                ("hessvec_count", C.c_int), ("precond_count", C.c_int), ("accepted", C.c_int),
                ("ls_backoffs", C.c_int)]


class Status(C.Structure):
    _fields_ = [("agent_id", C.c_int), ("state", C.c_int), ("instance_number", C.c_int),
                ("iteration_number", C.c_int), ("ready_to_terminate", C.c_int),
                ("relative_change", C.c_double)]


METHOD_RTR, METHOD_RGD = 0, 1
COST_L2, COST_L1, COST_HUBER, COST_TLS, COST_GM, COST_GNC_TLS = 0, 1, 2, 3, 4, 5
WEIGHT_LIBRARY, WEIGHT_WRAPPER = 0, 1


# flags of the library in use (bench.py prints them next to the CPU baseline)
BUILD = {"compiler": "gcc", "flags": "-O3 -march=x86-64-v2 -std=gnu99 -fno-fast-math (oracle/Makefile)"}


def build_native():
    """the same sources built FOR THE HOST THAT RUNS THEM (-O3 -march=native, the reference's own flags:
    /root/reference/CMakeLists.txt:9), outside the tree: the timed CPU legs of bench.py use this one (DPGO_ORACLE_NATIVE=1).
    The in-tree liboracle.so stays portable (x86-64-v2): it is built in one container and run on another host."""
    import tempfile
    out_dir = os.path.join(tempfile.gettempdir(), "dpgo_oracle_native_%d" % os.getuid())
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "liboracle.so")
    flags = ["-O3", "-march=native", "-std=gnu99", "-fPIC", "-fno-fast-math"]
    srcs = [os.path.join(_HERE, f) for f in ("orc_io.c", "orc_core.c", "orc_agent.c", "orc_align.c")]
    deps = srcs + [os.path.join(_HERE, f) for f in ("dpgo_oracle.h", "orc_internal.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        # (several processes may arrive here at once -- the ranks of a multi-process test: each builds its own file, the
        # rename is atomic)
        tmp = "%s.%d.tmp" % (so, os.getpid())
        subprocess.check_call(["gcc"] + flags + ["-shared", "-o", tmp] + srcs + ["-lm"])
        os.replace(tmp, so)
    try:
        ver = subprocess.check_output(["gcc", "--version"], text=True).splitlines()[0]
    except Exception:
        ver = "gcc"
    BUILD.update({"compiler": ver, "flags": " ".join(flags)})
    return so


def build(force=False):
    if os.environ.get("DPGO_ORACLE_NATIVE") == "1":
        return build_native()
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_io.c", "orc_core.c", "orc_agent.c",
                                             "dpgo_oracle.h", "orc_internal.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
        L.orc_agent_new.restype = vp
        L.orc_team_new.restype = vp
        L.orc_team_agent.restype = vp
        for f in ("orc_agent_eval", "orc_team_cost", "orc_robust_weight", "orc_error_threshold_at_quantile",
                  "orc_measurement_cost"):
            getattr(L, f).restype = C.c_double
    return _LIB


def _d(a):
    return a.ctypes.data_as(C.c_void_p)


def default_params(r=5, num_robots=1, **kw):
    p = Params()
    lib().orc_default_params(C.byref(p), r, num_robots)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


def read_g2o(path, weight_mode=WEIGHT_LIBRARY):
    out = C.c_void_p()
    n = C.c_int()
    nm = lib().orc_read_g2o(path.encode(), weight_mode, C.byref(out), C.byref(n))
    if nm < 0:
        raise FileNotFoundError(path)
    arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(nm * MEAS_DTYPE.itemsize,))
    m = np.frombuffer(arr.tobytes(), dtype=MEAS_DTYPE).copy()
    lib().orc_free(out)
    return m, n.value


def read_csv(path, weight_mode=WEIGHT_LIBRARY):
    out = C.c_void_p()
    nm = lib().orc_read_measurements_csv(path.encode(), weight_mode, C.byref(out))
    if nm < 0:
        raise FileNotFoundError(path)
    arr = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)), shape=(max(nm, 1) * MEAS_DTYPE.itemsize,))
    m = np.frombuffer(arr.tobytes(), dtype=MEAS_DTYPE)[:nm].copy()
    lib().orc_free(out)
    return m


def partition(m, num_poses, num_robots, weight_mode=WEIGHT_LIBRARY):
    m = m.copy()
    lib().orc_partition(_d(m), len(m), num_poses, num_robots, weight_mode)
    return m


def odometry_init(m, num_poses):
    T = np.zeros(12 * num_poses)
    lib().orc_odometry_init(_d(m), len(m), num_poses, _d(T))
    return T


def chordal_init(m, num_poses):
    T = np.zeros(12 * num_poses)
    rc = lib().orc_chordal_init(_d(m), len(m), num_poses, _d(T))
    if rc != 0:
        raise RuntimeError("chordal init failed")
    return T


def robust_frame_alignment(Tc, max_rotation_error_rad=0.5, max_translation_error=1.0, min_inliers=2):
    Tc = np.ascontiguousarray(Tc, dtype=np.float64).reshape(-1, 12)
    T, inl = np.zeros(12), np.zeros(len(Tc), dtype=np.int32)
    rc = lib().orc_robust_frame_alignment(_d(Tc), len(Tc), C.c_double(max_rotation_error_rad),
                                          C.c_double(max_translation_error), min_inliers, _d(T), _d(inl))
    return (T, inl.astype(bool)) if rc == 0 else None


def fixed_stiefel(r):
    Y = np.zeros(3 * r)
    lib().orc_fixed_stiefel(r, _d(Y))
    return Y


def lift(T, num_poses, YLift, r):
    X = np.zeros(r * 4 * num_poses)
    lib().orc_lift(_d(T), num_poses, _d(YLift), r, _d(X))
    return X


def measurement_cost(m, X, r):
    return lib().orc_measurement_cost(_d(m), len(m), _d(X), r)


def project_stiefel(A, r):
    out = np.zeros(3 * r)
    lib().orc_project_stiefel(_d(np.ascontiguousarray(A, dtype=np.float64)), r, _d(out))
    return out


def project_manifold(X, r, n):
    out = np.zeros_like(X)
    lib().orc_project_manifold(_d(X), r, n, _d(out))
    return out


def tangent_project(X, V, r, n):
    out = np.zeros_like(X)
    lib().orc_tangent_project(_d(X), _d(V), r, n, _d(out))
    return out


def retract(X, eta, r, n):
    out = np.zeros_like(X)
    lib().orc_retract(_d(X), _d(eta), r, n, _d(out))
    return out


class Agent:
    """Mirror of the DPGO::PGOAgent call surface used by PGOAgentROS."""

    def __init__(self, agent_id, params, handle=None):
        self.params = params
        self.id = agent_id
        self.r = params.r
        self._own = handle is None
        self.h = C.c_void_p(lib().orc_agent_new(agent_id, C.byref(params))) if handle is None else C.c_void_p(handle)

    def __del__(self):
        if getattr(self, "_own", False) and self.h:
            lib().orc_agent_free(self.h)
            self.h = None

    def add_measurements(self, m):
        m = np.ascontiguousarray(m)
        for k in range(len(m)):
            lib().orc_agent_add_measurement(self.h, C.c_void_p(m.ctypes.data + k * MEAS_DTYPE.itemsize))

    @property
    def n(self):
        return lib().orc_agent_num_poses(self.h)

    def _vec(self):
        return np.zeros(self.r * 4 * self.n)

    def neighbors(self):
        c = lib().orc_agent_num_neighbors(self.h, None)
        ids = np.zeros(max(c, 1), dtype=np.int32)
        lib().orc_agent_num_neighbors(self.h, _d(ids))
        return ids[:c].tolist()

    def public_pose_ids(self, nbr):
        c = lib().orc_agent_public_pose_ids(self.h, nbr, None)
        f = np.zeros(max(c, 1), dtype=np.int32)
        lib().orc_agent_public_pose_ids(self.h, nbr, _d(f))
        return f[:c]

    def neighbor_pose_ids(self, nbr):
        c = lib().orc_agent_neighbor_pose_ids(self.h, nbr, None)
        f = np.zeros(max(c, 1), dtype=np.int32)
        lib().orc_agent_neighbor_pose_ids(self.h, nbr, _d(f))
        return f[:c]

    def set_X(self, X):
        X = np.ascontiguousarray(X, dtype=np.float64)
        assert X.size == self.r * 4 * self.n
        lib().orc_agent_set_X(self.h, _d(X))

    def get_X(self):
        X = self._vec(); lib().orc_agent_get_X(self.h, _d(X)); return X

    def get_Y(self):
        X = self._vec(); lib().orc_agent_get_Y(self.h, _d(X)); return X

    def get_V(self):
        X = self._vec(); lib().orc_agent_get_V(self.h, _d(X)); return X

    def get_public_poses(self, nbr, aux=False):
        ids = self.public_pose_ids(nbr)
        out = np.zeros(max(len(ids), 1) * 4 * self.r)
        lib().orc_agent_get_public_poses(self.h, nbr, int(aux), _d(out))
        return ids, out[:len(ids) * 4 * self.r]

    def update_neighbor_poses(self, nbr, frames, poses, aux=False):
        frames = np.ascontiguousarray(frames, dtype=np.int32)
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        lib().orc_agent_update_neighbor_poses(self.h, nbr, int(aux), len(frames), _d(frames), _d(poses))

    def iterate(self, do_opt=True):
        return bool(lib().orc_agent_iterate(self.h, int(do_opt)))

    def status(self):
        s = Status(); lib().orc_agent_get_status(self.h, C.byref(s)); return s

    def opt_result(self):
        s = OptResult(); lib().orc_agent_get_opt_result(self.h, C.byref(s)); return s

    def build_problem(self, aux=False):
        lib().orc_agent_build_problem(self.h, int(aux))

    def eval(self, X):
        eg, rg = self._vec(), self._vec()
        f = lib().orc_agent_eval(self.h, _d(np.ascontiguousarray(X)), _d(eg), _d(rg))
        return f, eg, rg

    def hessvec(self, X, eta):
        out = self._vec()
        lib().orc_agent_hessvec(self.h, _d(np.ascontiguousarray(X)), _d(np.ascontiguousarray(eta)), _d(out))
        return out

    def precondition(self, X, V):
        out = self._vec()
        lib().orc_agent_precondition(self.h, _d(np.ascontiguousarray(X)), _d(np.ascontiguousarray(V)), _d(out))
        return out

    def get_Q(self):
        nb = lib().orc_agent_get_Q(self.h, None, None, None)
        rowptr = np.zeros(self.n + 1, dtype=np.int32); col = np.zeros(nb, dtype=np.int32); val = np.zeros(16 * nb)
        lib().orc_agent_get_Q(self.h, _d(rowptr), _d(col), _d(val))
        return rowptr, col, val

    def get_G(self):
        G = self._vec(); lib().orc_agent_get_G(self.h, _d(G)); return G

    def measurements(self):
        c = lib().orc_agent_get_measurements(self.h, None)
        m = np.zeros(max(c, 1), dtype=MEAS_DTYPE)
        lib().orc_agent_get_measurements(self.h, _d(m))
        return m[:c]

    def compute_residual(self, meas_row):
        m = np.ascontiguousarray(np.array([meas_row], dtype=MEAS_DTYPE))
        res = C.c_double()
        ok = lib().orc_agent_compute_residual(self.h, _d(m), C.byref(res))
        return bool(ok), res.value

    def robust_weight(self, residual):
        return lib().orc_robust_weight(self.h, C.c_double(residual))

    def should_update_weights(self):
        return bool(lib().orc_agent_should_update_weights(self.h))

    def update_measurement_weights(self):
        lib().orc_agent_update_measurement_weights(self.h)

    def set_measurement_weight(self, r1, p1, r2, p2, w, fixed=False):
        return bool(lib().orc_agent_set_measurement_weight(self.h, r1, p1, r2, p2, C.c_double(w), int(fixed)))

    def clear_data_matrices(self):
        lib().orc_agent_clear_data_matrices(self.h)


class Team:
    """Synchronous schedule driver (PGOAgentROS.cpp:129-220, 443-504, 1161-1189)."""

    def __init__(self, meas, num_poses, params):
        self.params = params
        self.r = params.r
        self.N = params.num_robots
        self.num_poses = num_poses
        meas = np.ascontiguousarray(meas)
        self.h = C.c_void_p(lib().orc_team_new(_d(meas), len(meas), num_poses, C.byref(params), 0))
        self.agents = [Agent(k, params, handle=lib().orc_team_agent(self.h, k)) for k in range(self.N)]

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_team_free(self.h)
            self.h = None

    def set_schedule(self, order):
        o = np.ascontiguousarray(order, dtype=np.int32)
        lib().orc_team_set_schedule(self.h, _d(o), len(o))

    def set_initial(self, T, YLift):
        lib().orc_team_set_initial(self.h, _d(np.ascontiguousarray(T)), _d(np.ascontiguousarray(YLift)))

    def iterate(self):
        return lib().orc_team_iterate(self.h)

    def exchange_all(self):
        lib().orc_team_exchange_all(self.h)

    def should_terminate(self):
        """PGOAgent::shouldTerminate() as the leader evaluates it (PGOAgentROS.cpp:208)."""
        return bool(lib().orc_team_should_terminate(self.h))

    def run_schedule(self, max_iters):
        """The synchronous schedule with the leader's TERMINATE / UPDATE_WEIGHT decisions (PGOAgentROS.cpp:206-214).
        Returns (iterations executed, terminated, weight-update rounds)."""
        term, rounds = C.c_int(0), C.c_int(0)
        done = lib().orc_team_run_schedule(self.h, int(max_iters), C.byref(term), C.byref(rounds))
        return done, bool(term.value), rounds.value

    def cost(self):
        return lib().orc_team_cost(self.h)

    def global_X(self):
        X = np.zeros(self.r * 4 * self.num_poses)
        lib().orc_team_get_global_X(self.h, _d(X))
        return X

    def update_weights(self):
        return lib().orc_team_update_weights(self.h)
