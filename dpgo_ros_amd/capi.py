"""ctypes binding of libdpgo_hip.so -- the C-ABI declared in include/dpgo_hip.h.

Python-side mirror of the reference interface for this path: the `Agent` class carries the
DPGO::PGOAgent method names the ROS wrapper calls (src/PGOAgentROS.cpp), `Team`
the synchronous schedule of src/PGOAgentROS.cpp:129-220.  There is no CPU fallback: loading fails
loudly when the HIP extension has not been built, and every compute call needs a GPU.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DPGO_HIP_LIB: an instrumented build of the same library, profiles/experiments/build_variant.sh)
LIB_PATH = os.environ.get("DPGO_HIP_LIB") or os.path.join(_HERE, "libdpgo_hip.so")
_LIB = None


class Measurement(C.Structure):
    _fields_ = [("r1", C.c_int), ("p1", C.c_int), ("r2", C.c_int), ("p2", C.c_int),
                ("R", C.c_double * 9), ("t", C.c_double * 3),
                ("kappa", C.c_double), ("tau", C.c_double), ("weight", C.c_double),
                ("fixed_weight", C.c_int), ("is_known_inlier", C.c_int)]


MEAS_DTYPE = np.dtype([("r1", "<i4"), ("p1", "<i4"), ("r2", "<i4"), ("p2", "<i4"),
                       ("R", "<f8", (9,)), ("t", "<f8", (3,)),
                       ("kappa", "<f8"), ("tau", "<f8"), ("weight", "<f8"),
                       ("fixed_weight", "<i4"), ("is_known_inlier", "<i4")], align=True)
assert MEAS_DTYPE.itemsize == C.sizeof(Measurement)


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
OK, NOT_READY, ERR = 0, 1, -1
PRECOND_AUTO, PRECOND_DENSE, PRECOND_BLOCK_JACOBI, PRECOND_TWO_LEVEL = 0, 1, 2, 3

# every symbol include/dpgo_hip.h declares (checked by tests/test_abi.py)
EXPORTS = """dpgo_default_params dpgo_last_error dpgo_read_g2o dpgo_read_measurements_csv dpgo_partition
dpgo_free dpgo_odometry_init dpgo_chordal_init dpgo_fixed_stiefel dpgo_lift dpgo_team_create dpgo_team_destroy
dpgo_team_num_local dpgo_team_stream dpgo_team_synchronize dpgo_agent_add_measurements
dpgo_agent_num_poses dpgo_agent_num_measurements dpgo_agent_get_neighbors dpgo_agent_public_pose_ids
dpgo_agent_neighbor_pose_ids dpgo_agent_set_X dpgo_agent_get_X dpgo_agent_get_public_poses
dpgo_agent_update_neighbor_poses dpgo_agent_pack_public_poses_device
dpgo_agent_unpack_neighbor_poses_device dpgo_agent_iterate dpgo_agent_get_status
dpgo_agent_get_opt_result dpgo_agent_iteration_number dpgo_agent_publish_requested dpgo_agent_set_iteration_number
dpgo_agent_build_problem dpgo_agent_eval dpgo_agent_hessvec dpgo_agent_precondition dpgo_agent_get_Q
dpgo_agent_get_G dpgo_project_manifold dpgo_tangent_project dpgo_retract dpgo_agent_compute_residual
dpgo_agent_robust_weight dpgo_agent_update_measurement_weights dpgo_agent_set_measurement_weight
dpgo_agent_get_measurements dpgo_agent_should_update_weights dpgo_agent_clear_data_matrices
dpgo_error_threshold_at_quantile dpgo_team_set_schedule dpgo_team_set_initial dpgo_team_exchange_all
dpgo_agent_pull_local dpgo_team_time_kernel dpgo_team_run dpgo_team_get_coloring dpgo_team_run_colored dpgo_team_set_groups dpgo_team_run_group dpgo_team_step_begin dpgo_team_step_end dpgo_team_iteration dpgo_team_cost dpgo_team_update_weights dpgo_team_get_counters
dpgo_write_measurements_csv dpgo_write_g2o dpgo_write_trajectory_csv dpgo_robust_frame_alignment dpgo_robust_local_init dpgo_team_run_simultaneous
dpgo_team_should_terminate dpgo_team_run_schedule dpgo_agent_compute_residuals dpgo_agent_set_measurement_weights
dpgo_agent_reset_acceleration dpgo_team_prepare dpgo_agent_read_partials dpgo_agent_preconditioner dpgo_agent_preconditioner_info dpgo_agent_preconditioner_residual dpgo_two_level_plan dpgo_agent_export_state dpgo_team_import_peer dpgo_team_export_mailbox dpgo_team_import_mailbox dpgo_team_run_peer dpgo_agent_read_rtr_handoff
dpgo_comm_unique_id dpgo_comm_create dpgo_comm_destroy dpgo_comm_rank dpgo_comm_world dpgo_comm_library
dpgo_comm_allreduce_sum dpgo_comm_allreduce_max dpgo_team_attach_comm dpgo_team_detach_comm dpgo_team_exchange_all_ranks
dpgo_team_run_ranks dpgo_comm_global_cost dpgo_team_comm_counters dpgo_team_set_iteration_log dpgo_team_run_simultaneous_ranks
dpgo_team_run_group_ranks dpgo_rank_plan_simulate dpgo_team_set_uniform_schedule""".split()


class DpgoError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libdpgo_hip.so is missing (%s): build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C dpgo_ros_amd/csrc`; there is no CPU fallback" % LIB_PATH)
        # One HIP runtime per process: torch ships its own libamdhip64 / librccl under /opt/rocm's sonames; loaded AFTER this
        # library (which binds /opt/rocm's) the process holds two runtimes and aborts at exit.  So where torch exists and has
        # not been imported yet, it goes first -- this library then binds to torch's copies through the sonames, whatever the
        # user's import order (DPGO_TORCH_FIRST=0: do not; dpgo_ros_amd.distributed needs torch anyway)
        if "torch" not in sys.modules and os.environ.get("DPGO_TORCH_FIRST", "1") != "0":
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        L.dpgo_last_error.restype = C.c_char_p
        L.dpgo_team_create.restype = C.c_void_p
        L.dpgo_team_stream.restype = C.c_void_p
        L.dpgo_comm_create.restype = C.c_void_p
        L.dpgo_agent_robust_weight.restype = C.c_double
        L.dpgo_error_threshold_at_quantile.restype = C.c_double
        _LIB = L
    return _LIB


def _d(a):
    return a.ctypes.data_as(C.c_void_p)


def two_level_plan(rowptr, col, max_sub=0):
    """the dissection of the two-level preconditioner for a block-CSR pattern (host arithmetic only):
    (sub_of[n] with -1 = separator, info dict)"""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    n = len(rowptr) - 1
    sub_of = np.zeros(n, dtype=np.int32)
    info = np.zeros(6)
    _chk(lib().dpgo_two_level_plan(n, _d(rowptr), _d(col), int(max_sub), _d(sub_of), _d(info)), "two_level_plan")
    return sub_of, dict(subdomains=int(info[0]), separator_poses=int(info[1]), workgroups=int(info[2]),
                        producer_workgroups=int(info[3]), bytes_per_apply=info[4], worthwhile=bool(info[5]))


def _chk(rc, what):
    if rc < 0:
        raise DpgoError("%s failed: %s" % (what, lib().dpgo_last_error().decode()))
    return rc


def default_params(r=5, num_robots=1, **kw):
    p = Params()
    lib().dpgo_default_params(C.byref(p), r, num_robots)
    for k, v in kw.items():
        if not hasattr(p, k):
            raise KeyError(k)
        setattr(p, k, v)
    return p


def read_g2o(path, weight_mode=WEIGHT_LIBRARY):
    out, n = C.c_void_p(), C.c_int()
    nm = lib().dpgo_read_g2o(path.encode(), weight_mode, C.byref(out), C.byref(n))
    if nm < 0:
        raise FileNotFoundError(path)
    raw = C.string_at(out, nm * MEAS_DTYPE.itemsize)
    lib().dpgo_free(out)
    return np.frombuffer(raw, dtype=MEAS_DTYPE).copy(), n.value


def read_csv(path, weight_mode=WEIGHT_LIBRARY):
    out = C.c_void_p()
    nm = lib().dpgo_read_measurements_csv(path.encode(), weight_mode, C.byref(out))
    if nm < 0:
        raise FileNotFoundError(path)
    raw = C.string_at(out, nm * MEAS_DTYPE.itemsize)
    lib().dpgo_free(out)
    return np.frombuffer(raw, dtype=MEAS_DTYPE).copy()


def robust_local_init(m, num_poses, params, device=0):
    """single-robot GNC-TLS solve on the device -> (T [12 n], final weights in input order)"""
    m = np.ascontiguousarray(m)
    T, w = np.zeros(12 * num_poses), np.zeros(len(m))
    _chk(lib().dpgo_robust_local_init(device, _d(m), len(m), num_poses, C.byref(params), _d(T), _d(w)), "robust_local_init")
    return T, w


def robust_frame_alignment(Tc, max_rotation_error_rad=0.5, max_translation_error=1.0, min_inliers=2):
    """Tc: n x 12 candidate transforms (3x4 column-major).  Returns (T, inlier mask) or None when fewer than
    min_inliers candidates agree."""
    Tc = np.ascontiguousarray(Tc, dtype=np.float64).reshape(-1, 12)
    T, inl = np.zeros(12), np.zeros(len(Tc), dtype=np.int32)
    rc = lib().dpgo_robust_frame_alignment(_d(Tc), len(Tc), C.c_double(max_rotation_error_rad),
                                           C.c_double(max_translation_error), min_inliers, _d(T), _d(inl))
    return (T, inl.astype(bool)) if rc == OK else None


def write_csv(path, m):
    """measurement list -> the CSV that read_csv / PGOLogger::loadMeasurements reads (weights included)"""
    m = np.ascontiguousarray(m)
    if lib().dpgo_write_measurements_csv(str(path).encode(), _d(m), len(m)) < 0:
        raise OSError("cannot write %s" % path)


def write_g2o(path, m, T=None, num_poses=0, robot_offsets=None):
    m = np.ascontiguousarray(m)
    Tp = _d(np.ascontiguousarray(T, dtype=np.float64)) if T is not None else None
    off = np.ascontiguousarray(robot_offsets, dtype=np.int32) if robot_offsets is not None else None
    if lib().dpgo_write_g2o(str(path).encode(), _d(m), len(m), Tp, num_poses if T is not None else 0,
                            _d(off) if off is not None else None) < 0:
        raise OSError("cannot write %s" % path)


def write_trajectory_csv(path, T, num_poses):
    T = np.ascontiguousarray(T, dtype=np.float64)
    if lib().dpgo_write_trajectory_csv(str(path).encode(), _d(T), num_poses) < 0:
        raise OSError("cannot write %s" % path)


class IterationLog:
    """per-iteration CSV in the column order of the reference's log (src/PGOAgentROS.cpp:863-864, 883-891),
    followed by the global-cost column the reference does not have."""
    HEADER = ("robot_id, cluster_id, num_active_robots, iteration, num_poses, bytes_received, "
              "iter_time_sec, total_time_sec, rel_change, global_cost \n")

    def __init__(self, path):
        self.f = open(path, "w")
        self.f.write(self.HEADER)

    def log(self, robot_id, cluster_id, num_active_robots, iteration, num_poses, bytes_received, iter_time_sec,
            total_time_sec, rel_change, global_cost=float("nan")):
        self.f.write("%d,%d,%d,%d,%d,%d,%.9g,%.9g,%.17g,%.17g\n" % (robot_id, cluster_id, num_active_robots, iteration,
                                                                     num_poses, bytes_received, iter_time_sec,
                                                                     total_time_sec, rel_change, global_cost))

    def log_string(self, s):  # "TERMINATE", "UPDATE_WEIGHT", ... (src/PGOAgentROS.cpp:896-909)
        self.f.write(s + "\n")

    def close(self):
        self.f.close()


def partition(m, num_poses, num_robots, weight_mode=WEIGHT_LIBRARY):
    m = m.copy()
    lib().dpgo_partition(_d(m), len(m), num_poses, num_robots, weight_mode)
    return m


def odometry_init(m, num_poses):
    T = np.zeros(12 * num_poses)
    lib().dpgo_odometry_init(_d(np.ascontiguousarray(m)), len(m), num_poses, _d(T))
    return T


def chordal_init(m, num_poses, device=0):
    T = np.zeros(12 * num_poses)
    _chk(lib().dpgo_chordal_init(device, _d(np.ascontiguousarray(m)), len(m), num_poses, _d(T)), "chordal_init")
    return T


def fixed_stiefel(r):
    Y = np.zeros(3 * r)
    lib().dpgo_fixed_stiefel(r, _d(Y))
    return Y


def lift(T, num_poses, YLift, r):
    X = np.zeros(r * 4 * num_poses)
    lib().dpgo_lift(_d(np.ascontiguousarray(T)), num_poses, _d(np.ascontiguousarray(YLift)), r, _d(X))
    return X


def error_threshold_at_quantile(q, dim):
    return lib().dpgo_error_threshold_at_quantile(C.c_double(q), dim)


class Agent:
    """One robot's block; method names follow DPGO::PGOAgent as used by PGOAgentROS."""

    def __init__(self, team, agent_id):
        self.team = team
        self.t = team.h
        self.id = agent_id
        self.r = team.r

    # --- structure
    def add_measurements(self, m):
        m = np.ascontiguousarray(m)
        _chk(lib().dpgo_agent_add_measurements(self.t, self.id, _d(m), len(m)), "addMeasurement")

    @property
    def n(self):
        return _chk(lib().dpgo_agent_num_poses(self.t, self.id), "num_poses")

    def _vec(self):
        return np.zeros(self.r * 4 * self.n)

    def _ids(self, fn, *args):
        c = _chk(fn(self.t, self.id, *args, None), fn.__name__)
        out = np.zeros(max(c, 1), dtype=np.int32)
        fn(self.t, self.id, *args, _d(out))
        return out[:c]

    def neighbors(self):
        return self._ids(lib().dpgo_agent_get_neighbors).tolist()

    def public_pose_ids(self, nbr):
        return self._ids(lib().dpgo_agent_public_pose_ids, nbr)

    def neighbor_pose_ids(self, nbr):
        return self._ids(lib().dpgo_agent_neighbor_pose_ids, nbr)

    # --- iterate / state
    def set_X(self, X):
        X = np.ascontiguousarray(X, dtype=np.float64)
        assert X.size == self.r * 4 * self.n
        _chk(lib().dpgo_agent_set_X(self.t, self.id, _d(X)), "set_X")

    def _get(self, which):
        X = self._vec()
        _chk(lib().dpgo_agent_get_X(self.t, self.id, which, _d(X)), "get_X")
        return X

    def get_X(self):
        return self._get(0)

    def get_Y(self):
        return self._get(1)

    def get_V(self):
        return self._get(2)

    def get_public_poses(self, nbr, aux=False):
        ids = self.public_pose_ids(nbr)
        out = np.zeros(max(len(ids), 1) * 4 * self.r)
        _chk(lib().dpgo_agent_get_public_poses(self.t, self.id, nbr, int(aux), _d(out)), "getSharedPoseDict")
        return ids, out[:len(ids) * 4 * self.r]

    def update_neighbor_poses(self, nbr, frames, poses, aux=False):
        frames = np.ascontiguousarray(frames, dtype=np.int32)
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        _chk(lib().dpgo_agent_update_neighbor_poses(self.t, self.id, nbr, int(aux), len(frames), _d(frames), _d(poses)),
             "updateNeighborPoses")

    def iterate(self, do_opt=True):
        return _chk(lib().dpgo_agent_iterate(self.t, self.id, int(do_opt)), "iterate") == OK

    def status(self):
        s = Status()
        _chk(lib().dpgo_agent_get_status(self.t, self.id, C.byref(s)), "getStatus")
        return s

    def opt_result(self):
        s = OptResult()
        _chk(lib().dpgo_agent_get_opt_result(self.t, self.id, C.byref(s)), "opt_result")
        return s

    def preconditioner(self):
        """1 dense inverse, 3 two-level (both exact), 2 block-Jacobi (on request, or where neither exact form fits)"""
        return _chk(lib().dpgo_agent_preconditioner(self.t, self.id), "preconditioner")

    def preconditioner_residual(self):
        """|z (Q + shift I) - v| / |v| for the operator the kernels apply (fixed pseudo-random v)"""
        rel = C.c_double()
        _chk(lib().dpgo_agent_preconditioner_residual(self.t, self.id, C.byref(rel)), "preconditioner_residual")
        return rel.value

    def preconditioner_info(self):
        out = np.zeros(8)
        _chk(lib().dpgo_agent_preconditioner_info(self.t, self.id, _d(out)), "preconditioner_info")
        return dict(mode=int(out[0]), subdomains=int(out[1]), separator_poses=int(out[2]), workgroups=int(out[3]),
                    producer_workgroups=int(out[4]), bytes_per_apply=out[5], dense_bytes=out[6], largest_subdomain=int(out[7]))

    def publish_requested(self, clear=False):
        return bool(lib().dpgo_agent_publish_requested(self.t, self.id, int(clear)))

    # --- QuadraticProblem surface
    def build_problem(self, aux=False):
        return _chk(lib().dpgo_agent_build_problem(self.t, self.id, int(aux)), "build_problem")

    def eval(self, X):
        f = C.c_double()
        eg, rg = self._vec(), self._vec()
        _chk(lib().dpgo_agent_eval(self.t, self.id, _d(np.ascontiguousarray(X)), C.byref(f), _d(eg), _d(rg)), "eval")
        return f.value, eg, rg

    def hessvec(self, X, eta):
        out = self._vec()
        _chk(lib().dpgo_agent_hessvec(self.t, self.id, _d(np.ascontiguousarray(X)), _d(np.ascontiguousarray(eta)), _d(out)),
             "hessvec")
        return out

    def precondition(self, X, V):
        out = self._vec()
        _chk(lib().dpgo_agent_precondition(self.t, self.id, _d(np.ascontiguousarray(X)), _d(np.ascontiguousarray(V)), _d(out)),
             "precondition")
        return out

    def get_Q(self):
        nb = _chk(lib().dpgo_agent_get_Q(self.t, self.id, None, None, None), "get_Q")
        rowptr = np.zeros(self.n + 1, dtype=np.int32)
        col = np.zeros(nb, dtype=np.int32)
        val = np.zeros(16 * nb)
        lib().dpgo_agent_get_Q(self.t, self.id, _d(rowptr), _d(col), _d(val))
        return rowptr, col, val

    def get_G(self):
        G = self._vec()
        _chk(lib().dpgo_agent_get_G(self.t, self.id, _d(G)), "get_G")
        return G

    # --- robust path
    def measurements(self):
        c = _chk(lib().dpgo_agent_get_measurements(self.t, self.id, None), "measurements")
        m = np.zeros(max(c, 1), dtype=MEAS_DTYPE)
        lib().dpgo_agent_get_measurements(self.t, self.id, _d(m))
        return m[:c]

    def compute_residual(self, meas_row):
        m = np.ascontiguousarray(np.array([meas_row], dtype=MEAS_DTYPE))
        res = C.c_double()
        rc = _chk(lib().dpgo_agent_compute_residual(self.t, self.id, _d(m), C.byref(res)), "computeMeasurementResidual")
        return rc == OK, res.value

    def robust_weight(self, residual):
        return lib().dpgo_agent_robust_weight(self.t, self.id, C.c_double(residual))

    def update_measurement_weights(self):
        _chk(lib().dpgo_agent_update_measurement_weights(self.t, self.id), "updateMeasurementWeights")

    def set_measurement_weight(self, r1, p1, r2, p2, w, fixed=False):
        return lib().dpgo_agent_set_measurement_weight(self.t, self.id, r1, p1, r2, p2, C.c_double(w), int(fixed)) == OK

    def clear_data_matrices(self):
        _chk(lib().dpgo_agent_clear_data_matrices(self.t, self.id), "clearDataMatrices")

    def pull_local(self):
        _chk(lib().dpgo_agent_pull_local(self.t, self.id), "pull_local")

    # --- device-buffer exchange (RCCL payloads)
    def pack_public_poses_device(self, nbr, aux, dev_ptr):
        return _chk(lib().dpgo_agent_pack_public_poses_device(self.t, self.id, nbr, int(aux), C.c_void_p(dev_ptr)), "pack")

    def unpack_neighbor_poses_device(self, nbr, aux, dev_ptr):
        return _chk(lib().dpgo_agent_unpack_neighbor_poses_device(self.t, self.id, nbr, int(aux), C.c_void_p(dev_ptr)), "unpack")


COMM_ID_BYTES = 128


def comm_unique_id():
    """DPGO_COMM_ID_BYTES bytes that name a new RCCL communicator: obtained on ONE rank, handed to every rank (any channel)"""
    b = (C.c_ubyte * COMM_ID_BYTES)()
    _chk(lib().dpgo_comm_unique_id(b), "comm_unique_id")
    return bytes(b)


def comm_library():
    """(path, version code) of the RCCL the library bound at run time"""
    buf = C.create_string_buffer(512)
    v = _chk(lib().dpgo_comm_library(buf, 512), "comm_library")
    return buf.value.decode(), v


def rank_plan_simulate(owner, npub, rank, world, sel_ids, acceleration=1, max_delayed_iterations=0, r=5):
    """the exchange's planning layer replayed for one rank (host arithmetic only): array [1 + iters, world, 4] of
    {doubles sent, doubles received, hash of the sent slabs, hash of the received slabs} per batch and peer rank"""
    owner = np.ascontiguousarray(owner, dtype=np.int32)
    N = len(owner)
    npub = np.ascontiguousarray(npub, dtype=np.int32).reshape(N, N)
    sel = np.ascontiguousarray(sel_ids, dtype=np.int32)
    out = np.zeros((1 + len(sel), world, 4), dtype=np.int64)
    _chk(lib().dpgo_rank_plan_simulate(N, int(world), int(rank), _d(owner), _d(npub), int(acceleration), int(max_delayed_iterations), int(r),
                                       _d(sel), len(sel), _d(out)), "rank_plan_simulate")
    return out


class Comm:
    """RCCL communicator owned by the library (csrc/rank_exchange.cpp): one per process, every rank takes part in its
    creation -- also the ranks that own no robot."""

    def __init__(self, unique_id, rank, world, device=0):
        b = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = lib().dpgo_comm_create(device, b, rank, world)
        if not h:
            raise DpgoError("dpgo_comm_create: " + lib().dpgo_last_error().decode())
        self.h = C.c_void_p(h)
        self.rank, self.world, self.device = rank, world, device

    def allreduce(self, values, op="sum", stream=None):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        fn = lib().dpgo_comm_allreduce_sum if op == "sum" else lib().dpgo_comm_allreduce_max
        _chk(fn(self.h, C.c_void_p(stream) if stream else None, _d(v), len(v)), "comm_allreduce")
        return v

    def global_cost(self, team=None, stream=None):
        f = C.c_double()
        _chk(lib().dpgo_comm_global_cost(self.h, team.h if team is not None else None, C.c_void_p(stream) if stream else None,
                                         C.byref(f)), "comm_global_cost")
        return f.value

    def close(self):
        if getattr(self, "h", None):
            lib().dpgo_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Team:
    """The agents resident on one GPU.  With every agent of the problem local, `run` executes the
    synchronous RBCD schedule entirely on the device."""

    def __init__(self, params, agent_ids, device=0, stream=None):
        self.params = params
        self.r = params.r
        ids = np.ascontiguousarray(agent_ids, dtype=np.int32)
        h = lib().dpgo_team_create(device, C.byref(params), len(ids), _d(ids), C.c_void_p(stream) if stream else None)
        if not h:
            raise DpgoError("dpgo_team_create: " + lib().dpgo_last_error().decode())
        self.h = C.c_void_p(h)
        self.agents = {int(i): Agent(self, int(i)) for i in ids}
        self.ids = [int(i) for i in ids]

    def close(self):
        if getattr(self, "h", None):
            lib().dpgo_team_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @classmethod
    def from_measurements(cls, meas, params, device=0, local_ids=None, stream=None):
        ids = list(range(params.num_robots)) if local_ids is None else list(local_ids)
        t = cls(params, ids, device=device, stream=stream)
        meas = np.ascontiguousarray(meas)
        for i in ids:
            t.agents[i].add_measurements(meas)
        return t

    def offsets(self):
        off, acc = [], 0
        for i in self.ids:
            off.append(acc)
            acc += self.agents[i].n
        return np.array(off, dtype=np.int32)

    def set_schedule(self, order):
        o = np.ascontiguousarray(order, dtype=np.int32)
        _chk(lib().dpgo_team_set_schedule(self.h, _d(o), len(o)), "set_schedule")

    def set_uniform_schedule(self, seed, length):
        """UpdateRule::Uniform (src/PGOAgentROS.cpp:446-463) with a given seed: draws `length` token holders, installs them as
        the schedule and returns them"""
        order = np.zeros(int(length), dtype=np.int32)
        _chk(lib().dpgo_team_set_uniform_schedule(self.h, C.c_uint(int(seed)), int(length), _d(order)), "set_uniform_schedule")
        return order

    def set_initial(self, T, YLift, offsets=None):
        off = self.offsets() if offsets is None else np.ascontiguousarray(offsets, dtype=np.int32)
        _chk(lib().dpgo_team_set_initial(self.h, _d(np.ascontiguousarray(T)), _d(np.ascontiguousarray(YLift)), _d(off)),
             "set_initial")

    def exchange_all(self):
        _chk(lib().dpgo_team_exchange_all(self.h), "exchange_all")

    def run(self, iters):
        _chk(lib().dpgo_team_run(self.h, iters), "team_run")

    def prepare(self, iters):
        """instantiate the graphs a run of `iters` iterations will replay, executing nothing"""
        _chk(lib().dpgo_team_prepare(self.h, iters), "prepare")

    def coloring(self):
        col = np.zeros(len(self.ids), dtype=np.int32)
        nc = _chk(lib().dpgo_team_get_coloring(self.h, _d(col)), "get_coloring")
        return nc, col

    def export_state(self, agent_id):
        """(64-byte IPC handle, offset of X, offset of Y, poses) of a local agent's arrays, for dpgo_team_import_peer"""
        h = (C.c_ubyte * 64)()
        ox, oy, n = C.c_longlong(), C.c_longlong(), C.c_int()
        _chk(lib().dpgo_agent_export_state(self.h, agent_id, h, C.byref(ox), C.byref(oy), C.byref(n)), "export_state")
        return bytes(h), ox.value, oy.value, n.value

    def import_peer(self, robot_id, handle, off_x, off_y, n):
        """read the public poses of a robot that lives in another process in place (HIP IPC / peer access)"""
        h = (C.c_ubyte * 64).from_buffer_copy(handle)
        _chk(lib().dpgo_team_import_peer(self.h, robot_id, h, C.c_longlong(off_x), C.c_longlong(off_y), n), "import_peer")

    def export_mailbox(self):
        """64-byte IPC handle of this team's mailbox (the device-side UPDATE token, dpgo_team_run_peer)"""
        h = (C.c_ubyte * 64)()
        _chk(lib().dpgo_team_export_mailbox(self.h, h), "export_mailbox")
        return bytes(h)

    def import_mailbox(self, handle, robot_ids):
        h = (C.c_ubyte * 64).from_buffer_copy(handle)
        ids = np.ascontiguousarray(robot_ids, dtype=np.int32)
        _chk(lib().dpgo_team_import_mailbox(self.h, h, _d(ids), len(ids)), "import_mailbox")

    def run_peer(self, sel_ids):
        """len(sel_ids) global iterations, robot sel_ids[q] holding the token in the q-th, enqueued without host
        synchronisation; remote neighbours are read in place and ordered by the device-side mailboxes"""
        ids = np.ascontiguousarray(sel_ids, dtype=np.int32)
        _chk(lib().dpgo_team_run_peer(self.h, _d(ids), len(ids)), "run_peer")

    def attach_comm(self, comm, owner_of_robot, max_delayed_iterations=0, loopback=False):
        """from here on run_ranks / exchange_all_ranks move the public poses between ranks with ncclSend / ncclRecv enqueued
        by the library on the team stream (include/dpgo_hip.h); loopback (world size 1): every pair as a self-send"""
        o = np.ascontiguousarray(owner_of_robot, dtype=np.int32)
        assert len(o) == self.params.num_robots
        _chk(lib().dpgo_team_attach_comm(self.h, comm.h, _d(o), int(max_delayed_iterations), int(bool(loopback))), "attach_comm")
        self._comm = comm  # (keeps the communicator alive as long as the team uses it)

    def detach_comm(self):
        _chk(lib().dpgo_team_detach_comm(self.h), "detach_comm")
        self._comm = None

    def exchange_all_ranks(self):
        _chk(lib().dpgo_team_exchange_all_ranks(self.h), "exchange_all_ranks")

    def run_ranks(self, sel_ids):
        """len(sel_ids) global iterations with robot sel_ids[q] holding the token in the q-th; the exchange is RCCL
        point-to-point inside the library, nothing synchronises with the host"""
        ids = np.ascontiguousarray(sel_ids, dtype=np.int32)
        _chk(lib().dpgo_team_run_ranks(self.h, _d(ids), len(ids)), "run_ranks")

    def run_simultaneous_ranks(self, ticks):
        """lockstep ASAPP ticks with the boundary slabs moved by the library (one batch of ncclSend / ncclRecv per tick)"""
        _chk(lib().dpgo_team_run_simultaneous_ranks(self.h, int(ticks)), "run_simultaneous_ranks")

    def run_group_ranks(self, g, count):
        """one colour class across ranks: members receive what moved, then update at once (set_groups first)"""
        _chk(lib().dpgo_team_run_group_ranks(self.h, int(g), int(count)), "run_group_ranks")

    def comm_counters(self):
        """messages sent / received by this rank and their bytes"""
        out = np.zeros(4)
        _chk(lib().dpgo_team_comm_counters(self.h, _d(out)), "comm_counters")
        return dict(messages_sent=int(out[0]), messages_received=int(out[1]), bytes_sent=out[2], bytes_received=out[3])

    def should_terminate(self):
        """PGOAgent::shouldTerminate() as the leader evaluates it (src/PGOAgentROS.cpp:208)"""
        return bool(_chk(lib().dpgo_team_should_terminate(self.h), "should_terminate"))

    def run_schedule(self, max_iters):
        """the synchronous schedule with the leader's TERMINATE / UPDATE_WEIGHT decisions (src/PGOAgentROS.cpp:206-214);
        returns (iterations executed, terminated, weight-update rounds)"""
        term, rounds = C.c_int(0), C.c_int(0)
        done = _chk(lib().dpgo_team_run_schedule(self.h, int(max_iters), C.byref(term), C.byref(rounds)), "run_schedule")
        return done, bool(term.value), rounds.value

    def set_iteration_log(self, directory):
        """one CSV per local robot in the reference's column order + global_cost (include/dpgo_hip.h); None closes"""
        _chk(lib().dpgo_team_set_iteration_log(self.h, str(directory).encode() if directory is not None else None), "set_iteration_log")

    def run_simultaneous(self, ticks):
        """every agent takes `ticks` RGD steps, all agents per tick in the same launches (ASAPP, clocks in lockstep)"""
        _chk(lib().dpgo_team_run_simultaneous(self.h, ticks), "run_simultaneous")

    def run_colored(self, sweeps):
        _chk(lib().dpgo_team_run_colored(self.h, sweeps), "run_colored")

    def set_groups(self, groups):
        ptr = np.cumsum([0] + [len(g) for g in groups]).astype(np.int32)
        mem = np.array([a for g in groups for a in g], dtype=np.int32)
        _chk(lib().dpgo_team_set_groups(self.h, len(groups), _d(ptr), _d(mem)), "set_groups")

    def run_group(self, g, count):
        _chk(lib().dpgo_team_run_group(self.h, g, count), "run_group")

    def step_begin(self, sel_id):
        _chk(lib().dpgo_team_step_begin(self.h, sel_id), "step_begin")

    def step_end(self, sel_id):
        _chk(lib().dpgo_team_step_end(self.h, sel_id), "step_end")

    def synchronize(self):
        _chk(lib().dpgo_team_synchronize(self.h), "synchronize")

    def iteration(self):
        return lib().dpgo_team_iteration(self.h)

    def cost(self):
        f = C.c_double()
        _chk(lib().dpgo_team_cost(self.h, C.byref(f)), "team_cost")
        return f.value

    def update_weights(self):
        return _chk(lib().dpgo_team_update_weights(self.h), "team_update_weights")

    def counters(self):
        out = np.zeros(11)
        lib().dpgo_team_get_counters(self.h, _d(out), 11)
        return out

    def stream(self):
        return lib().dpgo_team_stream(self.h)

    def time_kernel(self, agent_id, which, reps=200):
        ms, nbytes = C.c_double(), C.c_double()
        _chk(lib().dpgo_team_time_kernel(self.h, agent_id, which, reps, C.byref(ms), C.byref(nbytes)), "time_kernel")
        return ms.value, nbytes.value

    def global_X(self):
        return np.concatenate([self.agents[i].get_X() for i in self.ids])
