"""Shared helpers of the parity tests: one set of seeded inputs, fed to the HIP path (through the
C-ABI) and to the CPU oracle."""
import os

import numpy as np

from dpgo_ros_amd import capi
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "data")


def load(dataset, num_robots, weight_mode=0):
    m, n = O.read_g2o(os.path.join(DATA, dataset + ".g2o"), weight_mode)
    mp = O.partition(m, n, num_robots, weight_mode) if num_robots > 1 else m
    return m, mp, n


def params_pair(**kw):
    return capi.default_params(**kw), O.default_params(**kw)


def random_point(rng, r, n):
    X = rng.standard_normal(r * 4 * n)
    return O.project_manifold(X, r, n)


def make_pair(dataset, num_robots, r=5, init="odom", **kw):
    """(hip Team, oracle Team, n) on identical inputs and initial guess."""
    m, mp, n = load(dataset, num_robots)
    ph, po = params_pair(r=r, num_robots=num_robots, **kw)
    T = O.odometry_init(m, n) if init == "odom" else O.chordal_init(m, n)
    Y = O.fixed_stiefel(r)
    th = capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE), ph)
    to = O.Team(mp, n, po)
    th.set_initial(T, Y)
    to.set_initial(T, Y)
    return th, to, n


def relerr(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def add_outliers(m, n, frac=0.1, seed=0):
    """seeded synthetic outlier loop closures (10 % extra edges, endpoints uniform, R uniform
    on SO(3), t uniform in the bounding box of the odometry-chained trajectory)."""
    rng = np.random.default_rng(seed)
    T = O.odometry_init(m, n).reshape(n, 4, 3)
    lo, hi = T[:, 3, :].min(0), T[:, 3, :].max(0)
    k = max(1, int(frac * len(m)))
    out = np.zeros(k, dtype=O.MEAS_DTYPE)
    for e in range(k):
        i, j = rng.integers(0, n, 2)
        while abs(int(i) - int(j)) < 2:
            i, j = rng.integers(0, n, 2)
        Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        Q *= np.sign(np.linalg.det(Q))
        out[e]["p1"], out[e]["p2"] = i, j
        out[e]["R"] = Q.reshape(-1)
        out[e]["t"] = lo + rng.random(3) * (hi - lo)
        out[e]["kappa"], out[e]["tau"], out[e]["weight"] = np.median(m["kappa"]), np.median(m["tau"]), 1.0
    return np.concatenate([m, out])


def load_tunnels(weight_mode=0):
    """the 8 per-robot CSVs merged into one edge list (a shared edge listed by both robots is kept once)"""
    seen, rows = set(), []
    for k in range(8):
        for e in O.read_csv(os.path.join(DATA, "tunnels", "robot%d" % k, "measurements.csv"), weight_mode):
            key = (int(e["r1"]), int(e["p1"]), int(e["r2"]), int(e["p2"]))
            if key not in seen:
                seen.add(key)
                rows.append(e)
    return np.array(rows, dtype=O.MEAS_DTYPE)


def merged_graph(names=("torus3D", "cubicle", "parking-garage"), outlier_frac=0.1, seed=0):
    """BASELINE configs[3] in the merged form SURVEY 8d-4 names: the datasets are concatenated into ONE pose graph
    (grid3D / rim are absent from the reference mount, so torus3D + cubicle + parking-garage stand in -- a declared
    substitution), pose indices offset per component, each component tied to the previous one by a single identity-pose
    edge between its first pose and the previous component's last pose (so that the odometry chain and the contiguous
    partition rule see one connected graph), plus seeded outlier loop closures per component (10 % extra edges,
    endpoints uniform, R uniform on SO(3), t uniform in the component's bounding box).  Agents of the contiguous
    partition then hold disjoint components with kappa from 2e-9 (garage) to 200 and anisotropic tau (cubicle)."""
    parts, off = [], 0
    for c, name in enumerate(names):
        m, n = O.read_g2o(os.path.join(DATA, name + ".g2o"))
        mo = add_outliers(m, n, frac=outlier_frac, seed=seed + c) if outlier_frac > 0 else m.copy()
        mo = mo.copy()
        mo["p1"] += off
        mo["p2"] += off
        if c > 0:
            tie = np.zeros(1, dtype=O.MEAS_DTYPE)
            tie["p1"], tie["p2"] = off - 1, off
            tie["R"] = np.eye(3).reshape(-1)
            tie["kappa"], tie["tau"], tie["weight"] = 1.0, 1.0, 1.0
            tie["fixed_weight"] = 1
            parts.append(tie)
        parts.append(mo)
        off += n
    return np.concatenate(parts), off


def synthetic_chain(n, seed=0, lc_every=40, noise=0.01):
    """a long single-robot pose graph: random-walk ground truth, odometry i -> i+1 and a loop closure i -> i+lc_every/2
    every lc_every poses, measurements perturbed by small rotations / translations (seeded); kappa 100, tau 50"""
    rng = np.random.default_rng(seed)

    def rot(w):
        th = np.linalg.norm(w, axis=-1, keepdims=True)
        k = w / np.maximum(th, 1e-12)
        K = np.zeros(w.shape[:-1] + (3, 3))
        K[..., 0, 1], K[..., 0, 2], K[..., 1, 0] = -k[..., 2], k[..., 1], k[..., 2]
        K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -k[..., 0], -k[..., 1], k[..., 0]
        s, c = np.sin(th)[..., None], (1 - np.cos(th))[..., None]
        return np.eye(3) + s * K + c * (K @ K)

    dR = rot(0.2 * rng.standard_normal((n - 1, 3)))
    dt = np.c_[np.ones(n - 1), 0.1 * rng.standard_normal((n - 1, 2))]
    Rg, tg = np.zeros((n, 3, 3)), np.zeros((n, 3))
    Rg[0] = np.eye(3)
    for i in range(n - 1):
        Rg[i + 1] = Rg[i] @ dR[i]
        tg[i + 1] = tg[i] + Rg[i] @ dt[i]
    src = np.r_[np.arange(n - 1), np.arange(0, n - lc_every, lc_every)]
    dst = np.r_[np.arange(1, n), np.arange(0, n - lc_every, lc_every) + lc_every // 2]
    m = np.zeros(len(src), dtype=O.MEAS_DTYPE)
    Rm = np.einsum("eji,ejk->eik", Rg[src], Rg[dst]) @ rot(noise * rng.standard_normal((len(src), 3)))
    tm = np.einsum("eji,ej->ei", Rg[src], tg[dst] - tg[src]) + noise * rng.standard_normal((len(src), 3))
    m["p1"], m["p2"] = src, dst
    m["R"] = Rm.reshape(len(src), 9)
    m["t"] = tm
    m["kappa"], m["tau"], m["weight"] = 100.0, 50.0, 1.0
    return m, n
