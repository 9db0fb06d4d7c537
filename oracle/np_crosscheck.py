#!/usr/bin/env python3
"""Independent numpy/scipy statement of the RBCD path -> golden fixtures under tests/golden/.

TEST INFRASTRUCTURE ONLY.  The reference (mit-acl/dpgo behind /root/reference) cannot be built or
imported in this image, so there are no reference-generated vectors (parity unpinned, see
oracle/dpgo_oracle.h).  This script is the second, independently written implementation that
SURVEY.md 8c asks for: dense linear algebra (numpy SVD / QR / solve, dense Q assembled straight from
the cost definition) instead of the C oracle's hand-rolled Jacobi / Gram-Schmidt / sparse Cholesky.
Agreement between the two pins each against typos; it does not pin either against dpgo.

Usage:  python oracle/np_crosscheck.py          (re-generates tests/golden/*.npz, deterministic)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


# ----------------------------------------------------------------------------- input
def quat_to_rot(q):
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def read_g2o(path):
    edges = []
    for line in open(path):
        tok = line.split()
        if not tok or tok[0] != "EDGE_SE3:QUAT":
            continue
        i, j = int(tok[1]), int(tok[2])
        v = np.array(tok[3:10], dtype=float)
        info = np.zeros((6, 6))
        info[np.triu_indices(6)] = np.array(tok[10:31], dtype=float)
        info = info + info.T - np.diag(np.diag(info))
        tau = 3.0 / np.trace(np.linalg.inv(info[:3, :3]))
        kappa = 3.0 / (2.0 * np.trace(np.linalg.inv(info[3:, 3:])))
        edges.append(dict(i=i, j=j, R=quat_to_rot(v[3:7]), t=v[:3], kappa=kappa, tau=tau, w=1.0))
    n = 1 + max(max(e["i"], e["j"]) for e in edges)
    return edges, n


def partition(edges, n, N):
    per = n // N
    out = []
    for e in edges:
        ra, rb = min(e["i"] // per, N - 1), min(e["j"] // per, N - 1)
        out.append(dict(e, r1=ra, p1=e["i"] - ra * per, r2=rb, p2=e["j"] - rb * per))
    return out


# ----------------------------------------------------------------------------- manifold
def polar(A):
    U, _, Vt = np.linalg.svd(A, full_matrices=False)
    return U @ Vt


def project_manifold(X, n):
    X = X.copy()
    for i in range(n):
        X[:, 4 * i:4 * i + 3] = polar(X[:, 4 * i:4 * i + 3])
    return X


def tangent_project(X, V, n):
    out = V.copy()
    for i in range(n):
        Y, W = X[:, 4 * i:4 * i + 3], V[:, 4 * i:4 * i + 3]
        S = Y.T @ W
        out[:, 4 * i:4 * i + 3] = W - Y @ (0.5 * (S + S.T))
    return out


def retract(X, eta, n):
    out = X + eta
    for i in range(n):
        Q, Rr = np.linalg.qr(out[:, 4 * i:4 * i + 3])
        out[:, 4 * i:4 * i + 3] = Q * np.sign(np.diag(Rr))
    return out


# ----------------------------------------------------------------------------- one agent's problem
class Problem:
    """f(X) = 1/2 <Q, X^T X> + <G, X> of agent `aid`, Q dense, from the definition of the edge cost
    w/2 (kappa |Y_j - Y_i R|^2 + tau |p_j - p_i - Y_i t|^2) = w/2 |(X_j - X_i T) Omega^(1/2)|^2."""

    def __init__(self, edges, aid, r, nbr_pose):
        self.aid, self.r = aid, r
        mine = [e for e in edges if e["r1"] == aid or e["r2"] == aid]
        n = 0
        for e in mine:
            if e["r1"] == aid:
                n = max(n, e["p1"] + 1)
            if e["r2"] == aid:
                n = max(n, e["p2"] + 1)
        self.n = n
        Q = np.zeros((4 * n, 4 * n))
        G = np.zeros((r, 4 * n))
        for e in mine:
            T = np.eye(4); T[:3, :3] = e["R"]; T[:3, 3] = e["t"]
            Om = np.diag([e["kappa"]] * 3 + [e["tau"]]) * e["w"]
            # selector form: residual = X S, S = E_j - E_i T for local endpoints
            i_loc, j_loc = e["r1"] == aid, e["r2"] == aid
            Si = np.zeros((4 * n, 4)); Sj = np.zeros((4 * n, 4))
            if i_loc:
                Si[4 * e["p1"]:4 * e["p1"] + 4] = T
            if j_loc:
                Sj[4 * e["p2"]:4 * e["p2"] + 4] = np.eye(4)
            S = Sj - Si
            Q += S @ Om @ S.T
            if not i_loc:   # constant part  C = -X_i T  (neighbour):  residual = X Sj + C
                C = -nbr_pose[(e["r1"], e["p1"])] @ T
                G += C @ Om @ Sj.T
            if not j_loc:   # residual = -X Si + X_j
                C = nbr_pose[(e["r2"], e["p2"])]
                G += -C @ Om @ Si.T
        self.Q, self.G = Q, G
        self.P = Q + 0.1 * np.eye(4 * n)

    def f(self, X):
        return 0.5 * np.sum((X @ self.Q) * X) + np.sum(self.G * X)

    def egrad(self, X):
        return X @ self.Q + self.G

    def rgrad(self, X):
        return tangent_project(X, self.egrad(X), self.n)

    def hess(self, X, eta):
        E = self.egrad(X)
        H = eta @ self.Q
        for i in range(self.n):
            Y, Ey = X[:, 4 * i:4 * i + 3], E[:, 4 * i:4 * i + 3]
            S = Y.T @ Ey
            H[:, 4 * i:4 * i + 3] -= eta[:, 4 * i:4 * i + 3] @ (0.5 * (S + S.T))
        return tangent_project(X, H, self.n)

    def precond(self, X, V):
        return tangent_project(X, np.linalg.solve(self.P, V.T).T, self.n)


def tcg(prob, X, g, Delta, max_inner):
    eta = np.zeros_like(g); r = g.copy()
    norm_r0 = np.linalg.norm(r)
    z = prob.precond(X, r)
    z_r = np.sum(z * r); d_Pd = z_r; e_Pd = 0.0; e_Pe = 0.0
    delta = -z
    status = "maxiter"; iters = 0
    for _ in range(max_inner):
        Hd = prob.hess(X, delta); iters += 1
        d_Hd = np.sum(delta * Hd)
        alpha = z_r / d_Hd
        e_Pe_new = e_Pe + 2 * alpha * e_Pd + alpha * alpha * d_Pd
        if d_Hd <= 0 or e_Pe_new >= Delta ** 2:
            tau = (-e_Pd + np.sqrt(e_Pd ** 2 + d_Pd * (Delta ** 2 - e_Pe))) / d_Pd
            eta = eta + tau * delta
            status = "negcurv" if d_Hd <= 0 else "boundary"
            break
        e_Pe = e_Pe_new
        eta = eta + alpha * delta
        r = r + alpha * Hd
        if np.linalg.norm(r) <= norm_r0 * min(norm_r0, 0.1):
            status = "converged"
            break
        z = prob.precond(X, r)
        z_r_new = np.sum(z * r)
        beta = z_r_new / z_r; z_r = z_r_new
        delta = -z + beta * delta
        e_Pd = beta * (e_Pd + alpha * d_Pd)
        d_Pd = z_r + beta * beta * d_Pd
    return eta, status, iters


def rtr(prob, X0, max_outer=3, max_inner=50, tol=1e-2, Delta0=100.0, Delta_max=500.0):
    X = X0.copy(); f1 = prob.f(X); g = prob.rgrad(X); Delta = Delta0
    total = 0; accepted = 0
    for _ in range(max_outer):
        if np.linalg.norm(g) < tol:
            break
        eta, status, it = tcg(prob, X, g, Delta, max_inner); total += it
        X2 = retract(X, eta, prob.n); f2 = prob.f(X2)
        Heta = prob.hess(X, eta)
        rho = (f1 - f2) / (-np.sum(g * eta) - 0.5 * np.sum(eta * Heta))
        if rho > 0.75:
            if status in ("negcurv", "boundary"):
                Delta = min(2 * Delta, Delta_max)
        elif rho < 0.25:
            Delta *= 0.25
        if rho > 0.1:
            X, f1 = X2, f2; g = prob.rgrad(X); accepted += 1
    return X, total, accepted


def rgd(prob, X0, step, use_precond=True):
    g = prob.rgrad(X0)
    d = prob.precond(X0, g) if use_precond else g
    return retract(X0, -step * d, prob.n)


# ----------------------------------------------------------------------------- team
class TeamNP:
    def __init__(self, edges, N, r, T0, method="rtr", accel=False, step=0.2, restart=50):
        self.edges, self.N, self.r = edges, N, r
        self.method, self.accel, self.step, self.restart = method, accel, step, restart
        self.n = [0] * N
        for e in edges:
            self.n[e["r1"]] = max(self.n[e["r1"]], e["p1"] + 1)
            self.n[e["r2"]] = max(self.n[e["r2"]], e["p2"] + 1)
        off = np.concatenate([[0], np.cumsum(self.n)])
        Ylift = np.zeros((r, 3)); Ylift[:3, :3] = np.eye(3)
        self.X = [Ylift @ T0[:, 4 * off[k]:4 * off[k + 1]] for k in range(N)]
        self.Y = [x.copy() for x in self.X]; self.V = [x.copy() for x in self.X]
        self.gamma = [0.0] * N; self.alpha = [0.0] * N; self.it = [0] * N
        self.k = 0

    def nbr(self, aid, seq):
        d = {}
        for e in self.edges:
            if e["r1"] == aid and e["r2"] != aid:
                d[(e["r2"], e["p2"])] = seq[e["r2"]][:, 4 * e["p2"]:4 * e["p2"] + 4]
            if e["r2"] == aid and e["r1"] != aid:
                d[(e["r1"], e["p1"])] = seq[e["r1"]][:, 4 * e["p1"]:4 * e["p1"] + 4]
        return d

    def solve(self, a, X0, seq):
        prob = Problem(self.edges, a, self.r, self.nbr(a, seq))
        if self.method == "rtr":
            return rtr(prob, X0)[0]
        return rgd(prob, X0, self.step)

    def iterate(self):
        sel = self.k % self.N
        order = [b for b in range(self.N) if b != sel] + [sel]
        for b in order:
            opt = b == sel
            self.it[b] += 1
            Xprev = self.X[b].copy()
            if self.accel:
                Nr = float(self.N)
                self.gamma[b] = (1 + np.sqrt(1 + 4 * Nr * Nr * self.gamma[b] ** 2)) / (2 * Nr)
                self.alpha[b] = 1.0 / (self.gamma[b] * Nr)
                self.Y[b] = project_manifold((1 - self.alpha[b]) * self.X[b] + self.alpha[b] * self.V[b], self.n[b])
                self.X[b] = self.solve(b, self.Y[b], self.Y) if opt else self.Y[b].copy()
                self.V[b] = project_manifold(self.V[b] + self.gamma[b] * (self.X[b] - self.Y[b]), self.n[b])
                if (self.it[b] + 1) % self.restart == 0:
                    self.X[b] = Xprev
                    if opt:
                        self.X[b] = self.solve(b, self.X[b], self.X)
                    self.V[b] = self.X[b].copy(); self.Y[b] = self.X[b].copy()
                    self.gamma[b] = 0.0; self.alpha[b] = 0.0
            elif opt:
                self.X[b] = self.solve(b, self.X[b], self.X)
        self.k += 1

    def cost(self):
        f = 0.0
        for e in self.edges:
            Xi = self.X[e["r1"]][:, 4 * e["p1"]:4 * e["p1"] + 4]
            Xj = self.X[e["r2"]][:, 4 * e["p2"]:4 * e["p2"] + 4]
            f += 0.5 * e["w"] * (e["kappa"] * np.sum((Xj[:, :3] - Xi[:, :3] @ e["R"]) ** 2)
                                 + e["tau"] * np.sum((Xj[:, 3] - Xi[:, 3] - Xi[:, :3] @ e["t"]) ** 2))
        return f


def odometry(edges, n):
    T = np.zeros((3, 4 * n)); T[:, :3] = np.eye(3)
    odo = {}
    for e in edges:
        if e["j"] == e["i"] + 1 and e["i"] not in odo:
            odo[e["i"]] = e
    for i in range(n - 1):
        R, t = T[:, 4 * i:4 * i + 3], T[:, 4 * i + 3]
        if i in odo:
            T[:, 4 * i + 4:4 * i + 7] = R @ odo[i]["R"]
            T[:, 4 * i + 7] = t + R @ odo[i]["t"]
        else:
            T[:, 4 * i + 4:4 * i + 8] = T[:, 4 * i:4 * i + 4]
    return T


def flat(X):
    return np.asarray(X).reshape(-1, order="F")


def generate(dataset, N, r=5):
    edges, n = read_g2o(os.path.join(ROOT, "data", dataset + ".g2o"))
    pe = partition(edges, n, N)
    T0 = odometry(edges, n)
    out = {"num_poses": n, "num_robots": N, "r": r, "T0": flat(T0)}
    rng = np.random.default_rng(12345)
    team = TeamNP(pe, N, r, T0)
    for a in range(N):
        prob = Problem(pe, a, r, team.nbr(a, team.X))
        X = project_manifold(rng.standard_normal((r, 4 * prob.n)), prob.n)
        eta = tangent_project(X, rng.standard_normal(X.shape), prob.n)
        V = rng.standard_normal(X.shape)
        out["a%d_Qdense" % a] = prob.Q
        out["a%d_G" % a] = flat(prob.G)
        out["a%d_X" % a] = flat(X); out["a%d_eta" % a] = flat(eta); out["a%d_V" % a] = flat(V)
        out["a%d_f" % a] = prob.f(X)
        out["a%d_egrad" % a] = flat(prob.egrad(X)); out["a%d_rgrad" % a] = flat(prob.rgrad(X))
        out["a%d_hess" % a] = flat(prob.hess(X, eta)); out["a%d_precond" % a] = flat(prob.precond(X, V))
        out["a%d_retract" % a] = flat(retract(X, 0.3 * eta, prob.n))
        out["a%d_project" % a] = flat(project_manifold(X + 0.2 * V, prob.n))
        out["a%d_rgd" % a] = flat(rgd(prob, team.X[a], 0.2))
        Xr, tot, acc = rtr(prob, team.X[a])
        out["a%d_rtr" % a] = flat(Xr); out["a%d_rtr_tcg" % a] = tot; out["a%d_rtr_acc" % a] = acc
    for name, kw in (("rtr", dict(method="rtr")), ("rtr_acc", dict(method="rtr", accel=True, restart=7)),
                     ("rgd_acc", dict(method="rgd", accel=True, step=0.2, restart=7))):
        t = TeamNP(pe, N, r, T0, **kw)
        costs = []
        for _ in range(10):
            t.iterate(); costs.append(t.cost())
        out["team_%s_cost" % name] = np.array(costs)
        out["team_%s_X" % name] = np.concatenate([flat(x) for x in t.X])
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, "%s_N%d_r%d.npz" % (dataset, N, r)), **out)
    print("wrote", dataset, N, "final costs", {k: float(v[-1]) for k, v in out.items() if k.endswith("_cost")})


if __name__ == "__main__":
    generate("tinyGrid3D", 2)
    generate("smallGrid3D", 2)
