// chordal.hip -- two-stage chordal initialisation on the GPU (PGOAgent::initialize() with
// InitializationMethod::Chordal, src/PGOAgentROS.cpp:348, src/PGOAgentROSNode.cpp:106-112).
//
// Published algorithm (Carlone et al., ICRA 2015; the "chordal initialization" of SE-Sync / dpgo):
//   (1) rotations: minimise sum_e kappa_e |R_j - R_i R~_e|_F^2 over unconstrained 3x3 blocks with R_0 = I
//       (a linear SPD system in the rotation connection Laplacian), then project every block to SO(3);
//   (2) translations: minimise sum_e tau_e |t_j - t_i - R_i t~_e|^2 with t_0 = 0 (a scalar graph Laplacian).
// MI355X-first: both SPD systems are inverted densely with the blocked Cholesky of dense_inverse.hip
// (7500^2 doubles = 450 MB for the whole sphere2500 graph; 288 GB available) and applied with one
// coalesced kernel; no sparse factorisation, no host arithmetic beyond assembling the edge triplets.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "../../include/dpgo_hip.h"
#include "device_math.h"
#include "kernels.h"
#include "team_internal.h"

namespace dpgo {

struct Trip { int row, col; double v; };

__global__ void k_scatter(const Trip *t, int nt, double *A, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nt) A[(size_t)t[i].col * N + t[i].row] = t[i].v;
}

// out (r x N, column-major r-vectors) = B (r x N) * M (N x N symmetric): one thread per output column,
// M read along rows of the symmetric matrix so that consecutive threads touch consecutive addresses
template <int RR>
__global__ void k_apply_sym(const double *B, const double *M, double *out, int N) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  double acc[RR];
#pragma unroll
  for (int a = 0; a < RR; ++a) acc[a] = 0;
  for (int k = 0; k < N; ++k) {
    const double m = M[(size_t)k * N + c];
#pragma unroll
    for (int a = 0; a < RR; ++a) acc[a] += B[(size_t)k * RR + a] * m;
  }
#pragma unroll
  for (int a = 0; a < RR; ++a) out[(size_t)c * RR + a] = acc[a];
}

// nearest rotation to each 3x3 block (column-major, 9 doubles per pose), det-corrected
__global__ void k_project_so3(double *Rm, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double A[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) A[e] = Rm[(size_t)9 * i + e];
  double S[9], w[3], V[9];
  gram3<3>(A, S);
  sym3_eig(S, w, V);
  const double det = A[0] * (A[4] * A[8] - A[7] * A[5]) - A[3] * (A[1] * A[8] - A[7] * A[2]) + A[6] * (A[1] * A[5] - A[4] * A[2]);
  int kmin = 0;
  if (w[1] < w[kmin]) kmin = 1;
  if (w[2] < w[kmin]) kmin = 2;
  double Mx[9], T[9];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) s += ((k == kmin && det < 0) ? -1.0 : 1.0) * V[3 * p + k] * V[3 * q + k] / sqrt(w[k]);
      Mx[3 * p + q] = s;
    }
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      double s = 0;
#pragma unroll
      for (int p = 0; p < 3; ++p) s += A[p * 3 + a] * Mx[3 * p + q];
      T[q * 3 + a] = s;
    }
#pragma unroll
  for (int e = 0; e < 9; ++e) Rm[(size_t)9 * i + e] = T[e];
}

}  // namespace dpgo

using namespace dpgo;

namespace {

struct DBuf {
  void *p = nullptr;
  ~DBuf() { if (p) (void)hipFree(p); }
  bool alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 8)) == hipSuccess; }
};


// solve  X A = B  for X (rr x N) with A given by merged triplets; returns X on the host.
// The right-hand sides ride in the factorisation as RR extra rows of the matrix (dense_spd_solve, augmented): with
// A' = [A B^T; B C I] the last rows of the Cholesky factor are (L^-1 b_a)^T -- the forward substitution for free --, C only
// has to keep A' positive definite (C > b_a^T A^-1 b_a) and touches nothing but the appended block itself.
template <int RR>
int dense_solve(hipStream_t s, const std::vector<Trip> &trips_in, int N, const std::vector<double> &B,
                std::vector<double> &X) {
  static const bool aug = !(std::getenv("DPGO_CHORDAL_AUG") && std::getenv("DPGO_CHORDAL_AUG")[0] == '0');  // (0: forward substitution as launches)
  const int Np = aug ? N + RR : N;
  std::vector<Trip> trips = trips_in;
  if (aug)
  for (int k = 0; k < N; ++k)
    for (int a = 0; a < RR; ++a) {
      const double v = B[(size_t)k * RR + a];
      if (v != 0.0) trips.push_back(Trip{N + a, k, v});  // (the factorisation reads the lower triangle only)
    }
  if (aug) for (int a = 0; a < RR; ++a) trips.push_back(Trip{N + a, N + a, 1e40});
  // (keeping the N^2 buffer between calls was measured and is SLOWER: 26.5 against 23.3 ms per call on sphere2500)
  DBuf dT, dA, dB, dX;
  const size_t NN = (size_t)Np * Np * sizeof(double);
  if (!dT.alloc(sizeof(Trip) * trips.size()) || !dA.alloc(NN) || !dB.alloc(sizeof(double) * RR * Np) ||
      !dX.alloc(sizeof(double) * RR * Np)) return -1;
  if (hipMemcpyAsync(dT.p, trips.data(), sizeof(Trip) * trips.size(), hipMemcpyHostToDevice, s) != hipSuccess) return -1;
  if (!aug && hipMemcpyAsync(dB.p, B.data(), sizeof(double) * RR * N, hipMemcpyHostToDevice, s) != hipSuccess) return -1;
  if (hipMemsetAsync(dA.p, 0, NN, s) != hipSuccess) return -1;
  hipLaunchKernelGGL(k_scatter, dim3(((int)trips.size() + 255) / 256), dim3(256), 0, s, (const Trip *)dT.p, (int)trips.size(),
                     (double *)dA.p, Np);
  // Cholesky factor + ONE block substitution (backward): a third of the arithmetic of the inverse, one N^2 buffer
  if (dense_spd_solve<RR>(s, (double *)dA.p, Np, (double *)dB.p, (double *)dX.p, aug) != 0) return -2;
  X.resize((size_t)RR * N);
  if (hipMemcpyAsync(X.data(), dB.p, sizeof(double) * RR * N, hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
  if (hipStreamSynchronize(s) != hipSuccess) return -1;
  return 0;
}

// Assembly of a sparse symmetric system as (row, col, value) triplets with every (row, col) listed once.  Diagonal
// entries accumulate in a dense vector; off-diagonal blocks are unique per measurement unless two measurements join the
// same pair of poses, in which case (rare) everything goes through an ordered map as before.  (One std::map insertion
// per scalar entry -- 120 000 of them for sphere2500 -- was 12 of the 43 ms of the whole initialisation.)
struct Assembler {
  int N;
  std::vector<double> diag;
  std::vector<Trip> off;
  std::map<std::pair<int, int>, double> merged;
  bool use_map;
  Assembler(int n, bool dup) : N(n), diag((size_t)n, 0.0), use_map(dup) {}
  void add(int row, int col, double v) {
    if (use_map) merged[{row, col}] += v;
    else if (row == col) diag[row] += v;
    else off.push_back(Trip{row, col, v});
  }
  std::vector<Trip> triplets() const {
    std::vector<Trip> t;
    if (use_map) {
      t.reserve(merged.size());
      for (const auto &kv : merged) t.push_back(Trip{kv.first.first, kv.first.second, kv.second});
      return t;
    }
    t = off;
    for (int i = 0; i < N; ++i) if (diag[i] != 0.0) t.push_back(Trip{i, i, diag[i]});
    return t;
  }
};

bool has_parallel_edges(const dpgo_measurement_t *m, int nm, int n) {
  std::vector<long long> keys((size_t)nm);
  for (int e = 0; e < nm; ++e) {
    const long long a = std::min(m[e].p1, m[e].p2), b = std::max(m[e].p1, m[e].p2);
    keys[e] = a * (long long)n + b;
  }
  std::sort(keys.begin(), keys.end());
  return std::adjacent_find(keys.begin(), keys.end()) != keys.end();
}

}  // namespace

// ---- the relaxation through the solver's own machinery (round 5) -------------------------------------------------------
// With every measured translation set to zero the SE(3) connection Laplacian Q of the solver (assembly.hip) falls apart
// into exactly the two matrices above: its rotation rows / columns are the rotation connection Laplacian of stage 1
// (kappa I on the diagonal, -kappa R~ off it), its translation entries the scalar graph Laplacian of stage 2 (tau).  Pinning
// pose 0 is what a SHARED edge does: pose 0 is handed to a second "robot", so the first robot's Q holds the diagonal terms
// of the edges into pose 0 and its linear term G the right-hand side (R_0 = I) -- stage 1 is X = -G Q^-1 on the rotation
// entries, stage 2 the same operator on the translation entries with a right-hand side built from the projected rotations.
// Q^-1 is the preconditioner with shift 0, which the library builds in its two-level (nested dissection / Schur) form for a
// graph of this size: 20 subdomain blocks of <= 700^2 and one separator block of ~1200^2 for sphere2500 -- a chain of
// ~80 block steps in batched launches where the dense Cholesky of the 7500^2 system walks 312 --, applied with one launch.
// One step of iterative refinement (residual through the sparse operator) brings the explicit-inverse form to the
// accuracy of a triangular solve.  DPGO_CHORDAL_DENSE=1 keeps the dense path above.
static int chordal_via_team(int device, const dpgo_measurement_t *m, int nm, int n, double *T) {
  const int r = 3, n1 = n - 1;
  std::vector<dpgo_measurement_t> mm;
  mm.reserve(nm);
  for (int e = 0; e < nm; ++e) {
    if (m[e].p1 == m[e].p2) continue;
    dpgo_measurement_t q = m[e];
    q.t[0] = q.t[1] = q.t[2] = 0.0;
    q.r1 = q.p1 == 0 ? 1 : 0; q.p1 = q.p1 == 0 ? 0 : q.p1 - 1;
    q.r2 = q.p2 == 0 ? 1 : 0; q.p2 = q.p2 == 0 ? 0 : q.p2 - 1;
    mm.push_back(q);
  }
  dpgo_params_t p;
  dpgo_default_params(&p, r, 2);
  p.precond_shift = 0.0;
  const int id0 = 0;
  static const bool timing = std::getenv("DPGO_TIMING") != nullptr;
  auto tq = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "chordal: %-28s %.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tq).count());
    tq = now;
  };
  dpgo_team_t *t = dpgo_team_create(device, &p, 1, &id0, nullptr);
  if (!t) return DPGO_ERR;
  t->tl_max_sub = 200;  // (a fixed dissection: the search for the cheapest apply costs more than this whole relaxation)
  lap("team_create");
  int rc = DPGO_ERR;
  const size_t len = (size_t)r * 4 * n1;
  std::vector<double> Z(len), V(len, 0.0), G(len), W(len), zero(len, 0.0);
  auto solve = [&](std::vector<double> &rhs, std::vector<double> &out) -> int {  // out = rhs Q^-1, refined once
    if (dpgo_agent_precondition(t, 0, zero.data(), rhs.data(), out.data())) return -1;
    if (dpgo_agent_hessvec(t, 0, zero.data(), out.data(), W.data())) return -1;   // (at X = 0: W = out Q)
    for (size_t i = 0; i < len; ++i) W[i] = rhs[i] - W[i];
    if (dpgo_agent_precondition(t, 0, zero.data(), W.data(), G.data())) return -1;
    for (size_t i = 0; i < len; ++i) out[i] += G[i];
    return 0;
  };
  do {
    if (dpgo_agent_add_measurements(t, 0, mm.data(), (int)mm.size()) < 0) break;
    if (dpgo_agent_num_poses(t, 0) != n1) break;  // (a pose no edge reaches: the dense path reports it as a failed pivot)
    double I34[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
    const int f0 = 0;
    if (dpgo_agent_update_neighbor_poses(t, 0, 1, 0, 1, &f0, I34) < 0) break;
    lap("add measurements");
    if (dpgo_agent_set_X(t, 0, zero.data())) break;
    lap("set_X (structure, Q, Q^-1)");
    if (dpgo_agent_build_problem(t, 0, 0)) break;
    if (dpgo_agent_get_G(t, 0, G.data())) break;
    for (size_t i = 0; i < len; ++i) V[i] = -G[i];
    lap("G");
    if (solve(V, Z)) break;
    lap("rotation solve");
    // rotations: pose 0 = I, pose i = the 3 x 3 block of Z (column c of the block = column c of R_i), projected to SO(3)
    std::vector<double> R((size_t)9 * n);
    for (int c = 0; c < 3; ++c) for (int a = 0; a < 3; ++a) R[3 * c + a] = (a == c) ? 1.0 : 0.0;
    for (int i = 1; i < n; ++i) std::memcpy(R.data() + (size_t)9 * i, Z.data() + (size_t)12 * (i - 1), sizeof(double) * 9);
    {
      DBuf dR;
      hipStream_t s = (hipStream_t)dpgo_team_stream(t);
      if (!dR.alloc(sizeof(double) * 9 * n)) break;
      if (hipMemcpyAsync(dR.p, R.data(), sizeof(double) * 9 * n, hipMemcpyHostToDevice, s) != hipSuccess) break;
      hipLaunchKernelGGL(k_project_so3, dim3((n + 63) / 64), dim3(64), 0, s, (double *)dR.p, n);
      if (hipMemcpyAsync(R.data(), dR.p, sizeof(double) * 9 * n, hipMemcpyDeviceToHost, s) != hipSuccess) break;
      if (hipStreamSynchronize(s) != hipSuccess) break;
    }
    std::memset(T, 0, sizeof(double) * 12 * (size_t)n);
    for (int i = 0; i < n; ++i) std::memcpy(T + (size_t)12 * i, R.data() + (size_t)9 * i, sizeof(double) * 9);
    // translations: minimise sum tau |t_j - t_i - R_i t~|^2, t_0 = 0 -- right-hand side on the translation entries
    std::fill(V.begin(), V.end(), 0.0);
    for (int e = 0; e < nm; ++e) {
      const int i = m[e].p1, j = m[e].p2;
      if (i == j) continue;
      const double tau = m[e].weight * m[e].tau;
      const double *Ri = T + (size_t)12 * i;
      for (int a = 0; a < 3; ++a) {
        double v = 0;
        for (int b = 0; b < 3; ++b) v += Ri[3 * b + a] * m[e].t[b];
        if (i != 0) V[((size_t)4 * (i - 1) + 3) * 3 + a] -= tau * v;
        if (j != 0) V[((size_t)4 * (j - 1) + 3) * 3 + a] += tau * v;
      }
    }
    lap("projection + rhs");
    if (solve(V, Z)) break;
    lap("translation solve");
    for (int i = 1; i < n; ++i) for (int a = 0; a < 3; ++a) T[(size_t)12 * i + 9 + a] = Z[((size_t)4 * (i - 1) + 3) * 3 + a];
    rc = DPGO_OK;
  } while (false);
  dpgo_team_destroy(t);
  lap("team_destroy");
  return rc;
}

extern "C" int dpgo_chordal_init(int device, const dpgo_measurement_t *m, int nm, int num_poses, double *T) {
  if (!m || nm < 0 || num_poses <= 0 || !T) return DPGO_ERR;
  for (int e = 0; e < nm; ++e)  // single-robot numbering: every endpoint inside [0, num_poses)
    if (m[e].p1 < 0 || m[e].p1 >= num_poses || m[e].p2 < 0 || m[e].p2 >= num_poses) return DPGO_ERR;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || hipSetDevice(device) != hipSuccess) return DPGO_ERR;
  {
    static const bool dense_only = std::getenv("DPGO_CHORDAL_DENSE") && std::getenv("DPGO_CHORDAL_DENSE")[0] == '1';
    // (needs at least one edge into pose 0 -- the pin is a shared edge -- and falls back to the dense path on any failure)
    if (!dense_only && num_poses >= 2 && num_poses <= DPGO_MAX_POSE_INDEX && chordal_via_team(device, m, nm, num_poses, T) == DPGO_OK)
      return DPGO_OK;
  }
  hipStream_t s;
  if (hipStreamCreate(&s) != hipSuccess) return DPGO_ERR;
  const int n = num_poses;
  int rc = DPGO_OK;
  const bool dup = has_parallel_edges(m, nm, n);
  std::memset(T, 0, sizeof(double) * 12 * (size_t)n);
  {
    // ---- stage 1: rotations.  Unknown X = [R_0 ... R_{n-1}] (3 x 3n), pose 0 pinned to I by a Dirichlet row.
    const int N = 3 * n;
    Assembler A(N, dup);
    std::vector<double> B((size_t)3 * N, 0.0);
    auto add = [&](int row, int col, double v) { A.add(row, col, v); };
    for (int a = 0; a < 3; ++a) { add(a, a, 1.0); B[(size_t)a * 3 + a] = 1.0; }
    for (int e = 0; e < nm; ++e) {
      const int i = m[e].p1, j = m[e].p2;
      const double k = m[e].weight * m[e].kappa;
      // k |R_j - R_i R~|^2:  A_ii += kI, A_jj += kI, A_ij += -k R~, A_ji += -k R~^T  (X A = B convention)
      if (i != 0) for (int a = 0; a < 3; ++a) add(3 * i + a, 3 * i + a, k);
      if (j != 0) for (int a = 0; a < 3; ++a) add(3 * j + a, 3 * j + a, k);
      if (i != 0 && j != 0) {
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) {
            add(3 * i + a, 3 * j + b, -k * m[e].R[3 * a + b]);
            add(3 * j + b, 3 * i + a, -k * m[e].R[3 * a + b]);
          }
      } else if (i == 0 && j != 0) {
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) B[(size_t)(3 * j + b) * 3 + a] += k * m[e].R[3 * a + b];
      } else if (j == 0 && i != 0) {
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) B[(size_t)(3 * i + b) * 3 + a] += k * m[e].R[3 * b + a];
      }
    }
    std::vector<double> X;
    if (dense_solve<3>(s, A.triplets(), N, B, X) != 0) rc = DPGO_ERR;
    if (rc == DPGO_OK) {
      DBuf dR;
      if (!dR.alloc(sizeof(double) * 9 * n)) rc = DPGO_ERR;
      else {
        (void)hipMemcpyAsync(dR.p, X.data(), sizeof(double) * 9 * n, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL(k_project_so3, dim3((n + 63) / 64), dim3(64), 0, s, (double *)dR.p, n);
        (void)hipMemcpyAsync(X.data(), dR.p, sizeof(double) * 9 * n, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) rc = DPGO_ERR;
        for (int i = 0; i < n; ++i) std::memcpy(T + (size_t)12 * i, X.data() + (size_t)9 * i, sizeof(double) * 9);
      }
    }
  }
  if (rc == DPGO_OK) {
    // ---- stage 2: translations, t_0 = 0
    Assembler A(n, dup);
    std::vector<double> B((size_t)3 * n, 0.0);
    A.add(0, 0, 1.0);
    for (int e = 0; e < nm; ++e) {
      const int i = m[e].p1, j = m[e].p2;
      const double tau = m[e].weight * m[e].tau;
      const double *Ri = T + (size_t)12 * i;
      double v[3];
      for (int a = 0; a < 3; ++a) { v[a] = 0; for (int b = 0; b < 3; ++b) v[a] += Ri[3 * b + a] * m[e].t[b]; }
      if (i != 0) { A.add(i, i, tau); for (int a = 0; a < 3; ++a) B[(size_t)i * 3 + a] -= tau * v[a]; }
      if (j != 0) { A.add(j, j, tau); for (int a = 0; a < 3; ++a) B[(size_t)j * 3 + a] += tau * v[a]; }
      if (i != 0 && j != 0) { A.add(i, j, -tau); A.add(j, i, -tau); }
    }
    std::vector<double> X;
    if (dense_solve<3>(s, A.triplets(), n, B, X) != 0) rc = DPGO_ERR;
    else for (int i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) T[(size_t)12 * i + 9 + a] = X[(size_t)i * 3 + a];
  }
  (void)hipStreamDestroy(s);
  return rc;
}
