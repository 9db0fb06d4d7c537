// frame_align.cpp -- robust inter-robot frame alignment for the initialisation path (SURVEY 8f-1;
// PGOAgent::initializeInGlobalFrame is fed by it when the first neighbour poses arrive,
// src/PGOAgentROS.cpp:1255-1284 -> updateNeighborPoses; parameter robustInitMinInliers, Node.cpp:150).
//
// Every shared loop closure with an initialised neighbour yields one candidate for T_world_robot.  With
// outlier loop closures the candidates disagree, so they are averaged robustly in two stages with
// graduated non-convexity / truncated least squares (Yang et al., RA-L 2020):
//   stage 1  single-rotation averaging under the chordal metric: R = proj_SO(3)(sum_i w_i R_i),
//            residual |R - R_i|_F, threshold = chordal length of `max_rotation_error_rad`;
//   stage 2  translation averaging over the rotation inliers: t = sum w_i t_i / sum w_i, residual |t - t_i|.
// Host arithmetic on <= a few hundred 3x4 matrices; nothing here belongs on the GPU.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/dpgo_hip.h"

namespace {

// cyclic Jacobi on a symmetric 3x3 (row/column-major agnostic): S = V diag(w) V^T
void jacobi3(double S[9], double w[3], double V[9]) {
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    const double off = S[1] * S[1] + S[2] * S[2] + S[5] * S[5];
    if (off < 1e-300) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = S[3 * p + q];
        if (std::fabs(apq) < 1e-300) continue;
        const double th = (S[3 * q + q] - S[3 * p + p]) / (2.0 * apq);
        const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {  // S <- S J
          const double skp = S[3 * k + p], skq = S[3 * k + q];
          S[3 * k + p] = c * skp - s * skq;
          S[3 * k + q] = s * skp + c * skq;
        }
        for (int k = 0; k < 3; ++k) {  // S <- J^T S
          const double spk = S[3 * p + k], sqk = S[3 * q + k];
          S[3 * p + k] = c * spk - s * sqk;
          S[3 * q + k] = s * spk + c * sqk;
        }
        for (int k = 0; k < 3; ++k) {
          const double vkp = V[3 * k + p], vkq = V[3 * k + q];
          V[3 * k + p] = c * vkp - s * vkq;
          V[3 * k + q] = s * vkp + c * vkq;
        }
      }
  }
  w[0] = S[0]; w[1] = S[4]; w[2] = S[8];
}

// nearest rotation to the column-major 3x3 A:  A (A^T A)^{-1/2} with the smallest singular direction flipped
// when det A < 0
void project_rotation(const double A[9], double out[9]) {
  double S[9], w[3], V[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += A[3 * i + k] * A[3 * j + k];  // (A^T A)_ij, columns of A are contiguous
      S[3 * i + j] = s;
    }
  jacobi3(S, w, V);
  const double det = A[0] * (A[4] * A[8] - A[7] * A[5]) - A[3] * (A[1] * A[8] - A[7] * A[2]) +
                     A[6] * (A[1] * A[5] - A[4] * A[2]);
  int kmin = 0;
  for (int k = 1; k < 3; ++k) if (w[k] < w[kmin]) kmin = k;
  double M[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) {
        const double sg = (k == kmin && det < 0) ? -1.0 : 1.0;
        s += sg * V[3 * i + k] * V[3 * j + k] / std::sqrt(std::fmax(w[k], 1e-300));
      }
      M[3 * i + j] = s;  // symmetric unless flipped; (i, j) = row i, column j
    }
  for (int j = 0; j < 3; ++j)
    for (int a = 0; a < 3; ++a) {
      double s = 0;
      for (int i = 0; i < 3; ++i) s += A[3 * i + a] * M[3 * i + j];
      out[3 * j + a] = s;
    }
}

double tls_weight(double r2, double mu, double barc2) {
  if (r2 >= (mu + 1.0) / mu * barc2) return 0.0;
  if (r2 <= mu / (mu + 1.0) * barc2) return 1.0;
  return std::sqrt(barc2 * mu * (mu + 1.0) / r2) - mu;
}

// GNC-TLS around a weighted-mean solver.  solve(w) updates the estimate, res2(i) returns the squared
// residual of candidate i against it.  Returns false when every weight vanishes.
template <class Solve, class Res2>
bool gnc_tls(int n, double barc, std::vector<double> &w, Solve solve, Res2 res2) {
  const double barc2 = barc * barc;
  if (!solve(w)) return false;
  double rmax2 = 0;
  for (int i = 0; i < n; ++i) if (w[i] > 0) rmax2 = std::fmax(rmax2, res2(i));
  double mu = barc2 / (2.0 * rmax2 - barc2);
  if (!(mu > 0)) return true;  // every residual is already below the threshold: plain mean
  for (int it = 0; it < 1000; ++it) {
    bool binary = true;
    std::vector<double> wn(n, 0.0);
    for (int i = 0; i < n; ++i) {
      if (w[i] < 0) { wn[i] = -1; continue; }  // excluded candidate
      wn[i] = tls_weight(res2(i), mu, barc2);
      if (wn[i] > 0 && wn[i] < 1) binary = false;
    }
    std::vector<double> keep = w;
    w = wn;
    if (!solve(w)) { w = keep; return false; }
    if (binary) break;
    mu *= 1.4;
  }
  return true;
}

}  // namespace

extern "C" int dpgo_robust_frame_alignment(const double *Tc, int n, double max_rotation_error_rad,
                                           double max_translation_error, int min_inliers, double *T_out,
                                           int *inlier) {
  if (n <= 0) return DPGO_NOT_READY;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
  // ---- stage 1: rotations
  std::vector<double> w(n, 1.0);
  auto solveR = [&](const std::vector<double> &ww) {
    double A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, sw = 0;
    for (int i = 0; i < n; ++i) {
      if (ww[i] <= 0) continue;
      sw += ww[i];
      for (int e = 0; e < 9; ++e) A[e] += ww[i] * Tc[(size_t)12 * i + e];
    }
    if (!(sw > 0)) return false;
    project_rotation(A, R);
    return true;
  };
  auto res2R = [&](int i) {
    double s = 0;
    for (int e = 0; e < 9; ++e) { const double d = R[e] - Tc[(size_t)12 * i + e]; s += d * d; }
    return s;
  };
  const double chordal = 2.0 * std::sqrt(2.0) * std::sin(0.5 * max_rotation_error_rad);
  if (!gnc_tls(n, chordal, w, solveR, res2R)) return DPGO_NOT_READY;
  int nin = 0;
  std::vector<double> wt(n, -1.0);  // stage 2 runs on the rotation inliers only
  for (int i = 0; i < n; ++i) if (w[i] > 0.5) { wt[i] = 1.0; ++nin; }
  if (nin < min_inliers) return DPGO_NOT_READY;
  solveR(wt);  // final rotation = plain chordal mean of the inliers
  // ---- stage 2: translations
  auto solveT = [&](const std::vector<double> &ww) {
    double s[3] = {0, 0, 0}, sw = 0;
    for (int i = 0; i < n; ++i) {
      if (ww[i] <= 0) continue;
      sw += ww[i];
      for (int a = 0; a < 3; ++a) s[a] += ww[i] * Tc[(size_t)12 * i + 9 + a];
    }
    if (!(sw > 0)) return false;
    for (int a = 0; a < 3; ++a) t[a] = s[a] / sw;
    return true;
  };
  auto res2T = [&](int i) {
    double s = 0;
    for (int a = 0; a < 3; ++a) { const double d = t[a] - Tc[(size_t)12 * i + 9 + a]; s += d * d; }
    return s;
  };
  if (!gnc_tls(n, max_translation_error, wt, solveT, res2T)) return DPGO_NOT_READY;
  nin = 0;
  for (int i = 0; i < n; ++i) {
    const int in = wt[i] > 0.5;
    if (inlier) inlier[i] = in;
    nin += in;
    wt[i] = in ? 1.0 : -1.0;
  }
  if (nin < min_inliers) return DPGO_NOT_READY;
  solveT(wt);
  solveR(wt);  // both factors from the final inlier set
  std::memcpy(T_out, R, sizeof(R));
  std::memcpy(T_out + 9, t, sizeof(t));
  return DPGO_OK;
}
