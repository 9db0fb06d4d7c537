// RelativeSEMeasurement(r1, r2, p1, p2, R, t, kappa, tau) with fields r1 r2 p1 p2 R t kappa tau weight
// fixedWeight  (src/utils.cpp:109-149; src/PGOAgentROS.cpp:740-745; src/PGODatasetPublisherNode.cpp:118-119)
#pragma once
#include "DPGO_types.h"
#include "../dpgo_hip.h"

namespace DPGO {

struct RelativeSEMeasurement {
  size_t r1 = 0, r2 = 0, p1 = 0, p2 = 0;
  Matrix R, t;
  double kappa = 0, tau = 0;
  bool fixedWeight = false;
  double weight = 1.0;
  RelativeSEMeasurement() = default;
  RelativeSEMeasurement(size_t r1_, size_t r2_, size_t p1_, size_t p2_, const Matrix &R_, const Matrix &t_, double kappa_,
                        double tau_)
      : r1(r1_), r2(r2_), p1(p1_), p2(p2_), R(R_), t(t_), kappa(kappa_), tau(tau_) {}

  dpgo_measurement_t toC() const {
    dpgo_measurement_t m;
    std::memset(&m, 0, sizeof m);
    m.r1 = (int)r1; m.p1 = (int)p1; m.r2 = (int)r2; m.p2 = (int)p2;
    for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) m.R[3 * a + b] = R(a, b); m.t[a] = t(a, 0); }
    m.kappa = kappa; m.tau = tau; m.weight = weight; m.fixed_weight = fixedWeight ? 1 : 0;
    return m;
  }
  static RelativeSEMeasurement fromC(const dpgo_measurement_t &m) {
    Matrix R = Matrix::Zero(3, 3), t = Matrix::Zero(3, 1);
    for (int a = 0; a < 3; ++a) { for (int b = 0; b < 3; ++b) R(a, b) = m.R[3 * a + b]; t(a, 0) = m.t[a]; }
    RelativeSEMeasurement out((size_t)m.r1, (size_t)m.r2, (size_t)m.p1, (size_t)m.p2, R, t, m.kappa, m.tau);
    out.weight = m.weight; out.fixedWeight = m.fixed_weight != 0;
    return out;
  }
};

}  // namespace DPGO
