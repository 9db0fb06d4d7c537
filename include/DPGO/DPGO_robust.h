// RobustCost: weight(residual) (src/PGOAgentROS.cpp:1050), computeErrorThresholdAtQuantile
// (src/PGOAgentROSNode.cpp:201).  GNC-TLS weight function of Yang et al. (RA-L 2020).
#pragma once
#include "DPGO_types.h"
#include "../dpgo_hip.h"

namespace DPGO {

class RobustCost {
 public:
  explicit RobustCost(const RobustCostParameters &p = RobustCostParameters()) : params_(p), mu_(p.GNCInitMu) {}
  double weight(double r) const {
    if (params_.costType == RobustCostParameters::Type::L2) return 1.0;
    const double r2 = r * r, b2 = params_.GNCBarc * params_.GNCBarc;
    const double upper = (mu_ + 1.0) / mu_ * b2, lower = mu_ / (mu_ + 1.0) * b2;
    if (r2 >= upper) return 0.0;
    if (r2 <= lower) return 1.0;
    return std::sqrt(b2 * mu_ * (mu_ + 1.0) / r2) - mu_;
  }
  void reset() { mu_ = params_.GNCInitMu; }
  void update() { mu_ *= params_.GNCMuStep; }
  double mu() const { return mu_; }
  static double computeErrorThresholdAtQuantile(double quantile, size_t dimension) {
    return dpgo_error_threshold_at_quantile(quantile, (int)dimension);
  }
 private:
  RobustCostParameters params_;
  double mu_;
};

}  // namespace DPGO
