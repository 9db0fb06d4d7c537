// RobustCost: weight(residual) (src/PGOAgentROS.cpp:1050), computeErrorThresholdAtQuantile
// (src/PGOAgentROSNode.cpp:201).  L2 / L1 / Huber / TLS / GM and the GNC-TLS weight function of Yang et al. (RA-L 2020).
#pragma once
#include "DPGO_types.h"
#include "../dpgo_hip.h"

namespace DPGO {

class RobustCost {
 public:
  explicit RobustCost(const RobustCostParameters &p = RobustCostParameters()) : params_(p), mu_(p.GNCInitMu) {}
  double weight(double r) const {
    // the six types of src/PGOAgentROSNode.cpp:178-188 (formulas: include/dpgo_hip.h, DPGO_COST_*)
    switch (params_.costType) {
      case RobustCostParameters::Type::L2: return 1.0;
      case RobustCostParameters::Type::L1: return 1.0 / r;
      case RobustCostParameters::Type::Huber: return r < params_.HuberThreshold ? 1.0 : params_.HuberThreshold / r;
      case RobustCostParameters::Type::TLS: return r < params_.TLSThreshold ? 1.0 : 0.0;
      case RobustCostParameters::Type::GM: { const double a = 1.0 + r * r; return 1.0 / (a * a); }
      default: break;  // GNC_TLS
    }
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
