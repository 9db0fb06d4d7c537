// PGOLogger::loadMeasurements(file, bool)  (src/PGODatasetPublisherNode.cpp:168) and the writers that
// round-trip through it: logMeasurements, logTrajectory (files land in the logger's directory).
#pragma once
#include <stdexcept>
#include "RelativeSEMeasurement.h"

namespace DPGO {

class PGOLogger {
 public:
  explicit PGOLogger(std::string dir = "") : dir_(std::move(dir)) {}
  // load_weight == false: the wrapper path, weights/inlier flags are re-derived downstream
  static std::vector<RelativeSEMeasurement> loadMeasurements(const std::string &filename, bool load_weight = false) {
    dpgo_measurement_t *raw = nullptr;
    const int nm = dpgo_read_measurements_csv(filename.c_str(), DPGO_WEIGHT_LIBRARY, &raw);
    if (nm < 0) throw std::runtime_error("loadMeasurements: cannot open " + filename);
    std::vector<RelativeSEMeasurement> out;
    for (int k = 0; k < nm; ++k) {
      RelativeSEMeasurement m = RelativeSEMeasurement::fromC(raw[k]);
      if (!load_weight) { m.weight = 1.0; m.fixedWeight = false; }
      out.push_back(m);
    }
    dpgo_free(raw);
    return out;
  }
  // measurement list (weights and inlier flags included) in the loadMeasurements format
  bool logMeasurements(const std::vector<RelativeSEMeasurement> &measurements, const std::string &filename) const {
    std::vector<dpgo_measurement_t> raw;
    raw.reserve(measurements.size());
    for (const auto &m : measurements) raw.push_back(m.toC());
    return dpgo_write_measurements_csv((dir_ + filename).c_str(), raw.data(), (int)raw.size()) >= 0;
  }
  // T = d x (d+1)n trajectory [R_0 t_0 | R_1 t_1 | ...] as returned by getTrajectoryInGlobalFrame
  bool logTrajectory(unsigned d, unsigned n, const Matrix &T, const std::string &filename) const {
    if (d != 3) return false;
    std::vector<double> flat((size_t)12 * n);
    for (unsigned i = 0; i < n; ++i)
      for (unsigned c = 0; c < 4; ++c)
        for (unsigned a = 0; a < 3; ++a) flat[(size_t)12 * i + 3 * c + a] = T(a, 4 * i + c);
    return dpgo_write_trajectory_csv((dir_ + filename).c_str(), flat.data(), (int)n) >= 0;
  }
 private:
  std::string dir_;
};

}  // namespace DPGO
