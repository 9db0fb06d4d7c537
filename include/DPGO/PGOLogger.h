// PGOLogger::loadMeasurements(file, bool)  (src/PGODatasetPublisherNode.cpp:168)
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
 private:
  std::string dir_;
};

}  // namespace DPGO
