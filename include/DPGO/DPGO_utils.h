// DPGO_utils.h -- free functions the wrapper calls: read_g2o_file (src/PGODatasetPublisherNode.cpp:80)
#pragma once
#include <stdexcept>
#include "RelativeSEMeasurement.h"

namespace DPGO {

inline std::vector<RelativeSEMeasurement> read_g2o_file(const std::string &filename, size_t &num_poses) {
  dpgo_measurement_t *raw = nullptr;
  int n = 0;
  const int nm = dpgo_read_g2o(filename.c_str(), DPGO_WEIGHT_LIBRARY, &raw, &n);
  if (nm < 0) throw std::runtime_error("read_g2o_file: cannot open " + filename);
  std::vector<RelativeSEMeasurement> out;
  out.reserve(nm);
  for (int k = 0; k < nm; ++k) out.push_back(RelativeSEMeasurement::fromC(raw[k]));
  dpgo_free(raw);
  num_poses = (size_t)n;
  return out;
}

}  // namespace DPGO
