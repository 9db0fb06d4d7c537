// PoseGraph(id, r, d): the measurement container the wrapper queries and mutates
// (src/PGOAgentROS.cpp:137,237,268-280,343-345,706,725,770,800,1048,1058,1351,1394,1431,1445).
// The data matrices themselves (Q, G, dense preconditioner) live in HBM behind the C-ABI;
// clearDataMatrices() only raises the flag PGOAgent consumes at its next iterate().
#pragma once
#include <algorithm>
#include <set>
#include "RelativeSEMeasurement.h"

namespace DPGO {

class PoseGraph {
 public:
  struct Statistics {
    double total_loop_closures = 0, accept_loop_closures = 0, reject_loop_closures = 0, undecided_loop_closures = 0;
  };
  PoseGraph(unsigned id, unsigned r, unsigned d) : id_(id), r_(r), d_(d) {}
  unsigned id() const { return id_; }
  unsigned r() const { return r_; }
  unsigned d() const { return d_; }
  unsigned n() const { return n_; }
  size_t numOdometry() const { return odometry_.size(); }
  size_t numPrivateLoopClosures() const { return private_lcs_.size(); }
  size_t numSharedLoopClosures() const { return shared_lcs_.size(); }
  size_t numMeasurements() const { return numOdometry() + numPrivateLoopClosures() + numSharedLoopClosures(); }
  bool hasMeasurement(const PoseID &src, const PoseID &dst) const { return edge_ids_.count(key(src, dst)) != 0; }
  void addMeasurement(const RelativeSEMeasurement &m) {
    const PoseID src((unsigned)m.r1, (unsigned)m.p1), dst((unsigned)m.r2, (unsigned)m.p2);
    if (hasMeasurement(src, dst)) return;
    if (m.r1 != id_ && m.r2 != id_) return;
    edge_ids_.insert(key(src, dst));
    if (m.r1 == id_ && m.r2 == id_) {
      (m.p1 + 1 == m.p2 ? odometry_ : private_lcs_).push_back(m);
      n_ = std::max<unsigned>(n_, (unsigned)std::max(m.p1, m.p2) + 1);
    } else {
      shared_lcs_.push_back(m);
      const bool out = (m.r1 == id_);
      n_ = std::max<unsigned>(n_, (unsigned)(out ? m.p1 : m.p2) + 1);
      nbr_ids_.insert((unsigned)(out ? m.r2 : m.r1));
      nbr_public_.insert(out ? dst : src);
    }
    dirty_structure_ = true;
  }
  const std::vector<RelativeSEMeasurement> &odometry() const { return odometry_; }
  const std::vector<RelativeSEMeasurement> &privateLoopClosures() const { return private_lcs_; }
  const std::vector<RelativeSEMeasurement> &sharedLoopClosures() const { return shared_lcs_; }
  std::set<unsigned> activeNeighborIDs() const { return nbr_ids_; }
  std::set<PoseID, ComparePoseID> activeNeighborPublicPoseIDs() const { return nbr_public_; }
  // mutable pointers: the wrapper writes m->weight / m->fixedWeight in place (:1053-1054, 1451)
  std::vector<RelativeSEMeasurement *> activeLoopClosures() {
    std::vector<RelativeSEMeasurement *> out;
    for (auto &m : private_lcs_) out.push_back(&m);
    for (auto &m : shared_lcs_) out.push_back(&m);
    return out;
  }
  std::vector<RelativeSEMeasurement *> inactiveLoopClosures() { return {}; }
  std::vector<RelativeSEMeasurement *> allMeasurements() {
    std::vector<RelativeSEMeasurement *> out;
    for (auto &m : odometry_) out.push_back(&m);
    for (auto &m : private_lcs_) out.push_back(&m);
    for (auto &m : shared_lcs_) out.push_back(&m);
    return out;
  }
  RelativeSEMeasurement *findMeasurement(const PoseID &src, const PoseID &dst) {
    for (auto *m : allMeasurements())
      if (m->r1 == src.robot_id && m->p1 == src.frame_id && m->r2 == dst.robot_id && m->p2 == dst.frame_id) return m;
    return nullptr;
  }
  Statistics statistics() const {
    Statistics s;
    for (const auto *v : {&private_lcs_, &shared_lcs_})
      for (const auto &m : *v) {
        s.total_loop_closures += 1;
        if (m.weight == 1) s.accept_loop_closures += 1;
        else if (m.weight == 0) s.reject_loop_closures += 1;
        else s.undecided_loop_closures += 1;
      }
    return s;
  }
  void clearDataMatrices() { dirty_data_ = true; }
  bool takeDirtyData() { const bool v = dirty_data_; dirty_data_ = false; return v; }
  bool takeDirtyStructure() { const bool v = dirty_structure_; dirty_structure_ = false; return v; }
 private:
  typedef std::pair<std::pair<unsigned, unsigned>, std::pair<unsigned, unsigned>> Key;
  static Key key(const PoseID &s, const PoseID &d) { return {{s.robot_id, s.frame_id}, {d.robot_id, d.frame_id}}; }
  unsigned id_, r_, d_, n_ = 0;
  std::vector<RelativeSEMeasurement> odometry_, private_lcs_, shared_lcs_;
  std::set<Key> edge_ids_;
  std::set<unsigned> nbr_ids_;
  std::set<PoseID, ComparePoseID> nbr_public_;
  bool dirty_data_ = false, dirty_structure_ = false;
};

}  // namespace DPGO
