// PGOAgent.h -- C++ facade `DPGO::PGOAgent` over the C-ABI of libdpgo_hip.so.
//
// Reproduces the surface src/PGOAgentROS.cpp consumes: the public methods with
// their bool / out-parameter conventions AND the protected members PGOAgentROS reads and writes
// directly (`class PGOAgentROS : public PGOAgent`, include/dpgo_ros/PGOAgentROS.h:121).  All
// optimisation state lives in HBM behind the C-ABI; this class only keeps the host-side mirror the
// wrapper touches (measurements with their mutable weights, neighbour pose dictionary, statuses).
// Declared simplifications ("next"): local initialisation is odometry chaining or the GPU chordal
// relaxation or the GPU single-robot GNC-TLS solve (dpgo_robust_local_init);
// the inter-robot frame alignment uses the first shared loop closure whose neighbour pose is known
// (L2 cost) or GNC-TLS two-stage averaging over all of them (robust cost); robot 0 draws a fixed, not random, YLift.
#pragma once
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <iostream>
#include <memory>
#include <mutex>
#include <random>
#include <set>
#include <thread>

#include "DPGO_robust.h"
#include "DPGO_types.h"
#include "DPGO_utils.h"
#include "PGOLogger.h"
#include "PoseGraph.h"

namespace DPGO {

class PGOAgentParameters {  // fields written by src/PGOAgentROSNode.cpp:80-231
 public:
  unsigned d, r, numRobots;
  ROptParameters localOptimizationParams;
  bool asynchronous = false;
  double asynchronousOptimizationRate = 1.0;
  InitializationMethod localInitializationMethod = InitializationMethod::Odometry;
  bool multirobotInitialization = true;
  bool acceleration = false;
  unsigned restartInterval = 30;
  RobustCostParameters robustCostParams;
  unsigned robustOptNumWeightUpdates = 4, robustOptNumResets = 0, robustOptInnerIters = 30;
  double robustOptMinConvergenceRatio = 0.8;
  unsigned robustInitMinInliers = 2;
  double robustInitMaxRotationError = 0.5;     // rad: candidates further than this from the consensus are outliers
  double robustInitMaxTranslationError = 1.0;  // same unit as the measurements' translations
  unsigned maxNumIters = 1000;
  double relChangeTol = 5e-3;
  bool verbose = false, logData = false;
  std::string logDirectory;
  PGOAgentParameters(unsigned dIn, unsigned rIn, unsigned numRobotsIn) : d(dIn), r(rIn), numRobots(numRobotsIn) {}
  inline friend std::ostream &operator<<(std::ostream &os, const PGOAgentParameters &p) {
    os << "PGOAgent parameters: d=" << p.d << " r=" << p.r << " robots=" << p.numRobots << " accel=" << p.acceleration
       << " restart=" << p.restartInterval << " relChangeTol=" << p.relChangeTol << std::endl;
    return os;
  }
};

#define DPGO_CHECK(cond)                                                                       \
  do {                                                                                         \
    if (!(cond)) { std::fprintf(stderr, "DPGO CHECK failed: %s (%s:%d): %s\n", #cond, __FILE__, __LINE__, dpgo_last_error()); std::abort(); } \
  } while (0)

class PGOAgent {
 public:
  PGOAgent(unsigned ID, const PGOAgentParameters &params)
      : mID(ID), d(params.d), r(params.r), mParams(params), mState(PGOAgentState::WAIT_FOR_DATA), mStatus(ID),
        mPoseGraph(std::make_shared<PoseGraph>(ID, params.r, params.d)), mRobustCost(params.robustCostParams),
        mTeamRobotActive(params.numRobots, false) {
    if (mID == 0) {  // robot 0 owns the lifting matrix (src/PGOAgentROS.cpp:404)
      Matrix Y = Matrix::Zero(r, d);
      std::vector<double> tmp(3 * r);
      dpgo_fixed_stiefel((int)r, tmp.data());
      for (unsigned c = 0; c < d; ++c) for (unsigned a = 0; a < r; ++a) Y(a, c) = tmp[c * r + a];
      YLift.emplace(Y);
    }
    mTeamRobotActive[mID] = true;
  }
  virtual ~PGOAgent() { endOptimizationLoop(); destroyTeam(); }
  PGOAgent(const PGOAgent &) = delete;
  PGOAgent &operator=(const PGOAgent &) = delete;

  // ---- identity / counters
  unsigned getID() const { return mID; }
  unsigned num_poses() const { return mPoseGraph->n(); }
  unsigned dimension() const { return d; }
  unsigned relaxation_rank() const { return r; }
  unsigned instance_number() const { return mInstanceNumber; }
  unsigned iteration_number() const { return mIterationNumber; }
  PGOAgentParameters getParams() const { return mParams; }
  std::vector<unsigned> getNeighbors() const {
    const auto s = mPoseGraph->activeNeighborIDs();
    return std::vector<unsigned>(s.begin(), s.end());
  }

  // ---- measurements
  void addMeasurement(const RelativeSEMeasurement &m) {
    std::lock_guard<std::recursive_mutex> lock_(mMutex);
  // :277, :1307
    if (mState != PGOAgentState::WAIT_FOR_DATA && mState != PGOAgentState::WAIT_FOR_INITIALIZATION && mPoseGraph->hasMeasurement(PoseID(m.r1, m.p1), PoseID(m.r2, m.p2))) return;
    mPoseGraph->addMeasurement(m);
  }

  // ---- lifting matrix / anchor
  bool getLiftingMatrix(Matrix &M) const { if (!YLift) return false; M = *YLift; return true; }  // :404
  void setLiftingMatrix(const Matrix &M) { YLift.emplace(M); }                                    // :928
  void setGlobalAnchor(const Matrix &M) { globalAnchor.emplace(LiftedPose(M)); }                  // :939, :1466

  // ---- initialisation (:348, :353-360)
  void initialize(const PoseArray *TInitPtr = nullptr) {
    std::lock_guard<std::recursive_mutex> lock_(mMutex);

    if (mPoseGraph->n() == 0) return;
    const unsigned n = mPoseGraph->n();
    PoseArray T(d, n);
    if (TInitPtr && TInitPtr->n() == n) T = *TInitPtr;
    else {
      std::vector<dpgo_measurement_t> odo;
      for (const auto &m : mPoseGraph->odometry()) { dpgo_measurement_t c = m.toC(); c.r1 = c.r2 = 0; odo.push_back(c); }
      std::vector<double> Tl(12 * (size_t)n);
      bool done = false;
      if (mParams.localInitializationMethod == InitializationMethod::Chordal) {
        // chordal relaxation of the local graph (odometry + private loop closures) on the GPU
        std::vector<dpgo_measurement_t> loc = odo;
        for (const auto &m : mPoseGraph->privateLoopClosures()) { dpgo_measurement_t c = m.toC(); c.r1 = c.r2 = 0; loc.push_back(c); }
        done = dpgo_chordal_init(0, loc.data(), (int)loc.size(), (int)n, Tl.data()) == DPGO_OK;
      }
      if (mParams.localInitializationMethod == InitializationMethod::GNC_TLS) {
        // robust single-robot solve on the GPU: odometry fixed, private loop closures re-weighted by GNC-TLS
        std::vector<dpgo_measurement_t> loc = odo;
        for (const auto &m : mPoseGraph->privateLoopClosures()) { dpgo_measurement_t c = m.toC(); c.r1 = c.r2 = 0; loc.push_back(c); }
        dpgo_params_t c = toC(mParams);
        done = dpgo_robust_local_init(0, loc.data(), (int)loc.size(), (int)n, &c, Tl.data(), nullptr) == DPGO_OK;
      }
      if (!done) dpgo_odometry_init(odo.data(), (int)odo.size(), (int)n, Tl.data());
      Matrix M = Matrix::Zero(d, (d + 1) * n);
      for (unsigned i = 0; i < n; ++i) for (unsigned c = 0; c < 4; ++c) for (unsigned b = 0; b < 3; ++b) M(b, 4 * i + c) = Tl[12 * i + 3 * c + b];
      T.setData(M);
    }
    TLocalInit.emplace(T);
    mState = PGOAgentState::WAIT_FOR_INITIALIZATION;
  }
  void initializeInGlobalFrame(const Pose &T_world_robot) {
    std::lock_guard<std::recursive_mutex> lock_(mMutex);

    DPGO_CHECK(YLift.has_value());
    if (!TLocalInit) initialize();
    DPGO_CHECK(TLocalInit.has_value());
    const unsigned n = mPoseGraph->n();
    Matrix X = Matrix::Zero(r, (d + 1) * n);
    for (unsigned i = 0; i < n; ++i) {
      const Pose Ti = T_world_robot * Pose(TLocalInit->pose(i));
      mat_set_block(X, 0, i * (d + 1), (*YLift) * Ti.getData());
    }
    ensureTeam();
    DPGO_CHECK(dpgo_agent_set_X(team_, (int)mID, X.data()) == DPGO_OK);
    mState = PGOAgentState::INITIALIZED;
    pushNeighborDict(false);
    pushNeighborDict(true);
    if (mParams.asynchronous) startOptimizationLoop(mParams.asynchronousOptimizationRate);
  }
  void anchorFirstPose() {}  // :360 -- gauge fixing of the single-robot case; no-op for the lifted iterate
  bool isRobotInitialized(unsigned id) const {  // :451,468,1144
    if (id == mID) return mState == PGOAgentState::INITIALIZED;
    auto it = mTeamStatus.find(id);
    return it != mTeamStatus.end() && it->second.state == PGOAgentState::INITIALIZED;
  }

  // ---- public pose exchange (:424, :666-668, :1276-1278)
  bool getSharedPose(unsigned index, Matrix &Mout) {
    std::lock_guard<std::recursive_mutex> lock_(mMutex);

    if (mState != PGOAgentState::INITIALIZED || index >= num_poses()) return false;
    const Matrix X = fetchX(0);
    Mout = mat_block(X, 0, index * (d + 1), r, d + 1);
    return true;
  }
  bool getSharedPoseDictWithNeighbor(PoseDict &map, unsigned neighborID) { return sharedDict(map, neighborID, 0); }
  bool getAuxSharedPoseDictWithNeighbor(PoseDict &map, unsigned neighborID) { return sharedDict(map, neighborID, 1); }
  void updateNeighborPoses(unsigned neighborID, const PoseDict &poseDict) {
    std::lock_guard<std::recursive_mutex> lock_(mMutex);

    for (const auto &kv : poseDict) neighborPoseDict[kv.first] = kv.second;
    if (mState == PGOAgentState::WAIT_FOR_INITIALIZATION && TLocalInit && YLift && mParams.multirobotInitialization)
      tryAlignWithNeighbor(neighborID, poseDict);
    if (mState == PGOAgentState::INITIALIZED) pushDict(neighborID, poseDict, false);
  }
  void updateAuxNeighborPoses(unsigned neighborID, const PoseDict &poseDict) {
    std::lock_guard<std::recursive_mutex> lock_(mMutex);

    for (const auto &kv : poseDict) neighborAuxPoseDict[kv.first] = kv.second;
    if (mState == PGOAgentState::INITIALIZED) pushDict(neighborID, poseDict, true);
  }

  // ---- the hot call (:160, :1185)
  bool iterate(bool doOptimization = true) {
    std::lock_guard<std::recursive_mutex> lock_(mMutex);

    if (mState != PGOAgentState::INITIALIZED) { mIterationNumber++; return false; }
    syncMeasurements();
    if ((int)mIterationNumber != dpgo_agent_iteration_number(team_, (int)mID))
      dpgo_agent_set_iteration_number(team_, (int)mID, (int)mIterationNumber);
    const int rc = dpgo_agent_iterate(team_, (int)mID, doOptimization ? 1 : 0);
    DPGO_CHECK(rc >= 0);
    mIterationNumber++;
    if (mParams.robustCostParams.costType != RobustCostParameters::Type::L2) mRobustOptInnerIter++;
    if (dpgo_agent_publish_requested(team_, (int)mID, 1)) mPublishPublicPosesRequested = true;
    dpgo_status_t s;
    DPGO_CHECK(dpgo_agent_get_status(team_, (int)mID, &s) == DPGO_OK);
    mStatus = PGOAgentStatus(mID, mState, mInstanceNumber, mIterationNumber, s.ready_to_terminate != 0, s.relative_change);
    if (doOptimization && rc == DPGO_OK) {
      dpgo_opt_result_t o;
      DPGO_CHECK(dpgo_agent_get_opt_result(team_, (int)mID, &o) == DPGO_OK);
      mLocalOptResult.success = o.success != 0;
      mLocalOptResult.fInit = o.f_init; mLocalOptResult.fOpt = o.f_opt;
      mLocalOptResult.gradNormInit = o.gradnorm_init; mLocalOptResult.gradNormOpt = o.gradnorm_opt;
    }
    return rc == DPGO_OK;
  }
  virtual void reset() {  // :223 (overridden by PGOAgentROS::reset, which calls this first)
    endOptimizationLoop();
    std::lock_guard<std::recursive_mutex> lock_(mMutex);
    destroyTeam();
    mInstanceNumber++;
    mIterationNumber = 0; mWeightUpdateCount = 0; mRobustOptInnerIter = 0;
    mState = PGOAgentState::WAIT_FOR_DATA;
    mStatus = PGOAgentStatus(mID, mState, mInstanceNumber, 0, false, 0);
    mPoseGraph = std::make_shared<PoseGraph>(mID, r, d);
    mTeamStatus.clear();
    neighborPoseDict.clear(); neighborAuxPoseDict.clear();
    TLocalInit.reset(); globalAnchor.reset();
    mRobustCost.reset();
    mPublishPublicPosesRequested = false; mPublishAsynchronousRequested = false;
    std::fill(mTeamRobotActive.begin(), mTeamRobotActive.end(), false);
    mTeamRobotActive[mID] = true;
    pushed_ = {0, 0, 0};
  }

  // ---- asynchronous (ASAPP) mode: a library-owned thread calls iterate(true) at exponentially distributed
  // intervals of mean 1/freq and raises mPublishAsynchronousRequested (src/PGOAgentROS.cpp:119-127); every entry
  // point of this class serialises on mMutex, so the ROS thread may call update*NeighborPoses / get*SharedPoseDict
  // / getStatus concurrently (SURVEY 3d, 8b "Threading")
  void startOptimizationLoop(double freq) {
    if (mOptimizationThread) return;
    mEndLoopRequested = false;
    mOptimizationThread.reset(new std::thread([this, freq]() {
      std::mt19937 rng(1234u + mID);
      std::exponential_distribution<double> gap(freq);
      while (!mEndLoopRequested) {
        std::this_thread::sleep_for(std::chrono::duration<double>(gap(rng)));
        if (mEndLoopRequested) break;
        std::lock_guard<std::recursive_mutex> lock_(mMutex);
        if (mState == PGOAgentState::INITIALIZED && iterate(true)) mPublishAsynchronousRequested = true;
      }
    }));
  }
  void endOptimizationLoop() {
    if (!mOptimizationThread) return;
    mEndLoopRequested = true;
    mOptimizationThread->join();
    mOptimizationThread.reset();
  }
  bool isOptimizationRunning() const { return (bool)mOptimizationThread; }

  // ---- status / team bookkeeping (:616, :965, :1116-1121, :208-210)
  PGOAgentStatus getStatus() { mStatus.agentID = mID; mStatus.state = mState; mStatus.instanceNumber = mInstanceNumber; mStatus.iterationNumber = mIterationNumber; return mStatus; }
  void setNeighborStatus(const PGOAgentStatus &s) { mTeamStatus[s.agentID] = s; }
  bool hasNeighborStatus(unsigned id) const { return mTeamStatus.count(id) != 0; }
  PGOAgentStatus getNeighborStatus(unsigned id) const { return mTeamStatus.at(id); }
  bool isRobotActive(unsigned id) const { return id < mTeamRobotActive.size() && mTeamRobotActive[id]; }
  void setRobotActive(unsigned id, bool active = true) { if (id < mTeamRobotActive.size()) mTeamRobotActive[id] = active; }
  size_t numActiveRobots() const { size_t c = 0; for (bool b : mTeamRobotActive) c += b; return c; }
  bool shouldTerminate() {
    if (mIterationNumber > mParams.maxNumIters) return true;
    for (unsigned id = 0; id < mParams.numRobots; ++id) {
      if (!isRobotActive(id)) continue;
      PGOAgentStatus s = (id == mID) ? getStatus() : (hasNeighborStatus(id) ? getNeighborStatus(id) : PGOAgentStatus(id));
      if (s.state != PGOAgentState::INITIALIZED || !s.readyToTerminate) return false;
    }
    if (mParams.robustCostParams.costType != RobustCostParameters::Type::L2 &&
        mWeightUpdateCount < (int)mParams.robustOptNumWeightUpdates) return false;
    return true;
  }
  bool shouldUpdateMeasurementWeights() const {
    if (mParams.robustCostParams.costType == RobustCostParameters::Type::L2) return false;
    if (mWeightUpdateCount >= (int)mParams.robustOptNumWeightUpdates) return false;
    return mRobustOptInnerIter >= (int)mParams.robustOptInnerIters;
  }

  // ---- robust path (:1218, :1049, :1341)
  bool computeMeasurementResidual(const RelativeSEMeasurement &m, double *residual) {
    std::lock_guard<std::recursive_mutex> lock_(mMutex);

    if (mState != PGOAgentState::INITIALIZED) return false;
    syncMeasurements();
    const dpgo_measurement_t c = m.toC();
    return dpgo_agent_compute_residual(team_, (int)mID, &c, residual) == DPGO_OK;
  }
  void updateMeasurementWeights() {
    std::lock_guard<std::recursive_mutex> lock_(mMutex);

    if (mState == PGOAgentState::INITIALIZED) {
      // one residual launch and one copy for all measurements (stored order = odometry, private, shared)
      syncMeasurements();
      const auto all = mPoseGraph->allMeasurements();
      std::vector<double> res(all.size());
      std::vector<int> ok(all.size());
      const int cnt = dpgo_agent_compute_residuals(team_, (int)mID, res.data(), ok.data());
      DPGO_CHECK(cnt == (int)all.size());
      for (size_t k = mPoseGraph->numOdometry(); k < all.size(); ++k) {
        RelativeSEMeasurement *m = all[k];
        if (m->fixedWeight || !ok[k]) continue;
        if (m->r1 != m->r2) { const unsigned other = (m->r1 == mID) ? m->r2 : m->r1; if (other < mID) continue; }
        m->weight = mRobustCost.weight(res[k]);
      }
    }
    mWeightUpdateCount++;
    mRobustCost.update();
    mRobustOptInnerIter = 0;
    mPoseGraph->clearDataMatrices();
    if (mParams.acceleration && mState == PGOAgentState::INITIALIZED) {
      // the Nesterov sequences restart from X under the new weights (V = Y = X, gamma = alpha = 0)
      syncMeasurements();
      DPGO_CHECK(dpgo_agent_reset_acceleration(team_, (int)mID) == DPGO_OK);
    }
  }
  bool setMeasurementWeight(const PoseID &src, const PoseID &dst, double weight, bool fixed_weight = false) {
    std::lock_guard<std::recursive_mutex> lock_(mMutex);

    RelativeSEMeasurement *m = mPoseGraph->findMeasurement(src, dst);
    if (!m) return false;
    m->weight = weight; m->fixedWeight = fixed_weight;
    return true;
  }

  // ---- rounding (:622-627, :774-812, :1395)
  bool getTrajectoryInGlobalFrame(PoseArray &Trajectory) {
    std::lock_guard<std::recursive_mutex> lock_(mMutex);

    if (!globalAnchor || mState != PGOAgentState::INITIALIZED) return false;
    const Matrix X = fetchX(0);
    Trajectory = PoseArray(d, num_poses());
    for (unsigned i = 0; i < num_poses(); ++i) Trajectory.setPose(i, roundPose(mat_block(X, 0, i * (d + 1), r, d + 1)));
    return true;
  }
  bool getPoseInGlobalFrame(unsigned poseID, Matrix &T) {
    Matrix Xi;
    if (!globalAnchor || !getSharedPose(poseID, Xi)) return false;
    T = roundPose(Xi);
    return true;
  }
  bool getNeighborPoseInGlobalFrame(unsigned neighborID, unsigned poseID, Matrix &T) {
    auto it = neighborPoseDict.find(PoseID(neighborID, poseID));
    if (!globalAnchor || it == neighborPoseDict.end()) return false;
    T = roundPose(it->second.getData());
    return true;
  }

 protected:
  // members PGOAgentROS touches directly (SURVEY App. A "Protected members touched")
  unsigned mID;
  unsigned d, r;
  const PGOAgentParameters mParams;
  PGOAgentState mState;
  PGOAgentStatus mStatus;
  std::shared_ptr<PoseGraph> mPoseGraph;
  RobustCost mRobustCost;
  ROPTResult mLocalOptResult;
  unsigned mInstanceNumber = 0, mIterationNumber = 0;
  int mWeightUpdateCount = 0, mRobustOptInnerIter = 0;
  std::map<unsigned, PGOAgentStatus> mTeamStatus;
  std::vector<bool> mTeamRobotActive;
  bool mPublishPublicPosesRequested = false, mPublishAsynchronousRequested = false;
  std::optional<Matrix> YLift;
  std::optional<LiftedPose> globalAnchor;
  std::optional<PoseArray> TLocalInit;
  PoseDict neighborPoseDict, neighborAuxPoseDict;
  std::recursive_mutex mMutex;
  std::unique_ptr<std::thread> mOptimizationThread;
  std::atomic<bool> mEndLoopRequested{false};

 private:
  dpgo_team_t *team_ = nullptr;
  const PoseGraph *synced_graph_ = nullptr;
  struct { size_t odom, priv, shared; } pushed_ = {0, 0, 0};

  static dpgo_params_t toC(const PGOAgentParameters &p) {
    dpgo_params_t c;
    dpgo_default_params(&c, (int)p.r, (int)p.numRobots);
    c.method = p.localOptimizationParams.method == ROptParameters::ROptMethod::RTR ? DPGO_METHOD_RTR : DPGO_METHOD_RGD;
    c.rgd_stepsize = p.localOptimizationParams.RGD_stepsize;
    c.rgd_use_preconditioner = p.localOptimizationParams.RGD_use_preconditioner;
    c.rtr_iterations = (int)p.localOptimizationParams.RTR_iterations;
    c.rtr_tcg_iterations = (int)p.localOptimizationParams.RTR_tCG_iterations;
    c.gradnorm_tol = p.localOptimizationParams.gradnorm_tol;
    c.rtr_initial_radius = p.localOptimizationParams.RTR_initial_radius;
    c.rtr_max_radius = 5 * c.rtr_initial_radius;
    c.rgd_line_search = p.localOptimizationParams.RGD_line_search ? 1 : 0;
    c.rgd_ls_max_backoffs = (int)p.localOptimizationParams.RGD_ls_max_backoffs;
    c.rgd_ls_shrink = p.localOptimizationParams.RGD_ls_shrink; c.rgd_ls_sigma = p.localOptimizationParams.RGD_ls_sigma;
    c.acceleration = p.acceleration; c.restart_interval = (int)p.restartInterval;
    c.rel_change_tol = p.relChangeTol; c.max_num_iters = (int)p.maxNumIters;
    // (Type's enumerators are declared in the order of DPGO_COST_*: L2, L1, Huber, TLS, GM, GNC_TLS)
    static_assert((int)RobustCostParameters::Type::L1 == DPGO_COST_L1 && (int)RobustCostParameters::Type::Huber == DPGO_COST_HUBER &&
                      (int)RobustCostParameters::Type::TLS == DPGO_COST_TLS && (int)RobustCostParameters::Type::GM == DPGO_COST_GM &&
                      (int)RobustCostParameters::Type::GNC_TLS == DPGO_COST_GNC_TLS, "RobustCostParameters::Type follows DPGO_COST_*");
    c.robust_cost_type = (int)p.robustCostParams.costType;
    c.tls_threshold = p.robustCostParams.TLSThreshold; c.huber_threshold = p.robustCostParams.HuberThreshold;
    c.gnc_barc = p.robustCostParams.GNCBarc; c.gnc_mu_step = p.robustCostParams.GNCMuStep; c.gnc_init_mu = p.robustCostParams.GNCInitMu;
    c.robust_opt_num_weight_updates = (int)p.robustOptNumWeightUpdates; c.robust_opt_inner_iters = (int)p.robustOptInnerIters;
    c.robust_opt_num_resets = (int)p.robustOptNumResets; c.robust_opt_min_convergence_ratio = p.robustOptMinConvergenceRatio;
    return c;
  }
  void destroyTeam() { if (team_) { dpgo_team_destroy(team_); team_ = nullptr; } synced_graph_ = nullptr; pushed_ = {0, 0, 0}; }
  void ensureTeam() {
    if (team_ && synced_graph_ != mPoseGraph.get()) destroyTeam();  // the wrapper replaced mPoseGraph (:237)
    if (!team_) {
      const dpgo_params_t c = toC(mParams);
      const int id = (int)mID;
      team_ = dpgo_team_create(0, &c, 1, &id, nullptr);
      DPGO_CHECK(team_ != nullptr);
      synced_graph_ = mPoseGraph.get();
    }
    syncMeasurements();
  }
  void syncMeasurements() {
    if (!team_) return;
    auto push = [&](const std::vector<RelativeSEMeasurement> &v, size_t &done) {
      for (; done < v.size(); ++done) { const dpgo_measurement_t c = v[done].toC(); DPGO_CHECK(dpgo_agent_add_measurements(team_, (int)mID, &c, 1) == DPGO_OK); }
    };
    push(mPoseGraph->odometry(), pushed_.odom);
    push(mPoseGraph->privateLoopClosures(), pushed_.priv);
    push(mPoseGraph->sharedLoopClosures(), pushed_.shared);
    mPoseGraph->takeDirtyStructure();
    if (mPoseGraph->takeDirtyData()) {  // weights were edited in place and clearDataMatrices() called (:1351)
      const auto all = mPoseGraph->allMeasurements();
      std::vector<double> w(all.size());
      std::vector<int> fx(all.size());
      for (size_t k = 0; k < all.size(); ++k) { w[k] = all[k]->weight; fx[k] = all[k]->fixedWeight ? 1 : 0; }
      DPGO_CHECK(dpgo_agent_set_measurement_weights(team_, (int)mID, w.data(), fx.data(), (int)all.size()) == DPGO_OK);
    }
  }
  Matrix fetchX(int which) {
    Matrix X = Matrix::Zero(r, (d + 1) * num_poses());
    DPGO_CHECK(dpgo_agent_get_X(team_, (int)mID, which, X.data()) == DPGO_OK);
    return X;
  }
  bool sharedDict(PoseDict &map, unsigned nbr, int aux) {
    std::lock_guard<std::recursive_mutex> lock_(mMutex);

    if (mState != PGOAgentState::INITIALIZED) return false;
    syncMeasurements();
    const int cnt = dpgo_agent_public_pose_ids(team_, (int)mID, (int)nbr, nullptr);
    if (cnt < 0) return false;
    std::vector<int> frames(cnt > 0 ? cnt : 1);
    std::vector<double> poses((size_t)(cnt > 0 ? cnt : 1) * 4 * r);
    dpgo_agent_public_pose_ids(team_, (int)mID, (int)nbr, frames.data());
    if (dpgo_agent_get_public_poses(team_, (int)mID, (int)nbr, aux, poses.data()) != DPGO_OK) return false;
    map.clear();
    for (int k = 0; k < cnt; ++k) {
      Matrix Xi = Matrix::Zero(r, d + 1);
      std::memcpy(Xi.data(), poses.data() + (size_t)k * 4 * r, sizeof(double) * 4 * r);
      map.emplace(PoseID(mID, (unsigned)frames[k]), LiftedPose(Xi));
    }
    return true;
  }
  void pushDict(unsigned nbr, const PoseDict &dict, bool aux) {
    if (!team_) return;
    syncMeasurements();
    std::vector<int> frames;
    std::vector<double> poses;
    for (const auto &kv : dict) {
      if (kv.first.robot_id != nbr) continue;
      frames.push_back((int)kv.first.frame_id);
      const Matrix &Xi = kv.second.getData();
      poses.insert(poses.end(), Xi.data(), Xi.data() + 4 * r);
    }
    if (!frames.empty())
      DPGO_CHECK(dpgo_agent_update_neighbor_poses(team_, (int)mID, (int)nbr, aux ? 1 : 0, (int)frames.size(), frames.data(), poses.data()) == DPGO_OK);
  }
  void pushNeighborDict(bool aux) {
    for (unsigned nbr : mPoseGraph->activeNeighborIDs()) pushDict(nbr, aux ? neighborAuxPoseDict : neighborPoseDict, aux);
  }
  static Matrix projectToRotationGroup(const Matrix &Min) {
    Matrix Rm = Min;
    for (int it = 0; it < 30; ++it) {  // Newton-Schulz polar iteration, input is close to a rotation
      Matrix G = Rm.transpose() * Rm;
      Matrix T = Matrix::Zero(3, 3);
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T(i, j) = (i == j ? 1.5 : 0.0) - 0.5 * G(i, j);
      Rm = Rm * T;
    }
    return Rm;
  }
  Matrix roundPose(const Matrix &Xi) const {
    // T = Ya^T X, translation relative to the anchor, rotation projected to SO(d)
    const Matrix Ya = globalAnchor->rotation(), pa = globalAnchor->translation();
    const Matrix Yt = Ya.transpose();
    Matrix T = Matrix::Zero(d, d + 1);
    mat_set_block(T, 0, 0, projectToRotationGroup(Yt * mat_block(Xi, 0, 0, r, d)));
    const Matrix t = Yt * (mat_block(Xi, 0, d, r, 1) - pa);
    mat_set_block(T, 0, d, t);
    return T;
  }
  // One candidate for T_world_robot per shared loop closure whose neighbour pose is in `dict`.  L2 cost: the
  // first candidate initialises the frame; robust cost: GNC-TLS two-stage averaging over all of them
  // (dpgo_robust_frame_alignment), and the agent stays uninitialised until robustInitMinInliers agree.
  void tryAlignWithNeighbor(unsigned nbr, const PoseDict &dict) {
    std::vector<double> cand;
    for (const auto &m : mPoseGraph->sharedLoopClosures()) {
      const bool out = (m.r1 == mID);
      if ((out ? m.r2 : m.r1) != nbr) continue;
      auto it = dict.find(PoseID(nbr, (unsigned)(out ? m.p2 : m.p1)));
      if (it == dict.end()) continue;
      const Matrix Tn_l = (*YLift).transpose() * it->second.getData();  // d x (d+1)
      Matrix Tn = Matrix::Zero(d, d + 1);
      mat_set_block(Tn, 0, 0, projectToRotationGroup(mat_block(Tn_l, 0, 0, d, d)));
      mat_set_block(Tn, 0, d, mat_block(Tn_l, 0, d, d, 1));
      Matrix Tm = Matrix::Zero(d, d + 1);
      mat_set_block(Tm, 0, 0, m.R); mat_set_block(Tm, 0, d, m.t);
      const Pose T_world_nbr(Tn), T_meas(Tm);
      const Pose T_world_mine = out ? T_world_nbr * T_meas.inverse() : T_world_nbr * T_meas;
      const Pose T_local(TLocalInit->pose((unsigned)(out ? m.p1 : m.p2)));
      const Pose T_world_robot = T_world_mine * T_local.inverse();
      if (mParams.robustCostParams.costType == RobustCostParameters::Type::L2) {
        initializeInGlobalFrame(T_world_robot);
        return;
      }
      const Matrix &Tc = T_world_robot.getData();
      for (unsigned c = 0; c < d + 1; ++c) for (unsigned a = 0; a < d; ++a) cand.push_back(Tc(a, c));
    }
    const int n = (int)(cand.size() / 12);
    if (n == 0) return;
    double Tout[12];
    if (dpgo_robust_frame_alignment(cand.data(), n, mParams.robustInitMaxRotationError, mParams.robustInitMaxTranslationError,
                                    (int)mParams.robustInitMinInliers, Tout, nullptr) != DPGO_OK) return;
    Matrix T = Matrix::Zero(d, d + 1);
    for (unsigned c = 0; c < d + 1; ++c) for (unsigned a = 0; a < d; ++a) T(a, c) = Tout[3 * c + a];
    initializeInGlobalFrame(Pose(T));
  }
};

}  // namespace DPGO
