// mock_wrapper.cpp -- drives the DPGO:: facade exactly the way PGOAgentROS does:
//   add measurements -> initialize -> lifting matrix -> public poses -> iterate(true/false) -> status.
// ROS is absent from this image, so this subclass stands in for `class PGOAgentROS : public PGOAgent`
// and touches the same protected members (src/PGOAgentROS.cpp).
// Usage: mock_wrapper <g2o> <num_robots> <iterations> [accel]   -> prints one cost line per iteration.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <memory>
#include <vector>

#include <DPGO/PGOAgent.h>

using namespace DPGO;

class MockAgentROS : public PGOAgent {
 public:
  MockAgentROS(unsigned id, const PGOAgentParameters &p) : PGOAgent(id, p) {}
  // protected members, as the wrapper uses them
  bool publishRequested() { const bool v = mPublishPublicPosesRequested; mPublishPublicPosesRequested = false; return v; }
  PGOAgentState state() const { return mState; }
  double relChange() const { return mStatus.relativeChange; }
  const ROPTResult &optResult() const { return mLocalOptResult; }
  std::shared_ptr<PoseGraph> graph() { return mPoseGraph; }
  void setIteration(unsigned k) { mIterationNumber = k; }
  bool hasYLift() const { return YLift.has_value(); }
  void reset() override { PGOAgent::reset(); }
  // the wrapper's own iteration log (src/PGOAgentROS.cpp:853-909), fed from the SAME facade members it reads there:
  // getID(), numActiveRobots(), iteration_number(), num_poses(), mStatus.relativeChange, mParams.logData / logDirectory
  bool openLog() {
    if (!mParams.logData) return false;
    log_ = std::fopen((mParams.logDirectory + "dpgo_log_robot" + std::to_string(getID()) + ".csv").c_str(), "w");
    if (!log_) return false;
    std::fputs("robot_id, cluster_id, num_active_robots, iteration, num_poses, bytes_received, "
               "iter_time_sec, total_time_sec, rel_change \n", log_);
    t0_ = std::chrono::steady_clock::now();
    return true;
  }
  void countBytes(size_t b) { bytes_ += b; }
  void logIteration(double iter_ms) {
    if (!mParams.logData || !log_) return;
    const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
    std::fprintf(log_, "%u,%u,%zu,%u,%u,%zu,%.9g,%.9g,%.17g\n", getID(), 0u, (size_t)numActiveRobots(), iteration_number(), num_poses(),
                 bytes_, iter_ms / 1e3, total, mStatus.relativeChange);
    std::fflush(log_);
  }
  void logString(const char *s) { if (mParams.logData && log_) { std::fprintf(log_, "%s\n", s); std::fflush(log_); } }
  ~MockAgentROS() override { if (log_) std::fclose(log_); }

 private:
  FILE *log_ = nullptr;
  size_t bytes_ = 0;
  std::chrono::steady_clock::time_point t0_{};
};

static void publish(std::vector<std::unique_ptr<MockAgentROS>> &team, unsigned b, bool aux) {
  for (unsigned nbr : team[b]->getNeighbors()) {
    PoseDict map;
    if (!(aux ? team[b]->getAuxSharedPoseDictWithNeighbor(map, nbr) : team[b]->getSharedPoseDictWithNeighbor(map, nbr))) continue;
    if (aux) team[nbr]->updateAuxNeighborPoses(b, map); else team[nbr]->updateNeighborPoses(b, map);
    team[nbr]->countBytes(map.size() * 8 * 4 * 5);  // (float64[] r x 4 per pose, msg/PublicPoses.msg; :1283 counts the message)
  }
}

static double global_cost(std::vector<std::unique_ptr<MockAgentROS>> &team) {
  double f = 0;
  for (auto &a : team)
    for (auto *m : a->graph()->allMeasurements()) {
      if (m->r1 != m->r2 && std::min(m->r1, m->r2) != a->getID()) continue;
      double res = 0;
      if (!a->computeMeasurementResidual(*m, &res)) { std::fprintf(stderr, "residual unavailable\n"); std::exit(2); }
      f += 0.5 * m->weight * res * res;
    }
  return f;
}

int main(int argc, char **argv) {
  if (argc < 4) return 1;
  const unsigned N = (unsigned)std::atoi(argv[2]);
  const int iters = std::atoi(argv[3]);
  const int mode = argc > 4 ? std::atoi(argv[4]) : 0;  // 0 plain, 1 accelerated, 2 asynchronous (ASAPP), 3 robust cost (GNC-TLS frame alignment),
                                                       // 4 local_initialization_method = GNC_TLS
  const bool accel = mode == 1 || mode == 5;  // 5: acceleration + GNC-TLS with the leader's UPDATE_WEIGHT / TERMINATE decisions
  size_t num_poses = 0;
  std::vector<RelativeSEMeasurement> dataset = read_g2o_file(argv[1], num_poses);
  PGOAgentParameters params(3, 5, N);
  params.localOptimizationParams.method = ROptParameters::ROptMethod::RTR;
  params.localOptimizationParams.gradnorm_tol = 1e-2;
  params.acceleration = accel;
  params.restartInterval = 7;
  params.relChangeTol = 0.2;
  if (mode == 2) {  // src/PGOAgentROSNode.cpp:86-93: asynchronous => RGD at asynchronous_rate; README.md:52 stepsize 0.2
    params.asynchronous = true;
    params.asynchronousOptimizationRate = 100.0;  // launch/asapp_demo.launch:25-26
    params.localOptimizationParams.method = ROptParameters::ROptMethod::RGD;
    params.localOptimizationParams.RGD_stepsize = 0.05;  // simultaneous (Jacobi-like) updates on this tightly coupled pair need a smaller step
  }
  if (mode == 3) {  // launch/dpgo_gnc_demo.launch:35-42; frame alignment averages robustly over every shared loop closure
    params.robustCostParams.costType = RobustCostParameters::Type::GNC_TLS;
    params.robustOptInnerIters = 1000000;  // no weight update inside this short run
    params.robustInitMinInliers = 2;
  }
  if (mode == 5) {  // launch/dpgo_gnc_demo.launch:35-42 scaled down, with acceleration on
    params.robustCostParams.costType = RobustCostParameters::Type::GNC_TLS;
    params.robustCostParams.GNCBarc = 3.0;
    params.robustCostParams.GNCMuStep = 2.0;
    params.robustCostParams.GNCInitMu = 1e-2;
    params.robustOptNumWeightUpdates = 3;
    params.robustOptInnerIters = 2 * N;
    params.robustOptMinConvergenceRatio = 0.0;
    params.robustInitMinInliers = 2;
    params.relChangeTol = 0.05;
    params.maxNumIters = 1000;
  }
  if (mode == 4) {  // src/PGOAgentROSNode.cpp:111-112
    params.localInitializationMethod = InitializationMethod::GNC_TLS;
    params.robustCostParams.GNCBarc = 5.0;
    params.robustOptNumWeightUpdates = 4;
    params.robustOptInnerIters = 5;
  }
  if (const char *dir = std::getenv("MOCK_LOG_DIR")) { params.logData = true; params.logDirectory = std::string(dir) + "/"; }
  std::vector<std::unique_ptr<MockAgentROS>> team;
  for (unsigned k = 0; k < N; ++k) team.emplace_back(new MockAgentROS(k, params));
  for (auto &a : team) a->openLog();
  for (auto &a : team) for (unsigned k = 0; k < N; ++k) a->setRobotActive(k, true);  // setActiveRobots() (:380-390)
  // src/PGODatasetPublisherNode.cpp:84-135 partition; every robot ends with all edges incident to it
  const unsigned per = (unsigned)num_poses / N;
  for (const auto &mIn : dataset) {
    const unsigned ra = std::min<unsigned>(mIn.p1 / per, N - 1), rb = std::min<unsigned>(mIn.p2 / per, N - 1);
    RelativeSEMeasurement m(ra, rb, mIn.p1 - ra * per, mIn.p2 - rb * per, mIn.R, mIn.t, mIn.kappa, mIn.tau);
    team[ra]->addMeasurement(m);
    if (rb != ra) team[rb]->addMeasurement(m);
  }
  // lifting matrix from robot 0 (:404, :928); local initialisation; robot 0 defines the global frame (:348-353)
  Matrix YLift;
  if (!team[0]->getLiftingMatrix(YLift)) return 3;
  for (unsigned k = 0; k < N; ++k) { team[k]->setLiftingMatrix(YLift); team[k]->initialize(); }
  team[0]->initializeInGlobalFrame(Pose(3));
  for (unsigned round = 0; round < N; ++round)  // frame alignment spreads along the chain (:1276)
    for (unsigned b = 0; b < N; ++b)
      if (team[b]->state() == PGOAgentState::INITIALIZED) { publish(team, b, false); if (accel) publish(team, b, true); }
  for (unsigned k = 0; k < N; ++k) if (team[k]->state() != PGOAgentState::INITIALIZED) { std::fprintf(stderr, "robot %u not initialised\n", k); return 4; }
  Matrix anchor;
  if (!team[0]->getSharedPose(0, anchor)) return 5;
  for (auto &a : team) a->setGlobalAnchor(anchor);  // :431-438, :932-939
  std::printf("init cost %.12e\n", global_cost(team));
  if (mode == 2) {
    // runOnceAsynchronous (:119-127) + publishPublicPoses (:109-113) polled from this "ROS" thread
    for (int k = 0; k < iters; ++k) {
      std::this_thread::sleep_for(std::chrono::milliseconds(4));
      for (unsigned b = 0; b < N; ++b) if (team[b]->publishRequested()) publish(team, b, false);
      unsigned its = 0;
      for (auto &a : team) its += a->iteration_number();
      std::printf("async poll %d local_iterations %u cost %.12e\n", k + 1, its, global_cost(team));
    }
    for (auto &a : team) a->endOptimizationLoop();
    for (unsigned b = 0; b < N; ++b) publish(team, b, false);  // consistent snapshot
    std::printf("final cost %.12e\n", global_cost(team));
    return 0;
  }
  for (int k = 0; k < iters; ++k) {
    const unsigned sel = (unsigned)k % N;  // RoundRobin token (:464-473)
    for (unsigned b = 0; b < N; ++b)
      if (b != sel) { team[b]->iterate(false); if (team[b]->publishRequested()) { publish(team, b, false); if (accel) publish(team, b, true); } }
    const auto it0 = std::chrono::steady_clock::now();
    if (!team[sel]->iterate(true)) { std::fprintf(stderr, "iterate failed\n"); return 6; }
    team[sel]->logIteration(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - it0).count());  // :185-189
    if (team[sel]->publishRequested()) { publish(team, sel, false); if (accel) publish(team, sel, true); }
    for (auto &a : team) for (auto &b : team) if (a != b) a->setNeighborStatus(b->getStatus());
    const bool term = team[0]->shouldTerminate();
    std::printf("iter %d robot %u cost %.12e relchange %.6e fdec %.3e terminate %d\n", k + 1, sel, global_cost(team),
                team[sel]->relChange(), team[sel]->optResult().fInit - team[sel]->optResult().fOpt, (int)term);
    if (mode == 5 && sel == 0) {
      // the leader's decision after its own block update (src/PGOAgentROS.cpp:206-214)
      if (term) { std::printf("TERMINATE at %d\n", k + 1); for (auto &a : team) a->logString("TERMINATE"); break; }  // :1042
      if (team[0]->shouldUpdateMeasurementWeights()) {
        for (auto &a : team) a->logString("UPDATE_WEIGHT");  // :1217
        // UPDATE_WEIGHT (:1211-1233): every robot re-weights what it owns, sends shared-edge weights to the higher-ID
        // endpoint (:721-754), which applies them and clears its data matrices (:1315-1353); public poses follow
        for (auto &a : team) a->updateMeasurementWeights();
        for (auto &a : team)
          for (const auto &m : a->graph()->sharedLoopClosures()) {
            const unsigned other = (m.r1 == a->getID()) ? m.r2 : m.r1;
            if (other <= a->getID()) continue;
            if (team[other]->setMeasurementWeight(PoseID(m.r1, m.p1), PoseID(m.r2, m.p2), m.weight, m.fixedWeight))
              team[other]->graph()->clearDataMatrices();
          }
        for (unsigned b = 0; b < N; ++b) { publish(team, b, false); publish(team, b, true); }
        std::printf("UPDATE_WEIGHT at %d cost %.12e\n", k + 1, global_cost(team));
      }
    }
  }
  PoseArray T(3, 1);
  if (!team[N - 1]->getTrajectoryInGlobalFrame(T)) return 7;
  const Matrix R0 = T.rotation(0);
  double orth = 0;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += R0(k, i) * R0(k, j); orth += std::abs(s - (i == j)); }
  std::printf("trajectory poses %u orthogonality_defect %.3e\n", T.n(), orth);
  return 0;
}
