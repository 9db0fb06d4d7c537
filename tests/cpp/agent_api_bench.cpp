// agent_api_bench.cpp -- the drop-in path timed from C++ (no interpreter in the loop): one single-agent team per
// robot, the way the ROS wrapper runs one PGOAgent per process; every RBCD iteration is
//   iterate(false) + getStatus on every robot but the token holder (concurrent processes in the reference: all of them
//   first), then their [get*SharedPoseDictWithNeighbor -> updateNeighborPoses of the neighbours], then
//   iterate(true) + getStatus + mLocalOptResult + publish on the token holder
// (src/PGOAgentROS.cpp:109-113,160,183-186,616,662-690,1255-1284), all exchange through HOST buffers.
// Usage: agent_api_bench <g2o> <robots> <method 0 RTR | 1 RGD> <accel> <iterations> [stepsize] [restart] [gradnorm_tol]
// Prints one JSON object.  Used by bench.py (convergence.agent_api.*.ms_per_iterate_cxx) and tests/test_facade.py.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dpgo_hip.h"

#define CK(x) do { if ((x) < 0) { std::fprintf(stderr, "%s failed: %s\n", #x, dpgo_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
  if (argc < 6) { std::fprintf(stderr, "usage: %s g2o robots method accel iterations [stepsize] [restart]\n", argv[0]); return 2; }
  const int N = std::atoi(argv[2]), method = std::atoi(argv[3]), accel = std::atoi(argv[4]), iters = std::atoi(argv[5]);
  const int r = 5;
  dpgo_measurement_t *m = nullptr;
  int n = 0;
  const int nm = dpgo_read_g2o(argv[1], DPGO_WEIGHT_LIBRARY, &m, &n);
  if (nm < 0) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  std::vector<double> T((size_t)12 * n), Y((size_t)3 * r), X((size_t)4 * r * n);
  dpgo_odometry_init(m, nm, n, T.data());
  dpgo_fixed_stiefel(r, Y.data());
  dpgo_lift(T.data(), n, Y.data(), r, X.data());
  dpgo_partition(m, nm, n, N, DPGO_WEIGHT_LIBRARY);
  dpgo_params_t p;
  dpgo_default_params(&p, r, N);
  p.method = method; p.acceleration = accel;
  p.rgd_stepsize = argc > 6 ? std::atof(argv[6]) : 0.2;
  p.restart_interval = argc > 7 ? std::atoi(argv[7]) : 20;
  p.rgd_use_preconditioner = 1; p.rtr_iterations = 3; p.rtr_tcg_iterations = 50;
  p.gradnorm_tol = argc > 8 ? std::atof(argv[8]) : 0.5;
  std::vector<dpgo_team_t *> team(N);
  const int per = n / N;
  for (int a = 0; a < N; ++a) {
    team[a] = dpgo_team_create(0, &p, 1, &a, nullptr);
    if (!team[a]) { std::fprintf(stderr, "team_create: %s\n", dpgo_last_error()); return 1; }
    CK(dpgo_agent_add_measurements(team[a], a, m, nm));
    CK(dpgo_agent_set_X(team[a], a, X.data() + (size_t)4 * r * per * a));
  }
  // neighbour tables
  std::vector<std::vector<int>> nbrs(N);
  std::vector<std::vector<std::vector<int>>> ids(N);
  std::vector<double> buf;
  size_t maxp = 1;
  for (int a = 0; a < N; ++a) {
    nbrs[a].resize(dpgo_agent_get_neighbors(team[a], a, nullptr));
    dpgo_agent_get_neighbors(team[a], a, nbrs[a].data());
    for (int c : nbrs[a]) {
      std::vector<int> f(dpgo_agent_public_pose_ids(team[a], a, c, nullptr));
      dpgo_agent_public_pose_ids(team[a], a, c, f.data());
      maxp = std::max(maxp, f.size());
      ids[a].push_back(f);
    }
  }
  buf.resize(maxp * 4 * r);
  double t_get = 0, t_upd = 0, t_get_aux = 0;
  auto publish = [&](int b) -> int {
    for (size_t q = 0; q < nbrs[b].size(); ++q) {
      const int c = nbrs[b][q];
      for (int aux = 0; aux <= (accel ? 1 : 0); ++aux) {
        const auto g0 = std::chrono::steady_clock::now();
        CK(dpgo_agent_get_public_poses(team[b], b, c, aux, buf.data()));
        const auto g1 = std::chrono::steady_clock::now();
        CK(dpgo_agent_update_neighbor_poses(team[c], c, b, aux, (int)ids[b][q].size(), ids[b][q].data(), buf.data()));
        t_get += std::chrono::duration<double, std::micro>(g1 - g0).count();
        if (aux) t_get_aux += std::chrono::duration<double, std::micro>(g1 - g0).count();
        t_upd += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g1).count();
      }
    }
    return 0;
  };
  for (int b = 0; b < N; ++b) if (publish(b)) return 1;
  dpgo_status_t st;
  dpgo_opt_result_t res{};
  double rel_sum = 0;
  // where the host's time goes: iterate(false) calls, iterate(true) calls, everything else (getters + exchange)
  double t_false = 0, t_true = 0;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::micro>(b - a).count();
  };
  // The robots are separate processes in the reference: on an UPDATE command every robot but the token holder runs
  // iterate(false) + publishStatus AT THE SAME TIME (:1183-1186), then publishes from its own runOnce (:109-113).  This
  // single thread plays one valid interleaving of that: every robot's iterate(false) + getStatus first (iterate(false)
  // only enqueues -- its kernels run side by side on the robots' streams), then every robot's publish.
  auto iteration = [&](int k) -> int {
    const int sel = k % N;
    for (int b = 0; b < N; ++b) {
      if (b == sel) continue;
      const auto a0 = now();
      CK(dpgo_agent_iterate(team[b], b, 0));
      t_false += us(a0, now());
      CK(dpgo_agent_get_status(team[b], b, &st));
    }
    for (int b = 0; b < N; ++b) {
      if (b == sel) continue;
      if (dpgo_agent_publish_requested(team[b], b, 1) > 0 && publish(b)) return 1;
    }
    const auto a0 = now();
    CK(dpgo_agent_iterate(team[sel], sel, 1));
    t_true += us(a0, now());
    CK(dpgo_agent_get_status(team[sel], sel, &st));
    rel_sum += st.relative_change;
    CK(dpgo_agent_get_opt_result(team[sel], sel, &res));
    if (dpgo_agent_publish_requested(team[sel], sel, 1) > 0 && publish(sel)) return 1;
    return 0;
  };
  const int warm = 2 * N;
  for (int k = 0; k < warm; ++k) if (iteration(k)) return 1;
  t_false = t_true = 0; t_get = t_upd = t_get_aux = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = warm; k < warm + iters; ++k) if (iteration(k)) return 1;
  for (int a = 0; a < N; ++a) CK(dpgo_team_synchronize(team[a]));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iters;
  double wait_us = 0, reports = 0;
  for (int a = 0; a < N; ++a) {
    double c[8];
    CK(dpgo_team_get_counters(team[a], c, 8));
    wait_us += c[5]; reports += c[6];
  }
  // a checksum of the final iterate so that callers can compare runs (sum of |X| entries of every agent)
  double checksum = 0;
  std::vector<double> Xa;
  for (int a = 0; a < N; ++a) {
    const int na = dpgo_agent_num_poses(team[a], a);
    Xa.resize((size_t)4 * r * na);
    CK(dpgo_agent_get_X(team[a], a, 0, Xa.data()));
    for (double v : Xa) checksum += std::fabs(v);
  }
  std::printf("{\"ms_per_iteration\": %.6f, \"iterations\": %d, \"us_per_iterate_false\": %.2f, \"us_per_iterate_true\": %.2f, "
              "\"us_other_per_iteration\": %.2f, \"us_get_public_poses_per_iteration\": %.2f, \"us_get_aux_poses_per_iteration\": %.2f, \"us_update_neighbor_poses_per_iteration\": %.2f, \"us_report_wait\": %.2f, \"f_opt_last\": %.12g, \"relchange_sum\": %.12g, \"checksum\": %.15g}\n",
              ms, iters, t_false / ((double)iters * (N - 1)), t_true / iters, 1e3 * ms - (t_false + t_true) / iters, t_get / iters, t_get_aux / iters, t_upd / iters, reports > 0 ? wait_us / reports : 0.0, res.f_opt, rel_sum,
              checksum);
  for (int a = 0; a < N; ++a) dpgo_team_destroy(team[a]);
  dpgo_free(m);
  return 0;
}
