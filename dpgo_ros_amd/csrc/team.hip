// team.hip -- host runtime + C-ABI of libdpgo_hip.so (see include/dpgo_hip.h for the contract).
//
// Mirrors the DPGO::PGOAgent call surface consumed by src/PGOAgentROS.cpp:
// addMeasurement, iterate, update(Aux)NeighborPoses, get(Aux)SharedPoseDictWithNeighbor,
// getStatus, mLocalOptResult, updateMeasurementWeights, setMeasurementWeight, clearDataMatrices.
// All state (X, XPrev, Y, V, Q, G, dense preconditioner, neighbour slabs, solver scalars) lives
// in HBM; the host only sequences launches.  There is no CPU fallback: every entry point that
// computes fails with DPGO_ERR when no HIP device is usable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/dpgo_hip.h"
#include "dpgo_dev.h"
#include "kernels.h"

using namespace dpgo;

namespace {

thread_local std::string g_err;
void set_err(const std::string &s) { g_err = s; }

#define HIPC(expr)                                                                         \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      set_err(std::string(#expr) + ": " + hipGetErrorString(e_) + " @" + std::to_string(__LINE__)); \
      return DPGO_ERR;                                                                     \
    }                                                                                      \
  } while (0)

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t count) {
    if (count <= n && p) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
    if (hipMalloc(&p, sizeof(T) * std::max<size_t>(count, 1)) != hipSuccess) return -1;
    n = std::max<size_t>(count, 1);
    return 0;
  }
  int upload(const std::vector<T> &v, hipStream_t s) {
    if (alloc(v.size())) return -1;
    if (v.empty()) return 0;
    return hipMemcpyAsync(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, s) == hipSuccess ? 0 : -1;
  }
};

struct Agent {
  int id = 0, local = 0;
  std::vector<dpgo_measurement_t> odom, priv, shared;
  bool index_dirty = true, data_dirty = true;
  int n = 0;
  // neighbour pose dictionary (sorted (robot, frame)) and per-neighbour public ids
  std::vector<std::pair<int, int>> np;
  std::vector<char> np_has[2];
  std::vector<int> neighbors;
  int state = DPGO_WAIT_FOR_DATA;
  int iter = 0, instance = 0;
  bool publish_requested = false;
  bool has_X = false;
  double mu = 0;
  int weight_update_count = 0, robust_inner_iter = 0;
  dpgo_opt_result_t opt{};
  bool opt_pending_rgd = false, last_success = true;
  // host copies of the sparse structure
  std::vector<int> rowptr, col;
  std::vector<double> qval;
  int npub = 0;
  // device storage
  DevBuf<int> d_rowptr, d_col, d_pub_pose, d_pub_ptr, d_idx, d_ell_col, d_trowptr, d_tcol, d_pub_index;
  DevBuf<double> d_qval, d_M, d_vec, d_nbr, d_part, d_scal, d_resid, d_ell_val, d_tval;
  std::map<int, std::unique_ptr<DevBuf<int>>> d_pubframes, d_nbrslots;  // per neighbour, cached on the device
  std::map<int, int> n_pubframes, n_nbrslots;
  DevBuf<double> d_xfer;
  int tcg_hint = 4, outer_hint = -1;  // launch-pattern sizing from the previous solve of this agent
  int rel_src = 0;  // where the last |X - XPrev|^2 partials live: 0 PART_D (per 64-pose tile), 1 PART_B[2] (fused RGD)
  DevBuf<SharedEdgeDev> d_se;
  DevBuf<EdgeDev> d_edges;
  DevBuf<RtrState> d_st;
  DevBuf<NestState> d_nest;
  AgentDev dev{};
  int nedges = 0;
};

}  // namespace

struct dpgo_team {
  int device = 0;
  dpgo_params_t prm{};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::vector<std::unique_ptr<Agent>> ag;
  std::map<int, int> id2local;
  DevBuf<AgentDev> d_agents;
  DevBuf<TeamDev> d_team;
  DevBuf<int> d_sched, d_group_ptr, d_group_members;
  std::vector<std::vector<int>> groups;  // colour classes (local agent indices), greedy colouring
  std::vector<int> color_of;
  bool user_groups = false;              // groups supplied by dpgo_team_set_groups (global colouring)
  RtrState *h_states = nullptr;          // pinned, one per local agent
  DevBuf<double> d_tmp;  // scratch for raw manifold ops / dense factorisation
  std::vector<int> sched;
  int iter = 0;
  bool descs_dirty = true;
  int max_n = 0, max_npub = 0;
  RtrState *h_state = nullptr;  // pinned
  double *h_scal = nullptr;     // pinned [16]
  static constexpr int NGRAPH = 5;       // graphs of 1, 2, 4, 8, 16 identical iterations
  hipGraphExec_t graph[NGRAPH] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  bool graph_valid = false;
  double counters[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  LaunchCtx ctx() { return LaunchCtx{prm.r, stream, d_agents.p, d_team.p}; }
};

namespace {

Agent *find_agent(dpgo_team *t, int id) {
  auto it = t->id2local.find(id);
  if (it == t->id2local.end()) { set_err("unknown agent id " + std::to_string(id)); return nullptr; }
  return t->ag[it->second].get();
}

// 4x4 column-major blocks of one edge:  TO = T Omega, TOT = T Omega T^T, Om = Omega (x weight)
void edge_blocks(const dpgo_measurement_t &m, double TO[16], double TOT[16], double Om[16]) {
  const double w = m.weight, k = m.kappa, tau = m.tau;
  std::fill(TO, TO + 16, 0.0); std::fill(TOT, TOT + 16, 0.0); std::fill(Om, Om + 16, 0.0);
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) {
      TO[a + 4 * b] = w * k * m.R[3 * a + b];
      TOT[a + 4 * b] = w * ((a == b ? k : 0.0) + tau * (m.t[a] * m.t[b]));
    }
    TO[a + 12] = w * tau * m.t[a];
    TOT[a + 12] = w * tau * m.t[a];
    TOT[3 + 4 * a] = w * tau * m.t[a];
    Om[5 * a] = w * k;
  }
  TO[15] = TOT[15] = Om[15] = w * tau;
}

void rebuild_index(Agent &a) {
  if (!a.index_dirty) return;
  int n = 0;
  auto upd = [&](int p) { n = std::max(n, p + 1); };
  for (auto &m : a.odom) { upd(m.p1); upd(m.p2); }
  for (auto &m : a.priv) { upd(m.p1); upd(m.p2); }
  std::vector<std::pair<int, int>> np;
  for (auto &m : a.shared) {
    if (m.r1 == a.id) { upd(m.p1); np.emplace_back(m.r2, m.p2); }
    else { upd(m.p2); np.emplace_back(m.r1, m.p1); }
  }
  std::sort(np.begin(), np.end());
  np.erase(std::unique(np.begin(), np.end()), np.end());
  a.np = np;
  a.np_has[0].assign(np.size(), 0);
  a.np_has[1].assign(np.size(), 0);
  a.neighbors.clear();
  for (auto &p : np) if (a.neighbors.empty() || a.neighbors.back() != p.first) a.neighbors.push_back(p.first);
  a.n = n;
  a.index_dirty = false;
  a.data_dirty = true;
}

int find_np(const Agent &a, int robot, int frame) {
  auto it = std::lower_bound(a.np.begin(), a.np.end(), std::make_pair(robot, frame));
  if (it == a.np.end() || *it != std::make_pair(robot, frame)) return -1;
  return int(it - a.np.begin());
}

std::vector<int> public_ids(const Agent &a, int nbr) {
  std::vector<int> f;
  for (auto &m : a.shared) {
    if (m.r1 == a.id && m.r2 == nbr) f.push_back(m.p1);
    else if (m.r2 == a.id && m.r1 == nbr) f.push_back(m.p2);
  }
  std::sort(f.begin(), f.end());
  f.erase(std::unique(f.begin(), f.end()), f.end());
  return f;
}

std::vector<int> neighbor_ids(const Agent &a, int nbr) {
  std::vector<int> f;
  for (auto &p : a.np) if (p.first == nbr) f.push_back(p.second);
  return f;
}

// connection Laplacian in block-CSR (row j lists (i, Q_ij)); duplicates merged in insertion order
void build_Q(Agent &a) {
  std::vector<std::map<int, std::array<double, 16>>> rows(a.n);
  auto add = [&](int row, int colm, const double *v, bool transpose, double sign) {
    auto &blk = rows[row][colm];
    for (int cp = 0; cp < 4; ++cp)
      for (int c = 0; c < 4; ++c) blk[cp + 4 * c] += sign * (transpose ? v[c + 4 * cp] : v[cp + 4 * c]);
  };
  double TO[16], TOT[16], Om[16];
  for (int i = 0; i < a.n; ++i) rows[i][i];  // every pose owns a diagonal block
  for (int pass = 0; pass < 2; ++pass)
    for (auto &m : (pass ? a.priv : a.odom)) {
      edge_blocks(m, TO, TOT, Om);
      add(m.p1, m.p1, TOT, false, 1.0);
      add(m.p2, m.p2, Om, false, 1.0);
      add(m.p2, m.p1, TO, false, -1.0);  // Q_ij stored in row j
      add(m.p1, m.p2, TO, true, -1.0);   // Q_ji = Q_ij^T stored in row i
    }
  for (auto &m : a.shared) {
    edge_blocks(m, TO, TOT, Om);
    if (m.r1 == a.id) add(m.p1, m.p1, TOT, false, 1.0);
    else add(m.p2, m.p2, Om, false, 1.0);
  }
  a.rowptr.assign(a.n + 1, 0);
  a.col.clear(); a.qval.clear();
  for (int j = 0; j < a.n; ++j) {
    for (auto &kv : rows[j]) {
      a.col.push_back(kv.first);
      a.qval.insert(a.qval.end(), kv.second.begin(), kv.second.end());
    }
    a.rowptr[j + 1] = (int)a.col.size();
  }
}

}  // namespace


namespace {

// upload structure + data matrices of one agent and (re)build the dense preconditioner
int finalize_agent(dpgo_team *t, Agent &a) {
  rebuild_index(a);
  if (!a.data_dirty) return 0;
  const int r = t->prm.r, n = a.n, N4 = 4 * n;
  const size_t len = (size_t)r * 4 * n;
  hipStream_t s = t->stream;
  build_Q(a);
  // shared edges sorted by local pose
  struct SE { int lpose; SharedEdgeDev d; };
  std::vector<SharedEdgeDev> se;
  double TO[16], TOT[16], Om[16];
  for (auto &m : a.shared) {
    edge_blocks(m, TO, TOT, Om);
    const bool out = (m.r1 == a.id);
    SharedEdgeDev d{};
    d.lpose = out ? m.p1 : m.p2;
    const int nr = out ? m.r2 : m.r1, nf = out ? m.p2 : m.p1;
    d.slot = find_np(a, nr, nf);
    auto it = t->id2local.find(nr);
    d.src_agent_local = (it == t->id2local.end()) ? -1 : it->second;
    d.src_frame = nf;
    for (int cp = 0; cp < 4; ++cp)
      for (int c = 0; c < 4; ++c) d.coef[cp + 4 * c] = out ? TO[c + 4 * cp] : TO[cp + 4 * c];
    se.push_back(d);
  }
  std::stable_sort(se.begin(), se.end(), [](const SharedEdgeDev &x, const SharedEdgeDev &y) { return x.lpose < y.lpose; });
  std::vector<int> pub_pose, pub_ptr;
  for (size_t e = 0; e < se.size(); ++e) {
    if (e == 0 || se[e].lpose != se[e - 1].lpose) { pub_pose.push_back(se[e].lpose); pub_ptr.push_back((int)e); }
  }
  pub_ptr.push_back((int)se.size());
  a.npub = (int)pub_pose.size();
  // edge records for residual / cost evaluation
  std::vector<EdgeDev> edges;
  auto push_edge = [&](const dpgo_measurement_t &m) {
    EdgeDev e{};
    e.i_local = (m.r1 == a.id) ? m.p1 : -1;
    e.j_local = (m.r2 == a.id) ? m.p2 : -1;
    e.i_slot = (m.r1 == a.id) ? -1 : find_np(a, m.r1, m.p1);
    e.j_slot = (m.r2 == a.id) ? -1 : find_np(a, m.r2, m.p2);
    std::memcpy(e.R, m.R, sizeof e.R);
    std::memcpy(e.t, m.t, sizeof e.t);
    e.kappa = m.kappa; e.tau = m.tau; e.weight = m.weight;
    e.count_in_cost = (m.r1 == m.r2) ? 1 : (std::min(m.r1, m.r2) == a.id);
    edges.push_back(e);
  };
  for (auto &m : a.odom) push_edge(m);
  for (auto &m : a.priv) push_edge(m);
  for (auto &m : a.shared) push_edge(m);
  a.nedges = (int)edges.size();

  // ELL (slot-major, width <= 8) + CSR tail copy of Q for the SpMM kernels
  int maxlen = 0;
  for (int j = 0; j < n; ++j) maxlen = std::max(maxlen, a.rowptr[j + 1] - a.rowptr[j]);
  const int EW = std::min(maxlen, 8);
  std::vector<int> ell_col((size_t)EW * n), trowptr(n + 1, 0), tcol;
  std::vector<double> ell_val((size_t)EW * n * 16, 0.0), tval;
  for (int j = 0; j < n; ++j) {
    const int p0 = a.rowptr[j], p1 = a.rowptr[j + 1];
    for (int u = 0; u < EW; ++u) {
      const int p = p0 + u;
      ell_col[(size_t)u * n + j] = (p < p1) ? a.col[p] : j;
      if (p < p1) std::copy(a.qval.begin() + (size_t)16 * p, a.qval.begin() + (size_t)16 * (p + 1), ell_val.begin() + ((size_t)u * n + j) * 16);
    }
    for (int p = p0 + EW; p < p1; ++p) { tcol.push_back(a.col[p]); tval.insert(tval.end(), a.qval.begin() + (size_t)16 * p, a.qval.begin() + (size_t)16 * (p + 1)); }
    trowptr[j + 1] = (int)tcol.size();
  }
  std::vector<int> pub_index(n, -1);
  for (size_t q = 0; q < pub_pose.size(); ++q) pub_index[pub_pose[q]] = (int)q;
  if (a.d_ell_col.upload(ell_col, s) || a.d_ell_val.upload(ell_val, s) || a.d_trowptr.upload(trowptr, s) ||
      a.d_tcol.upload(tcol, s) || a.d_tval.upload(tval, s) || a.d_pub_index.upload(pub_index, s)) {
    set_err("device allocation/upload failed");
    return DPGO_ERR;
  }
  const bool fresh_vec = a.d_vec.n < len * NBUF;
  if (a.d_rowptr.upload(a.rowptr, s) || a.d_col.upload(a.col, s) || a.d_qval.upload(a.qval, s) ||
      a.d_pub_pose.upload(pub_pose, s) || a.d_pub_ptr.upload(pub_ptr, s) || a.d_se.upload(se, s) ||
      a.d_edges.upload(edges, s) || a.d_vec.alloc(len * NBUF) || a.d_nbr.alloc(2 * a.np.size() * 4 * r) ||
      a.d_part.alloc(PART_TOTAL) || a.d_scal.alloc(16) || a.d_resid.alloc(edges.size()) || a.d_st.alloc(2) ||
      a.d_nest.alloc(1) || a.d_M.alloc((size_t)N4 * N4)) {
    set_err("device allocation/upload failed");
    return DPGO_ERR;
  }
  if (fresh_vec) {
    HIPC(hipMemsetAsync(a.d_vec.p, 0, sizeof(double) * len * NBUF, s));
    HIPC(hipMemsetAsync(a.d_nbr.p, 0, sizeof(double) * a.d_nbr.n, s));
    HIPC(hipMemsetAsync(a.d_scal.p, 0, sizeof(double) * 16, s));
    HIPC(hipMemsetAsync(a.d_nest.p, 0, sizeof(NestState), s));
    HIPC(hipMemsetAsync(a.d_st.p, 0, sizeof(RtrState) * 2, s));
    HIPC(hipMemsetAsync(a.d_part.p, 0, sizeof(double) * PART_TOTAL, s));
  }
  // dense preconditioner  M = (Q + shift I)^-1
  if (t->d_tmp.alloc(2 * (size_t)N4 * N4)) { set_err("scratch allocation failed"); return DPGO_ERR; }
  double *A = t->d_tmp.p, *W = t->d_tmp.p + (size_t)N4 * N4;
  launch_bsr_to_dense(s, a.d_rowptr.p, a.d_col.p, a.d_qval.p, n, t->prm.precond_shift, A);
  const int fail = dense_spd_inverse(s, A, W, a.d_M.p, N4);
  if (fail != 0) { set_err("dense Cholesky of Q + shift I failed at pivot " + std::to_string(fail)); return DPGO_ERR; }

  // per-neighbour index tables for the packed-slab exchange (a7)
  size_t max_xfer = 1;
  for (int nb : a.neighbors) {
    const std::vector<int> fr = public_ids(a, nb);
    std::vector<int> slots;
    for (size_t q = 0; q < a.np.size(); ++q) if (a.np[q].first == nb) slots.push_back((int)q);
    auto &bf = a.d_pubframes[nb]; if (!bf) bf = std::make_unique<DevBuf<int>>();
    auto &bs = a.d_nbrslots[nb]; if (!bs) bs = std::make_unique<DevBuf<int>>();
    if (bf->upload(fr, s) || bs->upload(slots, s)) { set_err("index upload failed"); return DPGO_ERR; }
    a.n_pubframes[nb] = (int)fr.size(); a.n_nbrslots[nb] = (int)slots.size();
    max_xfer = std::max(max_xfer, std::max(fr.size(), slots.size()));
  }
  if (a.d_xfer.alloc(max_xfer * 4 * r)) { set_err("device allocation failed"); return DPGO_ERR; }

  AgentDev &d = a.dev;
  d.id = a.id; d.n = n; d.nb = (int)a.col.size(); d.N4 = N4;
  d.npub = a.npub; d.nshared = (int)se.size(); d.nnp = (int)a.np.size(); d.nedges = a.nedges;
  d.rowptr = a.d_rowptr.p; d.col = a.d_col.p; d.qval = a.d_qval.p; d.M = a.d_M.p;
  d.ell_w = EW; d.ell_col = a.d_ell_col.p; d.ell_val = a.d_ell_val.p;
  d.trowptr = a.d_trowptr.p; d.tcol = a.d_tcol.p; d.tval = a.d_tval.p; d.pub_index = a.d_pub_index.p;
  d.pub_pose = a.d_pub_pose.p; d.pub_ptr = a.d_pub_ptr.p; d.se = a.d_se.p; d.edges = a.d_edges.p;
  d.nbr[0] = a.d_nbr.p; d.nbr[1] = a.d_nbr.p + a.np.size() * 4 * r;
  for (int b = 0; b < NBUF; ++b) d.buf[b] = a.d_vec.p + len * b;
  d.part = a.d_part.p; d.st = a.d_st.p; d.nest = a.d_nest.p; d.scal = a.d_scal.p; d.resid = a.d_resid.p;
  a.data_dirty = false;
  t->descs_dirty = true;
  t->graph_valid = false;
  return 0;
}

int sync_descs(dpgo_team *t) {
  for (auto &a : t->ag) {
    const int rc = finalize_agent(t, *a);
    if (rc) return rc;
  }
  if (!t->descs_dirty) return 0;
  std::vector<AgentDev> descs;
  t->max_n = 0; t->max_npub = 0;
  for (auto &a : t->ag) {
    descs.push_back(a->dev);
    t->max_n = std::max(t->max_n, a->n);
    t->max_npub = std::max(t->max_npub, a->npub);
  }
  if (t->d_agents.upload(descs, t->stream)) { set_err("descriptor upload failed"); return DPGO_ERR; }
  if (t->sched.empty()) for (size_t k = 0; k < t->ag.size(); ++k) t->sched.push_back((int)k);
  if (t->d_sched.upload(t->sched, t->stream) || t->d_team.alloc(1)) { set_err("schedule upload failed"); return DPGO_ERR; }
  // greedy colouring of the (local) agent graph in index order: same colour = no shared edge
  const int na_ = (int)t->ag.size();
  if (!t->user_groups) {
  t->color_of.assign(na_, -1);
  t->groups.clear();
  for (int k = 0; k < na_; ++k) {
    std::vector<char> used(na_ + 1, 0);
    for (int nb : t->ag[k]->neighbors) {
      auto it = t->id2local.find(nb);
      if (it != t->id2local.end() && t->color_of[it->second] >= 0) used[t->color_of[it->second]] = 1;
    }
    int col = 0;
    while (used[col]) ++col;
    t->color_of[k] = col;
    if ((int)t->groups.size() <= col) t->groups.resize(col + 1);
    t->groups[col].push_back(k);
  }
  }
  std::vector<int> gptr(1, 0), gmem;
  for (auto &g : t->groups) { gmem.insert(gmem.end(), g.begin(), g.end()); gptr.push_back((int)gmem.size()); }
  if (t->d_group_ptr.upload(gptr, t->stream) || t->d_group_members.upload(gmem, t->stream)) { set_err("group upload failed"); return DPGO_ERR; }
  TeamDev td{};
  td.num_agents = (int)t->ag.size(); td.sched_len = (int)t->sched.size(); td.iter = t->iter;
  td.restart_interval = t->prm.restart_interval; td.sched = t->d_sched.p;
  td.group_ptr = t->d_group_ptr.p; td.group_members = t->d_group_members.p;
  HIPC(hipMemcpyAsync(t->d_team.p, &td, sizeof td, hipMemcpyHostToDevice, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  t->descs_dirty = false;
  t->graph_valid = false;
  return 0;
}

bool neighbor_poses_ready(const Agent &a, int aux) {
  for (char h : a.np_has[aux]) if (!h) return false;
  return true;
}

// ---- the local solve (QuadraticOptimizer::optimize), enqueued on the team stream -------------
// sel >= 0: that local agent (host-driven), sel == -1: device-selected (graph capture).
// RGD returns after enqueueing; the RTR path synchronises once per tCG chunk to read the
// device-side solver state.
//   fused: the iteration's tail (Nesterov V update, |X - XPrev|^2, end-of-iteration bookkeeping)
//          is folded into the RGD kernels (no restart in this iteration); `last` folds k_advance.
struct OptFlags { int aux = 0, pull = 0; bool capture = false, fused = false, last_advances = false; };

EvalOpts eval_opts(const dpgo_team *t, int gmode, int aux, int advance) {
  EvalOpts o;
  o.gmode = gmode; o.aux = aux; o.advance = advance;
  o.accel = t->prm.acceleration; o.num_robots = t->prm.num_robots; o.restart_interval = t->prm.restart_interval;
  return o;
}

double spmm_bytes_of(const dpgo_team *t, const Agent &a) {
  return 8.0 * (16.0 * a.col.size() + 3.0 * t->prm.r * 4 * a.n) + 4.0 * (a.col.size() + a.n + 1);
}

int enqueue_optimize(dpgo_team *t, int sel, const OptFlags &fl) {
  LaunchCtx c = t->ctx();
  const dpgo_params_t &p = t->prm;
  const int mn = (sel >= 0) ? t->ag[sel]->n : t->max_n;
  const int N4 = 4 * mn;
  const int gmode = fl.pull ? 2 : 1;
  if (p.method == DPGO_METHOD_RGD) {
    launch_eval(c, sel, mn, B_X, B_EGRAD, B_GF, PART_C, eval_opts(t, gmode, fl.aux, 0));
    if (fl.fused && p.rgd_use_preconditioner) {
      // K3: preconditioner + RGD step + Nesterov V + |dX|^2 (+ the team's end-of-iteration bookkeeping);
      // K5: f_opt / gradnorm_opt on the snapshot B_X2 that K3 leaves behind
      launch_precond(c, sel, mn, PM_RGD_, B_X, B_GF, B_Z, 0, 0, p.rgd_stepsize, p.acceleration, p.num_robots,
                     fl.last_advances ? 1 : 0, p.restart_interval);
      launch_eval(c, sel, mn, B_X2, B_EGRAD2, B_GF2, PART_A, eval_opts(t, 0, 0, 0));
    } else {
      int dirb = B_GF;
      if (p.rgd_use_preconditioner) {
        launch_precond(c, sel, mn, PM_PLAIN_, B_X, B_GF, B_Z, 0, 0, 0.0, 0, p.num_robots);
        dirb = B_Z;
      }
      launch_retract(c, sel, mn, B_X, dirb, -p.rgd_stepsize, B_X, -1);
      launch_eval(c, sel, mn, B_X, B_EGRAD, B_GF, PART_A, eval_opts(t, 0, 0, 0));
    }
    if (sel >= 0 && !fl.capture) {
      Agent &a = *t->ag[sel];
      if (p.rgd_use_preconditioner) { t->counters[0] += 1; t->counters[1] += 8.0 * N4 * (double)N4; }
      t->counters[2] += 2; t->counters[3] += 2 * spmm_bytes_of(t, a);
      a.opt_pending_rgd = true;
    }
    return 0;
  }
  if (fl.capture) { set_err("RTR cannot be captured"); return DPGO_ERR; }
  // ---- RTR: trust-region Newton with truncated CG; scalars stay on the device
  Agent &a = *t->ag[sel];
  launch_eval(c, sel, mn, B_X, B_EGRAD, B_GF, PART_A, eval_opts(t, gmode, fl.aux, 0));
  launch_rtr_begin(c, sel, p.rtr_initial_radius, p.gradnorm_tol, p.rtr_iterations);
  int sp = 0;
  RtrState *hs = t->h_state;
  auto read_state = [&]() -> int {
    HIPC(hipMemcpyAsync(hs, a.dev.st + sp, sizeof(RtrState), hipMemcpyDeviceToHost, t->stream));
    HIPC(hipStreamSynchronize(t->stream));
    return 0;
  };
  // One outer iteration = [tCG init, (Hess-vec, step) x J, retract, evaluate, accept].  Every kernel is
  // gated by the device-side phase, so whole patterns are enqueued blindly: the expected number of outer
  // iterations first, then one read-back; more patterns only if the state says the solve is not done.
  const int J = std::max(1, std::min(a.tcg_hint, p.rtr_tcg_iterations));
  auto pattern = [&]() {
    launch_precond(c, sel, mn, PM_TCG_INIT_, B_X, 0, 0, sp, p.rtr_tcg_iterations, 0.0, 0, p.num_robots); sp ^= 1;
    for (int q = 0; q < J; ++q) {
      launch_tcg_hv(c, sel, mn, sp, p.rtr_tcg_iterations); sp ^= 1;
      launch_precond(c, sel, mn, PM_TCG_STEP_, B_X, 0, 0, sp, p.rtr_tcg_iterations, 0.0, 0, p.num_robots); sp ^= 1;
    }
    launch_retract(c, sel, mn, B_X, B_ETA, 1.0, B_X2, sp);
    launch_rtr_eval2(c, sel, mn, sp);
    launch_rtr_accept(c, sel, mn, sp, p.gradnorm_tol, p.rtr_iterations, p.rtr_max_radius); sp ^= 1;
  };
  bool have_state = false;
  if (a.outer_hint == 0) {  // the previous solve of this agent started below the gradient tolerance
    if (read_state()) return DPGO_ERR;
    have_state = true;
  }
  if (!have_state || !hs->outer_done) {
    const int first = (a.outer_hint > 0) ? std::min(a.outer_hint, p.rtr_iterations) : p.rtr_iterations;
    for (int o = 0; o < first; ++o) pattern();
    if (read_state()) return DPGO_ERR;
    int guard = 0;
    while (!hs->outer_done && guard++ < 100000) {
      pattern();
      if (read_state()) return DPGO_ERR;
    }
  }
  a.outer_hint = hs->outer_count;
  if (hs->outer_count > 0) a.tcg_hint = std::max(2, std::min(8, (hs->tcg_total + hs->outer_count - 1) / hs->outer_count + 1));
  a.opt.success = 1;
  a.opt.f_init = hs->f_init; a.opt.gradnorm_init = hs->gn_init;
  a.opt.f_opt = hs->f1; a.opt.gradnorm_opt = hs->ngf;
  a.opt.rtr_outer_iters = hs->outer_count; a.opt.tcg_iters_total = hs->tcg_total;
  a.opt.hessvec_count = hs->hv_count; a.opt.precond_count = hs->pc_count; a.opt.accepted = hs->accepted;
  a.opt_pending_rgd = false;
  t->counters[0] += hs->pc_count; t->counters[1] += hs->pc_count * 8.0 * N4 * (double)N4;
  t->counters[2] += hs->hv_count + 1 + hs->outer_count;
  t->counters[3] += (hs->hv_count + 1 + hs->outer_count) * spmm_bytes_of(t, a);
  return 0;
}

// one PGOAgent::iterate for local agent `li` (host-driven variant used by the per-agent API)
int enqueue_iterate(dpgo_team *t, int li, int do_opt) {
  Agent &a = *t->ag[li];
  LaunchCtx c = t->ctx();
  const dpgo_params_t &p = t->prm;
  const bool restart = p.acceleration && ((a.iter + 2) % p.restart_interval) == 0;
  const bool fused = do_opt && p.method == DPGO_METHOD_RGD && p.rgd_use_preconditioner && !restart;
  OptFlags fl;
  fl.fused = fused;
  int rc = 0;
  a.rel_src = 0;
  if (p.acceleration) {
    launch_nest_pre(c, do_opt ? li : -2, li, 1, a.n, p.num_robots, p.restart_interval);
    if (do_opt) {
      fl.aux = 1;
      rc = enqueue_optimize(t, li, fl);
      if (rc) return rc;
      if (!fused) launch_nest_post(c, li, a.n, p.num_robots, p.restart_interval);
      if (restart) {
        fl.aux = 0;
        rc = enqueue_optimize(t, li, fl);
        if (rc) return rc;
        launch_nest_reset(c, li, a.n);
      }
      if (fused) a.rel_src = 1; else launch_status(c, li, li, 1, a.n);
    }
  } else {
    launch_copy(c, li, li, 1, a.n, B_X, B_XPREV, 0);
    if (do_opt) {
      rc = enqueue_optimize(t, li, fl);
      if (rc) return rc;
    }
    if (fused) a.rel_src = 1; else launch_status(c, li, li, 1, a.n);
  }
  launch_advance(c, li, 1, p.acceleration, p.num_robots, p.restart_interval, 0);
  return 0;
}

int fetch_scal(dpgo_team *t, Agent &a) {
  HIPC(hipMemcpyAsync(t->h_scal, a.dev.scal, sizeof(double) * 16, hipMemcpyDeviceToHost, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  return 0;
}

int refresh_rgd_result(dpgo_team *t, Agent &a) {
  if (!a.opt_pending_rgd) return 0;
  const int ppb = 64 / t->prm.r, nb = (a.n + ppb - 1) / ppb;
  std::vector<double> pc((size_t)PART_STRIDE * nb), pa((size_t)PART_STRIDE * nb);
  HIPC(hipStreamSynchronize(t->stream));
  HIPC(hipMemcpy(pc.data(), a.dev.part + PART_C, sizeof(double) * pc.size(), hipMemcpyDeviceToHost));
  HIPC(hipMemcpy(pa.data(), a.dev.part + PART_A, sizeof(double) * pa.size(), hipMemcpyDeviceToHost));
  auto sum = [&](const std::vector<double> &p, int off) { double s = 0; for (int i = 0; i < nb; ++i) s += p[(size_t)i * PART_STRIDE + off]; return s; };
  a.opt.success = 1;
  a.opt.f_init = sum(pc, 0); a.opt.gradnorm_init = std::sqrt(sum(pc, 1));
  a.opt.f_opt = sum(pa, 0); a.opt.gradnorm_opt = std::sqrt(sum(pa, 1));
  a.opt.rtr_outer_iters = 0; a.opt.tcg_iters_total = 0; a.opt.hessvec_count = 0;
  a.opt.precond_count = t->prm.rgd_use_preconditioner ? 1 : 0; a.opt.accepted = 1;
  a.opt_pending_rgd = false;
  return 0;
}

double robust_weight(const dpgo_params_t &p, double mu, double residual) {
  if (p.robust_cost_type == DPGO_COST_L2) return 1.0;
  const double r2 = residual * residual, b2 = p.gnc_barc * p.gnc_barc;
  const double upper = (mu + 1.0) / mu * b2, lower = mu / (mu + 1.0) * b2;
  if (r2 >= upper) return 0.0;
  if (r2 <= lower) return 1.0;
  return std::sqrt(b2 * mu * (mu + 1.0) / r2) - mu;
}

int compute_residuals(dpgo_team *t, Agent &a, std::vector<double> &res) {
  LaunchCtx c = t->ctx();
  launch_residuals(c, a.local, a.nedges);
  res.resize(a.nedges);
  if (a.nedges) HIPC(hipMemcpyAsync(res.data(), a.dev.resid, sizeof(double) * a.nedges, hipMemcpyDeviceToHost, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  return 0;
}

}  // namespace

// =================================================================================================
extern "C" {

const char *dpgo_last_error(void) { return g_err.c_str(); }

void dpgo_default_params(dpgo_params_t *p, int r, int num_robots) {
  std::memset(p, 0, sizeof *p);
  p->d = 3; p->r = r; p->num_robots = num_robots;
  p->method = DPGO_METHOD_RTR;
  p->rgd_stepsize = 1e-3; p->rgd_use_preconditioner = 1;       // launch/PGOAgent.launch:16-17
  p->rtr_iterations = 3; p->rtr_tcg_iterations = 50; p->gradnorm_tol = 1e-2;  // :18-20
  p->rtr_initial_radius = 100.0; p->rtr_max_radius = 500.0; p->precond_shift = 0.1;
  p->acceleration = 0; p->restart_interval = 50;                // :24-25
  p->rel_change_tol = 0.1; p->max_num_iters = 1000;             // :37-38
  p->robust_cost_type = DPGO_COST_L2;
  p->gnc_barc = 5.0; p->gnc_mu_step = 2.0; p->gnc_init_mu = 1e-5;
  p->robust_opt_num_weight_updates = 4; p->robust_opt_inner_iters = 10 * num_robots;
  p->robust_opt_min_convergence_ratio = 0.8;
  p->weights_as_float32 = 0;
}

dpgo_team_t *dpgo_team_create(int device, const dpgo_params_t *p, int num_local, const int *agent_ids, void *stream) {
  if (p->d != 3 || p->r < 3 || p->r > 8) { set_err("d must be 3 and r in [3,8]"); return nullptr; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_err("no HIP device: libdpgo_hip has no CPU fallback");
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) { set_err("hipSetDevice failed"); return nullptr; }
  auto *t = new dpgo_team();
  t->device = device; t->prm = *p;
  if (stream) t->stream = (hipStream_t)stream;
  else { if (hipStreamCreate(&t->stream) != hipSuccess) { delete t; set_err("hipStreamCreate failed"); return nullptr; } t->own_stream = true; }
  if (hipHostMalloc((void **)&t->h_states, sizeof(RtrState) * std::max(1, num_local)) != hipSuccess ||
      hipHostMalloc((void **)&t->h_state, sizeof(RtrState)) != hipSuccess ||
      hipHostMalloc((void **)&t->h_scal, sizeof(double) * 16) != hipSuccess) {
    delete t; set_err("pinned allocation failed"); return nullptr;
  }
  for (int k = 0; k < num_local; ++k) {
    auto a = std::make_unique<Agent>();
    a->id = agent_ids[k]; a->local = k; a->mu = p->gnc_init_mu;
    t->id2local[a->id] = k;
    t->ag.push_back(std::move(a));
  }
  return t;
}

void dpgo_team_destroy(dpgo_team_t *t) {
  if (!t) return;
  (void)hipSetDevice(t->device);
  (void)hipStreamSynchronize(t->stream);
  for (auto &g : t->graph) if (g) (void)hipGraphExecDestroy(g);
  t->ag.clear();
  if (t->h_state) (void)hipHostFree(t->h_state);
  if (t->h_states) (void)hipHostFree(t->h_states);
  if (t->h_scal) (void)hipHostFree(t->h_scal);
  if (t->own_stream) (void)hipStreamDestroy(t->stream);
  delete t;
}

int dpgo_team_num_local(const dpgo_team_t *t) { return (int)t->ag.size(); }
void *dpgo_team_stream(dpgo_team_t *t) { return (void *)t->stream; }
int dpgo_team_synchronize(dpgo_team_t *t) { HIPC(hipStreamSynchronize(t->stream)); return 0; }

int dpgo_agent_add_measurements(dpgo_team_t *t, int id, const dpgo_measurement_t *m, int count) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  for (int k = 0; k < count; ++k) {
    const dpgo_measurement_t &e = m[k];
    if (e.r1 == id && e.r2 == id) { if (e.p1 + 1 == e.p2) a->odom.push_back(e); else a->priv.push_back(e); }
    else if (e.r1 == id || e.r2 == id) a->shared.push_back(e);
    else continue;
    a->index_dirty = true;
    if (a->state == DPGO_WAIT_FOR_DATA) a->state = DPGO_WAIT_FOR_INITIALIZATION;
  }
  return 0;
}

int dpgo_agent_num_poses(dpgo_team_t *t, int id) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  rebuild_index(*a);
  return a->n;
}

int dpgo_agent_num_measurements(dpgo_team_t *t, int id, int *odom, int *priv, int *shared) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (odom) *odom = (int)a->odom.size();
  if (priv) *priv = (int)a->priv.size();
  if (shared) *shared = (int)a->shared.size();
  return (int)(a->odom.size() + a->priv.size() + a->shared.size());
}

int dpgo_agent_get_neighbors(dpgo_team_t *t, int id, int *ids) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  rebuild_index(*a);
  if (ids) std::copy(a->neighbors.begin(), a->neighbors.end(), ids);
  return (int)a->neighbors.size();
}

int dpgo_agent_public_pose_ids(dpgo_team_t *t, int id, int nbr, int *frames) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  auto f = public_ids(*a, nbr);
  if (frames) std::copy(f.begin(), f.end(), frames);
  return (int)f.size();
}

int dpgo_agent_neighbor_pose_ids(dpgo_team_t *t, int id, int nbr, int *frames) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  rebuild_index(*a);
  auto f = neighbor_ids(*a, nbr);
  if (frames) std::copy(f.begin(), f.end(), frames);
  return (int)f.size();
}

int dpgo_agent_set_X(dpgo_team_t *t, int id, const double *X) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  const size_t bytes = sizeof(double) * (size_t)t->prm.r * 4 * a->n;
  for (int b : {B_X, B_XPREV, B_Y, B_V}) HIPC(hipMemcpyAsync(a->dev.buf[b], X, bytes, hipMemcpyHostToDevice, t->stream));
  NestState ns{}; ns.iter = a->iter;
  HIPC(hipMemcpyAsync(a->dev.nest, &ns, sizeof ns, hipMemcpyHostToDevice, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  a->has_X = true;
  a->state = DPGO_INITIALIZED;
  return 0;
}

int dpgo_agent_get_X(dpgo_team_t *t, int id, int which, double *X) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (!a->has_X) return DPGO_NOT_READY;
  static const int map[4] = {B_X, B_Y, B_V, B_XPREV};
  if (which < 0 || which > 3) return DPGO_ERR;
  HIPC(hipMemcpyAsync(X, a->dev.buf[map[which]], sizeof(double) * (size_t)t->prm.r * 4 * a->n, hipMemcpyDeviceToHost,
                      t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  return 0;
}

int dpgo_agent_get_public_poses(dpgo_team_t *t, int id, int nbr, int aux, double *poses) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (!a->has_X) return DPGO_NOT_READY;
  if (sync_descs(t)) return DPGO_ERR;
  auto it = a->d_pubframes.find(nbr);
  if (it == a->d_pubframes.end()) return 0;
  const int cnt = a->n_pubframes[nbr];
  launch_pack(t->ctx(), a->dev.buf[aux ? B_Y : B_X], it->second->p, cnt, a->d_xfer.p);
  HIPC(hipMemcpyAsync(poses, a->d_xfer.p, sizeof(double) * (size_t)cnt * 4 * t->prm.r, hipMemcpyDeviceToHost, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  return 0;
}

int dpgo_agent_update_neighbor_poses(dpgo_team_t *t, int id, int nbr, int aux, int count, const int *frames,
                                     const double *poses) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  const size_t B = (size_t)4 * t->prm.r;
  // one staged upload + one scatter kernel instead of a copy per pose
  std::vector<int> slots;
  std::vector<double> packed;
  for (int k = 0; k < count; ++k) {
    const int q = find_np(*a, nbr, frames[k]);
    if (q < 0) continue;  // not an endpoint of any shared edge: dropped
    slots.push_back(q);
    packed.insert(packed.end(), poses + k * B, poses + (k + 1) * B);
    a->np_has[aux ? 1 : 0][q] = 1;
  }
  if (slots.empty()) return DPGO_OK;
  if (a->d_idx.alloc(slots.size()) || a->d_xfer.alloc(packed.size())) { set_err("device allocation failed"); return DPGO_ERR; }
  HIPC(hipMemcpyAsync(a->d_idx.p, slots.data(), sizeof(int) * slots.size(), hipMemcpyHostToDevice, t->stream));
  HIPC(hipMemcpyAsync(a->d_xfer.p, packed.data(), sizeof(double) * packed.size(), hipMemcpyHostToDevice, t->stream));
  launch_unpack(t->ctx(), a->dev.nbr[aux ? 1 : 0], a->d_idx.p, (int)slots.size(), a->d_xfer.p);
  HIPC(hipStreamSynchronize(t->stream));  // the host vectors above go out of scope
  return DPGO_OK;
}

int dpgo_agent_pack_public_poses_device(dpgo_team_t *t, int id, int nbr, int aux, double *dev_out) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (!a->has_X) return DPGO_NOT_READY;
  if (sync_descs(t)) return DPGO_ERR;
  auto it = a->d_pubframes.find(nbr);
  if (it == a->d_pubframes.end()) return 0;
  launch_pack(t->ctx(), a->dev.buf[aux ? B_Y : B_X], it->second->p, a->n_pubframes[nbr], dev_out);  // asynchronous
  return a->n_pubframes[nbr];
}

int dpgo_agent_unpack_neighbor_poses_device(dpgo_team_t *t, int id, int nbr, int aux, const double *dev_in) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  auto it = a->d_nbrslots.find(nbr);
  if (it == a->d_nbrslots.end()) return 0;
  for (size_t q = 0; q < a->np.size(); ++q) if (a->np[q].first == nbr) a->np_has[aux ? 1 : 0][q] = 1;
  launch_unpack(t->ctx(), a->dev.nbr[aux ? 1 : 0], it->second->p, a->n_nbrslots[nbr], dev_in);  // asynchronous
  return a->n_nbrslots[nbr];
}

int dpgo_agent_iterate(dpgo_team_t *t, int id, int do_optimization) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (a->state != DPGO_INITIALIZED || !a->has_X) { a->iter++; return DPGO_NOT_READY; }
  if (sync_descs(t)) return DPGO_ERR;
  if (t->prm.robust_cost_type != DPGO_COST_L2) a->robust_inner_iter++;
  bool opt = do_optimization != 0;
  a->last_success = true;
  if (opt && !neighbor_poses_ready(*a, t->prm.acceleration ? 1 : 0)) { opt = false; a->last_success = false; }
  const int rc = enqueue_iterate(t, a->local, opt ? 1 : 0);
  if (rc) return rc;
  a->iter++;
  if (t->prm.acceleration || opt) a->publish_requested = true;
  return a->last_success ? DPGO_OK : DPGO_NOT_READY;
}

int dpgo_agent_get_status(dpgo_team_t *t, int id, dpgo_status_t *s) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  s->agent_id = a->id; s->state = a->state; s->instance_number = a->instance; s->iteration_number = a->iter;
  s->relative_change = 0; s->ready_to_terminate = 0;
  if (!a->has_X || a->iter == 0 || a->rel_src == 2) { s->ready_to_terminate = a->has_X && a->iter > 0 && a->last_success; return DPGO_OK; }
  // |X - XPrev|^2 partials were left by the last kernel that moved X (fixed summation order)
  const int cnt = a->rel_src ? (4 * a->n + 7) / 8 : (a->n + 63) / 64;
  const int off = a->rel_src ? PART_B + 2 : PART_D;
  std::vector<double> part((size_t)cnt * PART_STRIDE);
  HIPC(hipMemcpyAsync(part.data(), a->dev.part + off, sizeof(double) * ((size_t)(cnt - 1) * PART_STRIDE + 1),
                      hipMemcpyDeviceToHost, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  double sum = 0;
  for (int k = 0; k < cnt; ++k) sum += part[(size_t)k * PART_STRIDE];
  s->relative_change = std::sqrt(sum / a->n);
  s->ready_to_terminate = a->last_success && (s->relative_change <= t->prm.rel_change_tol);
  return DPGO_OK;
}

int dpgo_agent_get_opt_result(dpgo_team_t *t, int id, dpgo_opt_result_t *r) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (refresh_rgd_result(t, *a)) return DPGO_ERR;
  *r = a->opt;
  return DPGO_OK;
}

int dpgo_agent_iteration_number(dpgo_team_t *t, int id) {
  Agent *a = find_agent(t, id);
  return a ? a->iter : DPGO_ERR;
}

int dpgo_agent_set_iteration_number(dpgo_team_t *t, int id, int iteration) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  a->iter = iteration;
  if (a->has_X) {
    NestState ns{};
    HIPC(hipMemcpy(&ns, a->dev.nest, sizeof ns, hipMemcpyDeviceToHost));
    ns.iter = iteration;
    HIPC(hipMemcpy(a->dev.nest, &ns, sizeof ns, hipMemcpyHostToDevice));
  }
  return DPGO_OK;
}

int dpgo_agent_publish_requested(dpgo_team_t *t, int id, int clear) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  const int v = a->publish_requested ? 1 : 0;
  if (clear) a->publish_requested = false;
  return v;
}

// ---- QuadraticProblem surface --------------------------------------------------------------------
int dpgo_agent_build_problem(dpgo_team_t *t, int id, int aux) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  if (!neighbor_poses_ready(*a, aux ? 1 : 0)) return DPGO_NOT_READY;
  launch_buildG(t->ctx(), a->local, a->npub, aux ? 1 : 0, 0);
  HIPC(hipStreamSynchronize(t->stream));
  return 0;
}

int dpgo_agent_eval(dpgo_team_t *t, int id, const double *X, double *f, double *egrad, double *rgrad) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  const size_t bytes = sizeof(double) * (size_t)t->prm.r * 4 * a->n;
  HIPC(hipMemcpyAsync(a->dev.buf[B_T0], X, bytes, hipMemcpyHostToDevice, t->stream));
  launch_eval(t->ctx(), a->local, a->n, B_T0, B_T1, B_T2, PART_A, eval_opts(t, 0, 0, 0));
  std::vector<double> part((size_t)PART_STRIDE * MAX_PART);
  HIPC(hipMemcpyAsync(part.data(), a->dev.part + PART_A, sizeof(double) * part.size(), hipMemcpyDeviceToHost, t->stream));
  if (egrad) HIPC(hipMemcpyAsync(egrad, a->dev.buf[B_T1], bytes, hipMemcpyDeviceToHost, t->stream));
  if (rgrad) HIPC(hipMemcpyAsync(rgrad, a->dev.buf[B_T2], bytes, hipMemcpyDeviceToHost, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  const int ppb = 64 / t->prm.r, nb = (a->n + ppb - 1) / ppb;
  double s = 0;
  for (int i = 0; i < nb; ++i) s += part[(size_t)i * PART_STRIDE];
  if (f) *f = s;
  return 0;
}

int dpgo_agent_hessvec(dpgo_team_t *t, int id, const double *X, const double *eta, double *out) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  const size_t bytes = sizeof(double) * (size_t)t->prm.r * 4 * a->n;
  HIPC(hipMemcpyAsync(a->dev.buf[B_T0], X, bytes, hipMemcpyHostToDevice, t->stream));
  HIPC(hipMemcpyAsync(a->dev.buf[B_X2], eta, bytes, hipMemcpyHostToDevice, t->stream));
  launch_eval(t->ctx(), a->local, a->n, B_T0, B_T1, B_T2, PART_A, eval_opts(t, 0, 0, 0));
  launch_hess(t->ctx(), a->local, a->n, B_T0, B_T1, B_X2, B_HETA, PART_A);
  HIPC(hipMemcpyAsync(out, a->dev.buf[B_HETA], bytes, hipMemcpyDeviceToHost, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  return 0;
}

int dpgo_agent_precondition(dpgo_team_t *t, int id, const double *X, const double *V, double *out) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  const size_t bytes = sizeof(double) * (size_t)t->prm.r * 4 * a->n;
  HIPC(hipMemcpyAsync(a->dev.buf[B_T0], X, bytes, hipMemcpyHostToDevice, t->stream));
  HIPC(hipMemcpyAsync(a->dev.buf[B_T1], V, bytes, hipMemcpyHostToDevice, t->stream));
  launch_precond(t->ctx(), a->local, a->n, PM_PLAIN_, B_T0, B_T1, B_T2, 0, 0, 0.0, 0, t->prm.num_robots);
  HIPC(hipMemcpyAsync(out, a->dev.buf[B_T2], bytes, hipMemcpyDeviceToHost, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  return 0;
}

int dpgo_agent_get_Q(dpgo_team_t *t, int id, int *rowptr, int *col, double *val) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  // read back from the device copy so the parity test sees what the kernels see
  if (rowptr) HIPC(hipMemcpy(rowptr, a->dev.rowptr, sizeof(int) * (a->n + 1), hipMemcpyDeviceToHost));
  if (col) HIPC(hipMemcpy(col, a->dev.col, sizeof(int) * a->col.size(), hipMemcpyDeviceToHost));
  if (val) HIPC(hipMemcpy(val, a->dev.qval, sizeof(double) * a->qval.size(), hipMemcpyDeviceToHost));
  return (int)a->col.size();
}

int dpgo_agent_get_G(dpgo_team_t *t, int id, double *G) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  HIPC(hipMemcpyAsync(G, a->dev.buf[B_G], sizeof(double) * (size_t)t->prm.r * 4 * a->n, hipMemcpyDeviceToHost, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  return 0;
}

// ---- manifold ops --------------------------------------------------------------------------------
static int raw_op(dpgo_team_t *t, int which, const double *X, const double *V, int n, double *out) {
  const size_t len = (size_t)t->prm.r * 4 * n;
  if (t->d_tmp.alloc(3 * len)) { set_err("scratch allocation failed"); return DPGO_ERR; }
  double *dX = t->d_tmp.p, *dV = dX + len, *dO = dV + len;
  HIPC(hipMemcpyAsync(dX, X, sizeof(double) * len, hipMemcpyHostToDevice, t->stream));
  if (V) HIPC(hipMemcpyAsync(dV, V, sizeof(double) * len, hipMemcpyHostToDevice, t->stream));
  LaunchCtx c = t->ctx();
  if (which == 0) launch_project_raw(c, dX, dO, n);
  else if (which == 1) launch_tangent_raw(c, dX, dV, dO, n);
  else launch_retract_raw(c, dX, dV, dO, n);
  HIPC(hipMemcpyAsync(out, dO, sizeof(double) * len, hipMemcpyDeviceToHost, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  return 0;
}
int dpgo_project_manifold(dpgo_team_t *t, const double *X, int n, double *out) { return raw_op(t, 0, X, nullptr, n, out); }
int dpgo_tangent_project(dpgo_team_t *t, const double *X, const double *V, int n, double *out) { return raw_op(t, 1, X, V, n, out); }
int dpgo_retract(dpgo_team_t *t, const double *X, const double *eta, int n, double *out) { return raw_op(t, 2, X, eta, n, out); }

// ---- robust path ---------------------------------------------------------------------------------
int dpgo_agent_compute_residual(dpgo_team_t *t, int id, const dpgo_measurement_t *m, double *residual) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (!a->has_X) return DPGO_NOT_READY;
  if (sync_descs(t)) return DPGO_ERR;
  int e = 0, found = -1;
  for (auto *vec : {&a->odom, &a->priv, &a->shared})
    for (auto &x : *vec) { if (x.r1 == m->r1 && x.p1 == m->p1 && x.r2 == m->r2 && x.p2 == m->p2) found = e; ++e; }
  if (found < 0) return DPGO_NOT_READY;
  if (m->r1 != a->id) { const int q = find_np(*a, m->r1, m->p1); if (q < 0 || !a->np_has[0][q]) return DPGO_NOT_READY; }
  if (m->r2 != a->id) { const int q = find_np(*a, m->r2, m->p2); if (q < 0 || !a->np_has[0][q]) return DPGO_NOT_READY; }
  std::vector<double> res;
  if (compute_residuals(t, *a, res)) return DPGO_ERR;
  *residual = res[found];
  return DPGO_OK;
}

double dpgo_agent_robust_weight(dpgo_team_t *t, int id, double residual) {
  Agent *a = find_agent(t, id);
  if (!a) return -1.0;
  return robust_weight(t->prm, a->mu, residual);
}

int dpgo_agent_update_measurement_weights(dpgo_team_t *t, int id) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  std::vector<double> res;
  if (a->has_X && compute_residuals(t, *a, res)) return DPGO_ERR;
  int e = (int)a->odom.size();
  if (a->has_X) {
    for (auto &m : a->priv) { if (!m.fixed_weight) m.weight = robust_weight(t->prm, a->mu, res[e]); ++e; }
    for (auto &m : a->shared) {
      const int other = (m.r1 == a->id) ? m.r2 : m.r1;
      bool ready = true;
      const int q = find_np(*a, other, (m.r1 == a->id) ? m.p2 : m.p1);
      if (q < 0 || !a->np_has[0][q]) ready = false;
      if (!m.fixed_weight && other > a->id && ready) m.weight = robust_weight(t->prm, a->mu, res[e]);
      ++e;
    }
  }
  a->weight_update_count++;
  a->mu *= t->prm.gnc_mu_step;
  a->robust_inner_iter = 0;
  a->data_dirty = true;
  if (sync_descs(t)) return DPGO_ERR;
  if (t->prm.acceleration && a->has_X) {
    launch_nest_reset(t->ctx(), a->local, a->n);
    NestState ns{}; ns.iter = a->iter;
    HIPC(hipMemcpyAsync(a->dev.nest, &ns, sizeof ns, hipMemcpyHostToDevice, t->stream));
    HIPC(hipStreamSynchronize(t->stream));
  }
  return DPGO_OK;
}

int dpgo_agent_set_measurement_weight(dpgo_team_t *t, int id, int r1, int p1, int r2, int p2, double w, int fixed) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  for (auto *vec : {&a->odom, &a->priv, &a->shared})
    for (auto &m : *vec)
      if (m.r1 == r1 && m.p1 == p1 && m.r2 == r2 && m.p2 == p2) { m.weight = w; m.fixed_weight = fixed; return DPGO_OK; }
  return DPGO_NOT_READY;
}

int dpgo_agent_get_measurements(dpgo_team_t *t, int id, dpgo_measurement_t *out) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  int c = 0;
  for (auto *vec : {&a->odom, &a->priv, &a->shared})
    for (auto &m : *vec) { if (out) out[c] = m; ++c; }
  return c;
}

int dpgo_agent_should_update_weights(dpgo_team_t *t, int id) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (t->prm.robust_cost_type == DPGO_COST_L2) return 0;
  if (a->weight_update_count >= t->prm.robust_opt_num_weight_updates) return 0;
  return a->robust_inner_iter >= t->prm.robust_opt_inner_iters;
}

int dpgo_agent_clear_data_matrices(dpgo_team_t *t, int id) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  a->data_dirty = true;
  return DPGO_OK;
}

double dpgo_error_threshold_at_quantile(double quantile, int dim) {
  // sqrt of the chi-square inverse CDF; regularised lower incomplete gamma by series + bisection
  auto P = [&](double x) {
    const double s = 0.5 * dim;
    if (x <= 0) return 0.0;
    double sum = 1.0 / s, term = 1.0 / s;
    for (int k = 1; k < 2000; ++k) { term *= x / (s + k); sum += term; if (term < 1e-17 * sum) break; }
    return std::exp(-x + s * std::log(x) - std::lgamma(s)) * sum;
  };
  double lo = 0, hi = 1000;
  for (int it = 0; it < 200; ++it) { const double mid = 0.5 * (lo + hi); if (P(0.5 * mid) < quantile) lo = mid; else hi = mid; }
  return std::sqrt(0.5 * (lo + hi));
}

// ---- team schedule -------------------------------------------------------------------------------
int dpgo_team_set_schedule(dpgo_team_t *t, const int *order, int len) {
  t->sched.clear();
  for (int k = 0; k < len; ++k) {
    auto it = t->id2local.find(order[k]);
    if (it == t->id2local.end()) { set_err("schedule names a non-local agent"); return DPGO_ERR; }
    t->sched.push_back(it->second);
  }
  t->descs_dirty = true;
  return 0;
}

int dpgo_team_set_initial(dpgo_team_t *t, const double *T, const double *YLift, const int *offsets) {
  const int r = t->prm.r;
  if (sync_descs(t)) return DPGO_ERR;
  for (auto &a : t->ag) {
    std::vector<double> X((size_t)r * 4 * a->n);
    dpgo_lift(T + (size_t)12 * offsets[a->local], a->n, YLift, r, X.data());
    const int rc = dpgo_agent_set_X(t, a->id, X.data());
    if (rc) return rc;
  }
  return dpgo_team_exchange_all(t);
}

int dpgo_team_exchange_all(dpgo_team_t *t) {
  if (sync_descs(t)) return DPGO_ERR;
  LaunchCtx c = t->ctx();
  for (auto &a : t->ag) {
    launch_pull(c, a->local, (int)a->shared.size());
    for (size_t q = 0; q < a->np.size(); ++q)
      if (t->id2local.count(a->np[q].first)) { a->np_has[0][q] = 1; a->np_has[1][q] = 1; }
  }
  HIPC(hipStreamSynchronize(t->stream));
  return 0;
}

// One global RBCD iteration over the agents of this team.
//   sel: local index of the agent that optimizes, -1 = device-selected (graph capture), -2 = the
//        selected agent lives on another rank (every local agent runs iterate(false)).
//   phase: 0 whole iteration; 1 = begin (everything before the neighbour exchange: Nesterov Y/X/V of all
//          local agents); 2 = end (local solve of `sel` + bookkeeping).
static int enqueue_team_iteration(dpgo_team_t *t, bool capture, bool restart, int sel, int phase) {
  LaunchCtx c = t->ctx();
  const dpgo_params_t &p = t->prm;
  const int na = (int)t->ag.size();
  const int mn = t->max_n;
  const bool fused = p.method == DPGO_METHOD_RGD && p.rgd_use_preconditioner && !restart && sel != -2;
  OptFlags fl;
  fl.pull = 1; fl.capture = capture; fl.fused = fused; fl.last_advances = fused;
  int rc = 0;
  if (phase != 2) {
    if (p.acceleration) launch_nest_pre(c, sel, -1, na, mn, p.num_robots, p.restart_interval);  // K1 (+ publishes cur_sel)
    else launch_copy(c, -3, -1, na, mn, B_X, B_XPREV, capture ? 1 : 0);
  }
  if (phase == 1) return 0;
  if (sel != -2) {
    fl.aux = p.acceleration ? 1 : 0;
    rc = enqueue_optimize(t, sel, fl);
    if (rc) return rc;
    if (!fused) {
      const int ns = (sel >= 0) ? t->ag[sel]->n : mn;
      if (p.acceleration) {
        launch_nest_post(c, sel, ns, p.num_robots, p.restart_interval);
        if (restart) {
          fl.aux = 0;
          rc = enqueue_optimize(t, sel, fl);
          if (rc) return rc;
          launch_nest_reset(c, sel, ns);
        }
      }
      launch_status(c, sel, -1, 1, ns);
    }
  }
  if (!fused) launch_advance(c, -1, na, p.acceleration, p.num_robots, p.restart_interval, 1);
  return 0;
}

// host-side bookkeeping after one global iteration in which local agent `sel` (or nobody: -2) optimized
static void account_iteration(dpgo_team_t *t, int sel, bool fused) {
  const dpgo_params_t &p = t->prm;
  for (auto &a : t->ag) {
    a->rel_src = p.acceleration ? 0 : 2;  // non-accelerated iterate(false) leaves X untouched
    a->iter += 1;
    if (p.robust_cost_type != DPGO_COST_L2) a->robust_inner_iter += 1;
    if (p.acceleration) a->publish_requested = true;
  }
  if (sel >= 0) {
    t->ag[sel]->rel_src = fused ? 1 : 0;
    t->ag[sel]->publish_requested = true;
  }
  t->iter += 1;
  t->counters[4] += 1;
}

int dpgo_team_step_begin(dpgo_team_t *t, int sel_id) {
  if (sync_descs(t)) return DPGO_ERR;
  auto it = t->id2local.find(sel_id);
  const int sel = (it == t->id2local.end()) ? -2 : it->second;
  const bool restart = t->prm.acceleration && ((t->iter + 2) % t->prm.restart_interval) == 0;
  return enqueue_team_iteration(t, false, restart, sel, 1);
}

int dpgo_team_step_end(dpgo_team_t *t, int sel_id) {
  auto it = t->id2local.find(sel_id);
  const int sel = (it == t->id2local.end()) ? -2 : it->second;
  const dpgo_params_t &p = t->prm;
  const bool restart = p.acceleration && ((t->iter + 2) % p.restart_interval) == 0;
  if (sel >= 0 && !neighbor_poses_ready(*t->ag[sel], p.acceleration ? 1 : 0)) { set_err("neighbour poses missing"); return DPGO_NOT_READY; }
  const int rc = enqueue_team_iteration(t, false, restart, sel, 2);
  if (rc) return rc;
  const bool fused = p.method == DPGO_METHOD_RGD && p.rgd_use_preconditioner && !restart && sel >= 0;
  account_iteration(t, sel, fused);
  return 0;
}

int dpgo_team_run(dpgo_team_t *t, int iters) {
  if (sync_descs(t)) return DPGO_ERR;
  const dpgo_params_t &p = t->prm;
  for (auto &a : t->ag) if (!a->has_X) { set_err("team_run before set_initial"); return DPGO_NOT_READY; }
  const bool graphable = (p.method == DPGO_METHOD_RGD) && p.rgd_use_preconditioner;
  if (graphable && !t->graph_valid) {
    // the schedule, the counters and the Nesterov scalars live on the device, so every iteration is
    // the same launch sequence: capture it 1/2/4/8/16 times to amortise the graph-launch gap
    for (int gi = 0; gi < dpgo_team::NGRAPH; ++gi) {
      hipGraph_t g = nullptr;
      HIPC(hipStreamBeginCapture(t->stream, hipStreamCaptureModeThreadLocal));
      int rc = 0;
      if (p.acceleration) {
        // 3 launches per iteration: [statistics of iteration k-1 + Nesterov step of iteration k] in one
        // heterogeneous kernel, cost/gradient (+ G from the neighbours' Y), preconditioner + RGD step +
        // Nesterov V + bookkeeping.  The first iteration has no statistics to close, the last one is closed
        // by a plain evaluation.
        LaunchCtx c = t->ctx();
        const int na = (int)t->ag.size(), mn = t->max_n;
        for (int rep = 0; rep < (1 << gi); ++rep) {
          if (rep == 0) launch_nest_pre(c, -1, -1, na, mn, p.num_robots, p.restart_interval);
          else launch_stats_nest(c, na, mn, p.num_robots, p.restart_interval);
          launch_eval(c, -1, mn, B_X, B_EGRAD, B_GF, PART_C, eval_opts(t, 2, 1, 0));
          launch_precond(c, -1, mn, PM_RGD_, B_X, B_GF, B_Z, 0, 0, p.rgd_stepsize, 1, p.num_robots, 1, p.restart_interval);
        }
        launch_eval(c, -5, mn, B_X2, B_EGRAD2, B_GF2, PART_A, eval_opts(t, 0, 0, 0));
      } else {
        for (int rep = 0; rep < (1 << gi) && !rc; ++rep) rc = enqueue_team_iteration(t, true, false, -1, 0);
      }
      HIPC(hipStreamEndCapture(t->stream, &g));
      if (rc) { (void)hipGraphDestroy(g); return rc; }
      if (t->graph[gi]) { (void)hipGraphExecDestroy(t->graph[gi]); t->graph[gi] = nullptr; }
      HIPC(hipGraphInstantiate(&t->graph[gi], g, nullptr, nullptr, 0));
      (void)hipGraphDestroy(g);
    }
    t->graph_valid = true;
  }
  auto account = [&](int sel) {
    const int n = t->ag[sel]->n, N4 = 4 * n;
    t->counters[0] += 1; t->counters[1] += 8.0 * N4 * (double)N4;
    t->counters[2] += 2; t->counters[3] += 2 * spmm_bytes_of(t, *t->ag[sel]);
  };
  int k = 0;
  while (k < iters) {
    const bool restart = p.acceleration && ((t->iter + 2) % p.restart_interval) == 0;
    int batch = 1;
    if (graphable && !restart) {
      // iterations until the next restart iteration (which runs un-captured)
      int until = iters - k;
      if (p.acceleration) {
        const int to_restart = (p.restart_interval - ((t->iter + 2) % p.restart_interval)) % p.restart_interval;
        until = std::min(until, to_restart == 0 ? 1 : to_restart);
      }
      int gi = 0;
      while (gi + 1 < dpgo_team::NGRAPH && (2 << gi) <= until) ++gi;
      batch = 1 << gi;
      HIPC(hipGraphLaunch(t->graph[gi], t->stream));
      for (auto &a : t->ag) a->rel_src = p.acceleration ? 0 : 2;
      for (int q = 0; q < batch; ++q) {
        const int sel = t->sched[(t->iter + q) % t->sched.size()];
        account(sel);
        if (q == batch - 1) { t->ag[sel]->opt_pending_rgd = true; t->ag[sel]->rel_src = 1; }
      }
    } else {
      const int sel = t->sched[t->iter % t->sched.size()];
      for (auto &a : t->ag) a->rel_src = p.acceleration ? 0 : 2;
      const int rc = enqueue_team_iteration(t, false, restart, sel, 0);
      if (rc) return rc;
      const bool fused = p.method == DPGO_METHOD_RGD && p.rgd_use_preconditioner && !restart;
      if (fused) t->ag[sel]->rel_src = 1;
    }
    t->iter += batch;
    for (auto &a : t->ag) { a->iter += batch; if (p.robust_cost_type != DPGO_COST_L2) a->robust_inner_iter += batch; }
    t->counters[4] += batch;
    k += batch;
  }
  return 0;
}

// ---- colour-parallel sweeps: the agents of one colour class share no edge, so their block
// updates commute; they run in the same launches (blockIdx.y = member) and the result equals the sequential
// schedule that visits the classes in order.  Non-accelerated RBCD only (the Nesterov scalars advance per
// global iteration and do not commute).
static int enqueue_optimize_group(dpgo_team_t *t, int g) {
  LaunchCtx c = t->ctx();
  const dpgo_params_t &p = t->prm;
  const std::vector<int> &mem = t->groups[g];
  c.ny = (int)mem.size();
  const int sel = SEL_GROUP0 - g;
  int mn = 0;
  for (int k : mem) mn = std::max(mn, t->ag[k]->n);
  if (p.method == DPGO_METHOD_RGD) {
    launch_eval(c, sel, mn, B_X, B_EGRAD, B_GF, PART_C, eval_opts(t, 2, 0, 0));
    int dirb = B_GF;
    if (p.rgd_use_preconditioner) { launch_precond(c, sel, mn, PM_PLAIN_, B_X, B_GF, B_Z, 0, 0, 0.0, 0, p.num_robots); dirb = B_Z; }
    launch_retract(c, sel, mn, B_X, dirb, -p.rgd_stepsize, B_X, -1);
    launch_eval(c, sel, mn, B_X, B_EGRAD, B_GF, PART_A, eval_opts(t, 0, 0, 0));
    for (int k : mem) {
      Agent &a = *t->ag[k];
      const double N4 = 4.0 * a.n;
      if (p.rgd_use_preconditioner) { t->counters[0] += 1; t->counters[1] += 8.0 * N4 * N4; }
      t->counters[2] += 2; t->counters[3] += 2 * spmm_bytes_of(t, a);
      a.opt_pending_rgd = true;
    }
    return 0;
  }
  launch_eval(c, sel, mn, B_X, B_EGRAD, B_GF, PART_A, eval_opts(t, 2, 0, 0));
  launch_rtr_begin(c, sel, p.rtr_initial_radius, p.gradnorm_tol, p.rtr_iterations);
  int sp = 0, J = 2;
  for (int k : mem) J = std::max(J, t->ag[k]->tcg_hint);
  J = std::min(J, p.rtr_tcg_iterations);
  auto pattern = [&]() {
    launch_precond(c, sel, mn, PM_TCG_INIT_, B_X, 0, 0, sp, p.rtr_tcg_iterations, 0.0, 0, p.num_robots); sp ^= 1;
    for (int q = 0; q < J; ++q) {
      launch_tcg_hv(c, sel, mn, sp, p.rtr_tcg_iterations); sp ^= 1;
      launch_precond(c, sel, mn, PM_TCG_STEP_, B_X, 0, 0, sp, p.rtr_tcg_iterations, 0.0, 0, p.num_robots); sp ^= 1;
    }
    launch_retract(c, sel, mn, B_X, B_ETA, 1.0, B_X2, sp);
    launch_rtr_eval2(c, sel, mn, sp);
    launch_rtr_accept(c, sel, mn, sp, p.gradnorm_tol, p.rtr_iterations, p.rtr_max_radius); sp ^= 1;
  };
  auto read_states = [&](bool &all_done) -> int {
    for (size_t q = 0; q < mem.size(); ++q)
      HIPC(hipMemcpyAsync(t->h_states + q, t->ag[mem[q]]->dev.st + sp, sizeof(RtrState), hipMemcpyDeviceToHost, t->stream));
    HIPC(hipStreamSynchronize(t->stream));
    all_done = true;
    for (size_t q = 0; q < mem.size(); ++q) all_done = all_done && t->h_states[q].outer_done;
    return 0;
  };
  bool done = false;
  for (int o = 0; o < p.rtr_iterations; ++o) pattern();
  if (read_states(done)) return DPGO_ERR;
  int guard = 0;
  while (!done && guard++ < 100000) {
    pattern();
    if (read_states(done)) return DPGO_ERR;
  }
  for (size_t q = 0; q < mem.size(); ++q) {
    Agent &a = *t->ag[mem[q]];
    const RtrState &hs = t->h_states[q];
    a.opt.success = 1;
    a.opt.f_init = hs.f_init; a.opt.gradnorm_init = hs.gn_init; a.opt.f_opt = hs.f1; a.opt.gradnorm_opt = hs.ngf;
    a.opt.rtr_outer_iters = hs.outer_count; a.opt.tcg_iters_total = hs.tcg_total;
    a.opt.hessvec_count = hs.hv_count; a.opt.precond_count = hs.pc_count; a.opt.accepted = hs.accepted;
    a.opt_pending_rgd = false;
    if (hs.outer_count > 0) a.tcg_hint = std::max(2, std::min(8, (hs.tcg_total + hs.outer_count - 1) / hs.outer_count + 1));
    const double N4 = 4.0 * a.n;
    t->counters[0] += hs.pc_count; t->counters[1] += hs.pc_count * 8.0 * N4 * N4;
    t->counters[2] += hs.hv_count + 1 + hs.outer_count;
    t->counters[3] += (hs.hv_count + 1 + hs.outer_count) * spmm_bytes_of(t, a);
  }
  return 0;
}

int dpgo_team_get_coloring(dpgo_team_t *t, int *color_of_agent) {
  if (sync_descs(t)) return DPGO_ERR;
  for (size_t k = 0; k < t->ag.size(); ++k) color_of_agent[k] = t->color_of[k];
  return (int)t->groups.size();
}

// explicit colour classes (global robot ids; ids that are not local are ignored) for teams that hold only part
// of the problem: the colouring must be global so that no two agents updated together share an edge
int dpgo_team_set_groups(dpgo_team_t *t, int num_groups, const int *group_ptr, const int *member_ids) {
  t->groups.assign(num_groups, {});
  t->color_of.assign(t->ag.size(), -1);
  for (int g = 0; g < num_groups; ++g)
    for (int q = group_ptr[g]; q < group_ptr[g + 1]; ++q) {
      auto it = t->id2local.find(member_ids[q]);
      if (it == t->id2local.end()) continue;
      t->groups[g].push_back(it->second);
      t->color_of[it->second] = g;
    }
  t->user_groups = true;
  t->descs_dirty = true;
  return 0;
}

// one colour class: `count` block updates of the global schedule (count = global size of the class)
int dpgo_team_run_group(dpgo_team_t *t, int g, int count) {
  if (sync_descs(t)) return DPGO_ERR;
  const dpgo_params_t &p = t->prm;
  if (p.acceleration) { set_err("colour-parallel sweeps need acceleration = 0"); return DPGO_ERR; }
  if (g < 0 || g >= (int)t->groups.size()) { set_err("bad group"); return DPGO_ERR; }
  LaunchCtx c = t->ctx();
  const int na = (int)t->ag.size();
  if (na == 0) return 0;
  launch_copy(c, -3, -1, na, t->max_n, B_X, B_XPREV, 0);
  if (!t->groups[g].empty()) {
    const int rc = enqueue_optimize_group(t, g);
    if (rc) return rc;
  }
  launch_status(c, -3, -1, na, t->max_n);
  launch_advance(c, -1, na, 0, p.num_robots, p.restart_interval, 1, count);
  for (auto &a : t->ag) { a->rel_src = 0; a->iter += count; if (p.robust_cost_type != DPGO_COST_L2) a->robust_inner_iter += count; }
  for (int k : t->groups[g]) t->ag[k]->publish_requested = true;
  t->iter += count;
  t->counters[4] += count;
  return 0;
}

int dpgo_team_run_colored(dpgo_team_t *t, int sweeps) {
  if (sync_descs(t)) return DPGO_ERR;
  const dpgo_params_t &p = t->prm;
  if (p.acceleration) { set_err("colour-parallel sweeps need acceleration = 0"); return DPGO_ERR; }
  for (auto &a : t->ag) if (!a->has_X) { set_err("run_colored before set_initial"); return DPGO_NOT_READY; }
  LaunchCtx c = t->ctx();
  const int na = (int)t->ag.size();
  for (int sw = 0; sw < sweeps; ++sw)
    for (size_t g = 0; g < t->groups.size(); ++g) {
      const int gs = (int)t->groups[g].size();
      launch_copy(c, -3, -1, na, t->max_n, B_X, B_XPREV, 0);
      const int rc = enqueue_optimize_group(t, (int)g);
      if (rc) return rc;
      launch_status(c, -3, -1, na, t->max_n);
      launch_advance(c, -1, na, 0, p.num_robots, p.restart_interval, 1, gs);
      for (auto &a : t->ag) { a->rel_src = 0; a->iter += gs; if (p.robust_cost_type != DPGO_COST_L2) a->robust_inner_iter += gs; }
      t->iter += gs;
      t->counters[4] += gs;
    }
  return 0;
}

int dpgo_team_iteration(dpgo_team_t *t) { return t->iter; }

int dpgo_team_cost(dpgo_team_t *t, double *f) {
  if (sync_descs(t)) return DPGO_ERR;
  LaunchCtx c = t->ctx();
  double total = 0;
  for (auto &a : t->ag) launch_pull(c, a->local, (int)a->shared.size());
  for (auto &a : t->ag) {
    launch_residuals(c, a->local, a->nedges);
    launch_cost(c, a->local);
  }
  for (auto &a : t->ag) {
    if (fetch_scal(t, *a)) return DPGO_ERR;
    total += t->h_scal[5];
  }
  *f = total;
  return 0;
}

int dpgo_team_update_weights(dpgo_team_t *t) {
  if (sync_descs(t)) return DPGO_ERR;
  if (dpgo_team_exchange_all(t)) return DPGO_ERR;
  int changed = 0;
  for (auto &a : t->ag) if (dpgo_agent_update_measurement_weights(t, a->id)) return DPGO_ERR;
  for (auto &a : t->ag)
    for (auto &m : a->shared) {
      const int other = (m.r1 == a->id) ? m.r2 : m.r1;
      if (other < a->id || !t->id2local.count(other)) continue;
      double w = m.weight;
      if (t->prm.weights_as_float32) w = (double)(float)w;
      if (dpgo_agent_set_measurement_weight(t, other, m.r1, m.p1, m.r2, m.p2, w, m.fixed_weight) == DPGO_OK) ++changed;
      t->ag[t->id2local[other]]->data_dirty = true;
    }
  if (sync_descs(t)) return DPGO_ERR;
  return changed;
}

int dpgo_team_get_counters(dpgo_team_t *t, double *out, int n) {
  for (int k = 0; k < n && k < 8; ++k) out[k] = t->counters[k];
  return 0;
}

int dpgo_agent_pull_local(dpgo_team_t *t, int id) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  launch_pull(t->ctx(), a->local, (int)a->shared.size());
  for (size_t q = 0; q < a->np.size(); ++q)
    if (t->id2local.count(a->np[q].first)) { a->np_has[0][q] = 1; a->np_has[1][q] = 1; }
  return 0;
}

// average duration of one launch of a hot kernel, HIP events on the team stream (roofline leg of bench.py)
int dpgo_team_time_kernel(dpgo_team_t *t, int id, int which, int reps, double *avg_ms, double *algorithmic_bytes) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  LaunchCtx c = t->ctx();
  const int n = a->n, r = t->prm.r;
  const double N4 = 4.0 * n;
  const double vec = 8.0 * r * 4 * n;
  auto launch = [&]() {
    if (which == 0) launch_precond(c, a->local, n, PM_PLAIN_, B_X, B_GF, B_T2, 0, 0, 0.0, 0, t->prm.num_robots);
    else if (which == 1) launch_eval(c, a->local, n, B_X, B_T1, B_T2, PART_C, eval_opts(t, 0, 0, 0));
    else if (which == 2) launch_hess(c, a->local, n, B_X, B_EGRAD, B_GF, B_T2, PART_C);
    else if (which == 3) launch_retract(c, a->local, n, B_X, B_GF, 0.0, B_X2, -1);
    else if (which == 4) launch_nest_pre(c, -2, -1, (int)t->ag.size(), t->max_n, t->prm.num_robots, 1 << 30);
    else if (which == 5) launch_noop(c, 1, 64);
    else if (which == 6) launch_noop(c, 256, 256);
    else if (which == 7) launch_status(c, -3, -1, (int)t->ag.size(), t->max_n);
    else launch_copy(c, -3, -1, (int)t->ag.size(), t->max_n, B_X, B_XPREV, 0);
  };
  if (which == 0) *algorithmic_bytes = 8.0 * N4 * N4 + 3.0 * vec;          // M once, v + X in, z out
  else *algorithmic_bytes = 8.0 * (16.0 * a->col.size() + 3.0 * r * 4 * n) + 4.0 * (a->col.size() + n + 1);
  hipEvent_t e0, e1;
  HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
  for (int k = 0; k < 3; ++k) launch();
  HIPC(hipEventRecord(e0, t->stream));
  for (int k = 0; k < reps; ++k) launch();
  HIPC(hipEventRecord(e1, t->stream));
  HIPC(hipEventSynchronize(e1));
  float ms = 0;
  HIPC(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *avg_ms = (double)ms / reps;
  return 0;
}

}  // extern "C"
