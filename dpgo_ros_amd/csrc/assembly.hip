// assembly.hip -- measurement lists -> block-CSR / ELL structure, data matrices Q and G, the dense
// preconditioner and the device descriptors of a team (PoseGraph of the reference: constructQ/constructG,
// SURVEY 8a rows 'PoseGraph data matrices').
#include <mutex>
#include "team_internal.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>

using namespace dpgo;

namespace dpgo_host {

// ---- the pool of device buffers behind DevBuf (team_internal.h)
namespace {
struct BufPool {
  std::mutex mu;
  std::map<std::pair<int, size_t>, std::vector<void *>> kept;  // (device, rounded bytes) -> idle buffers
  size_t held = 0;
};
BufPool &pool() { static BufPool *p = new BufPool; return *p; }  // (never destroyed: the HIP runtime may be gone by then)
constexpr size_t POOL_CAP = (size_t)4 << 30;
}  // namespace

size_t pool_round(size_t bytes) {
  // sizes rounded up to three significant bits (at most 12.5 % over), 256 bytes at least
  size_t b = std::max<size_t>(bytes, 256);
  int top = 63 - __builtin_clzll((unsigned long long)b);
  const size_t step = (size_t)1 << std::max(0, top - 3);
  return (b + step - 1) / step * step;
}

void *pool_take(size_t rounded, int dev) {
  BufPool &P = pool();
  std::lock_guard<std::mutex> g(P.mu);
  auto it = P.kept.find({dev, rounded});
  if (it == P.kept.end() || it->second.empty()) return nullptr;
  void *p = it->second.back();
  it->second.pop_back();
  P.held -= rounded;
  return p;
}

void pool_give(void *p, size_t rounded) {
  static const bool off = std::getenv("DPGO_NO_POOL") != nullptr;
  hipPointerAttribute_t at{};
  int dev = -1;
  if (!off && hipPointerGetAttributes(&at, p) == hipSuccess) dev = at.device;
  BufPool &P = pool();
  {
    std::lock_guard<std::mutex> g(P.mu);
    if (dev >= 0 && P.held + rounded <= POOL_CAP) {
      P.kept[{dev, rounded}].push_back(p);
      P.held += rounded;
      return;
    }
  }
  (void)hipFree(p);
}

size_t pool_flush(int dev) {
  BufPool &P = pool();
  std::vector<void *> drop;
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> g(P.mu);
    for (auto &kv : P.kept)
      if (kv.first.first == dev) {
        for (void *q : kv.second) { drop.push_back(q); bytes += kv.first.second; }
        kv.second.clear();
      }
    P.held -= bytes;
  }
  for (void *q : drop) (void)hipFree(q);  // (hipFree synchronises the device: whoever still had work on one of them is done)
  return bytes;
}

size_t pool_held(int dev) {
  BufPool &P = pool();
  std::lock_guard<std::mutex> g(P.mu);
  size_t bytes = 0;
  for (auto &kv : P.kept)
    if (kv.first.first == dev) bytes += kv.first.second * kv.second.size();
  return bytes;
}

namespace {
struct PinPool {
  std::mutex mu;
  std::map<std::pair<int, size_t>, std::vector<void *>> kept;  // (coherent, rounded bytes)
  size_t held = 0;
};
PinPool &pin_pool() { static PinPool *p = new PinPool; return *p; }
}  // namespace

void *pinned_take(size_t rounded, bool coherent) {
  PinPool &P = pin_pool();
  {
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.kept.find({coherent ? 1 : 0, rounded});
    if (it != P.kept.end() && !it->second.empty()) {
      void *p = it->second.back();
      it->second.pop_back();
      P.held -= rounded;
      return p;
    }
  }
  void *p = nullptr;
  if (hipHostMalloc(&p, rounded, coherent ? hipHostMallocCoherent : hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void pinned_give(void *p, size_t rounded, bool coherent) {
  static const bool off = std::getenv("DPGO_NO_POOL") != nullptr;
  PinPool &P = pin_pool();
  if (!off && rounded) {
    std::lock_guard<std::mutex> g(P.mu);
    if (P.held + rounded <= ((size_t)256 << 20)) {
      P.kept[{coherent ? 1 : 0, rounded}].push_back(p);
      P.held += rounded;
      return;
    }
  }
  (void)hipHostFree(p);
}

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
  // poses staged for the old numbering of the slots would land in the wrong ones: they are dropped with it (their
  // np_has flags are gone too, so the next iterate(true) waits for fresh ones)
  for (int s = 0; s < 2; ++s) { a.stage_slots[s].clear(); a.stage_data[s].clear(); a.stage_pos[s].assign(np.size(), -1); }
  a.neighbors.clear();
  for (auto &p : np) if (a.neighbors.empty() || a.neighbors.back() != p.first) a.neighbors.push_back(p.first);
  a.n = n;
  a.index_dirty = false;
  a.data_dirty = true;
  a.struct_uploaded = false;
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
  double TO[16], TOT[16], Om[16];
  if (a.struct_uploaded && (int)a.rowptr.size() == a.n + 1 && a.qval.size() == 16 * a.col.size()) {
    // same measurements, new weights (an UPDATE_WEIGHT round): the pattern stands, the blocks are accumulated in place --
    // in the order of the general path below, so the values are bitwise the same (the per-row maps were 1 ms of a round)
    std::fill(a.qval.begin(), a.qval.end(), 0.0);
    bool pattern_ok = true;  // (a block the stored pattern does not have: fall through to the general path below)
    auto add = [&](int row, int colm, const double *v, bool transpose, double sign) {
      if (row < 0 || row >= a.n) { pattern_ok = false; return; }
      int p = a.rowptr[row];
      const int pe = a.rowptr[row + 1];
      while (p < pe && a.col[p] != colm) ++p;
      if (p >= pe) { pattern_ok = false; return; }
      double *blk = a.qval.data() + (size_t)16 * p;
      for (int cp = 0; cp < 4; ++cp)
        for (int c = 0; c < 4; ++c) blk[cp + 4 * c] += sign * (transpose ? v[c + 4 * cp] : v[cp + 4 * c]);
    };
    for (int pass = 0; pass < 2; ++pass)
      for (auto &m : (pass ? a.priv : a.odom)) {
        edge_blocks(m, TO, TOT, Om);
        add(m.p1, m.p1, TOT, false, 1.0);
        add(m.p2, m.p2, Om, false, 1.0);
        add(m.p2, m.p1, TO, false, -1.0);
        add(m.p1, m.p2, TO, true, -1.0);
      }
    for (auto &m : a.shared) {
      edge_blocks(m, TO, TOT, Om);
      if (m.r1 == a.id) add(m.p1, m.p1, TOT, false, 1.0);
      else add(m.p2, m.p2, Om, false, 1.0);
    }
    if (pattern_ok) return;
    a.struct_uploaded = false;
  }
  std::vector<std::map<int, std::array<double, 16>>> rows(a.n);
  auto add = [&](int row, int colm, const double *v, bool transpose, double sign) {
    auto &blk = rows[row][colm];
    for (int cp = 0; cp < 4; ++cp)
      for (int c = 0; c < 4; ++c) blk[cp + 4 * c] += sign * (transpose ? v[c + 4 * cp] : v[cp + 4 * c]);
  };
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

// Which form of the preconditioner an agent runs.  Both exact forms are the reference's operator (Q + shift I)^-1:
//   * the two-level form (twolevel.h) where it streams less than half the bytes of the dense inverse (agents of a few
//     hundred poses and more) -- or for every agent when precond_mode asks for it;
//   * the dense inverse otherwise (N4^2 doubles for M and twice that as scratch while it is factored).
// Block-Jacobi (the inverted 4 x 4 diagonal blocks of Q + shift I; declared NOT the reference's preconditioner,
// include/dpgo_hip.h) runs only on request, or where neither exact form fits the device.
// `budget`: bytes of device memory still unclaimed by the agents decided before this one in the same set-up pass.
static int choose_precond(dpgo_team *t, Agent &a, double &budget) {
  const double N4 = 4.0 * a.n, need = 3.0 * 8.0 * N4 * N4;
  const double held = 8.0 * (double)a.d_M.n;  // an inverse this agent already holds is reused
  const bool fits = need - held + 512e6 <= budget;
  int mode = t->prm.precond_mode;
  if (mode != DPGO_PRECOND_AUTO && mode != DPGO_PRECOND_DENSE && mode != DPGO_PRECOND_BLOCK_JACOBI && mode != DPGO_PRECOND_TWO_LEVEL) {
    set_err("bad precond_mode");
    return DPGO_ERR;
  }
  double tl_need = 0;
  bool tl_ok = false;
  if (mode == DPGO_PRECOND_AUTO || mode == DPGO_PRECOND_TWO_LEVEL) {
    // the dissection depends on the sparsity pattern only: a weight update keeps it
    if (a.tl_plan.n != a.n || a.tl_rowptr != a.rowptr || a.tl_col != a.col) {
      // RTR: prefer a dissection whose slabs fit the one-launch solve (two workgroups per CU) over the one that streams
      // the fewest bytes per stand-alone apply
      const bool rtr = t->prm.method == DPGO_METHOD_RTR && t->use_fused_rtr;
      a.tl_plan = tl_make_plan(a.n, a.rowptr, a.col, t->tl_max_sub, rtr ? rtr_fused_tl_fit_pairs(t->prm.r) : 0, rtr ? std::min(512, 2 * t->num_cus) : 0);
      a.tl_rowptr = a.rowptr; a.tl_col = a.col;
      a.tl_plan_serial += 1;
    }
    const TLPlan &pl = a.tl_plan;
    double blocks = 16.0 * pl.ns * pl.ns, coup = 0;
    for (size_t i = 0; i < pl.sub.size(); ++i) {
      blocks += 16.0 * (double)pl.sub[i].size() * (double)pl.sub[i].size();
      coup += 16.0 * (double)pl.sub[i].size() * (double)pl.adj_sep[i].size();
    }
    tl_need = pl.bytes + 8.0 * (3.0 * blocks + coup) + 4.0 * (double)pl.nwg * (double)pl.n / std::max<size_t>(1, pl.sub.size()) * 2.0;
    tl_ok = !pl.sub.empty() && tl_need - 8.0 * (double)a.d_tl_slabs.n + 512e6 <= budget;
  }
  if (mode == DPGO_PRECOND_DENSE && !fits) {
    char msg[400];
    std::snprintf(msg, sizeof msg, "agent %d: the dense preconditioner of %d poses needs %.1f GB of device memory (%.1f GB for "
                  "the inverse, the rest while it is factored), %.1f GB are available; precond_mode = 0 (automatic) or 3 "
                  "(two-level) runs the same operator in a fraction of that", a.id, a.n, need / 1e9, need / 3e9, budget / 1e9);
    set_err(msg);
    return DPGO_ERR;
  }
  if (mode == DPGO_PRECOND_TWO_LEVEL && !tl_ok) {
    char msg[300];
    std::snprintf(msg, sizeof msg, "agent %d: the two-level preconditioner of %d poses needs %.1f GB of device memory, %.1f GB are "
                  "available", a.id, a.n, tl_need / 1e9, budget / 1e9);
    set_err(msg);
    return DPGO_ERR;
  }
  if (mode == DPGO_PRECOND_AUTO) {
    // RTR: an agent too large for the dense one-launch solve with two poses per workgroup (its 8-column slabs do not fit
    // LDS beyond 512 poses) keeps the solve in one launch with the two-level form, whose slabs do (rtr_fused.hip) -- worth
    // more than the cheaper single apply of the dense inverse.  (The dense solve with THREE poses per workgroup, 513 .. 640
    // poses, serves agents whose dense form is asked for: measured on the GNC torus3D schedule it is 8 % faster in all,
    // 701 against 761 ms, but an UPDATE_WEIGHT round costs 14.7 instead of 5.9 ms -- the automatic mode keeps two-level.)
    const bool rtr_tl = t->prm.method == DPGO_METHOD_RTR && t->use_fused_rtr && tl_ok &&
                        !(rtr_fused_np(t->prm.r, a.n, t->num_cus) == 2 && rtr_fused_lds_bytes(t->prm.r, a.n) <= (size_t)t->max_lds) &&
                        a.tl_plan.prod_post &&
                        rtr_fused_tl_eligible(t->prm.r, a.tl_plan.nwg - a.tl_plan.nS2, tl_max_pre_poses(a.tl_plan), a.tl_plan.ns, t->num_cus, t->max_lds);
    if (rtr_tl) mode = DPGO_PRECOND_TWO_LEVEL;
    else if (tl_ok && tl_worthwhile(a.tl_plan)) mode = DPGO_PRECOND_TWO_LEVEL;
    else if (fits) mode = DPGO_PRECOND_DENSE;
    else mode = tl_ok ? DPGO_PRECOND_TWO_LEVEL : DPGO_PRECOND_BLOCK_JACOBI;
  }
  if (mode == DPGO_PRECOND_DENSE) budget -= need - held;
  if (mode == DPGO_PRECOND_TWO_LEVEL) budget -= tl_need - 8.0 * (double)a.d_tl_slabs.n;
  a.precond = mode;
  return 0;
}

// upload structure + data matrices of one agent and assemble Q + shift I densely in `scratch` (2 N4^2 doubles: the
// matrix and the work area of its inversion, which sync_descs runs for all re-assembled agents at once)
int finalize_agent(dpgo_team *t, Agent &a, double *scratch) {
  rebuild_index(a);
  if (!a.data_dirty) return 0;
  const int r = t->prm.r, n = a.n, N4 = 4 * n;
  const size_t len = (size_t)r * 4 * n;
  hipStream_t s = t->stream;
  // (Q itself was assembled by sync_descs, which needs its pattern to choose the preconditioner)
  // shared edges sorted by local pose
  // A weight update (the measurement STRUCTURE stands: struct_uploaded) only refreshes what depends on the edges' values
  // -- the coefficients of the shared edges, the records of the residual kernel --: slots, sources, the sort, the public
  // pose tables and the chunk order are those of the last full build (0.5 ms of a round on the bench's GNC graph went
  // into rebuilding them, two binary searches per edge among it)
  const bool tables_cached = a.struct_uploaded && a.se_host.size() == a.shared.size() && a.se_order.size() == a.shared.size() &&
                             a.edges_host.size() == a.odom.size() + a.priv.size() + a.shared.size() &&
                             std::getenv("DPGO_HOST_LAYOUTS") == nullptr;
  std::vector<SharedEdgeDev> &se = a.se_host;  // (the neighbour-pose pointers are filled in by sync_descs once every agent's buffers exist)
  std::vector<EdgeDev> &edges = a.edges_host;
  double TO[16], TOT[16], Om[16];
  auto se_coef = [&](const dpgo_measurement_t &m, SharedEdgeDev &d) {
    edge_blocks(m, TO, TOT, Om);
    const bool out = (m.r1 == a.id);
    for (int cp = 0; cp < 4; ++cp)
      for (int c = 0; c < 4; ++c) d.coef[cp + 4 * c] = out ? TO[c + 4 * cp] : TO[cp + 4 * c];
  };
  auto edge_values = [&](const dpgo_measurement_t &m, EdgeDev &e) {
    std::memcpy(e.R, m.R, sizeof e.R);
    std::memcpy(e.t, m.t, sizeof e.t);
    e.kappa = m.kappa; e.tau = m.tau; e.weight = m.weight;
  };
  std::vector<int> pub_pose, pub_ptr, pose_eptr;
  if (tables_cached) {
    for (size_t e = 0; e < se.size(); ++e) se_coef(a.shared[a.se_order[e]], se[e]);
    size_t k = 0;
    for (auto &m : a.odom) edge_values(m, edges[k++]);
    for (auto &m : a.priv) edge_values(m, edges[k++]);
    for (auto &m : a.shared) edge_values(m, edges[k++]);
    a.dev.fe_code_ok = 0;
  } else {
  a.se_order.resize(a.shared.size());
  for (size_t e = 0; e < a.shared.size(); ++e) a.se_order[e] = (int)e;
  auto lpose_of = [&](int e) { const auto &m = a.shared[e]; return (m.r1 == a.id) ? m.p1 : m.p2; };
  std::stable_sort(a.se_order.begin(), a.se_order.end(), [&](int x, int y) { return lpose_of(x) < lpose_of(y); });
  se.clear();
  for (int src : a.se_order) {
    const auto &m = a.shared[src];
    const bool out = (m.r1 == a.id);
    SharedEdgeDev d{};
    d.lpose = out ? m.p1 : m.p2;
    const int nr = out ? m.r2 : m.r1, nf = out ? m.p2 : m.p1;
    d.slot = find_np(a, nr, nf);
    auto it = t->id2local.find(nr);
    d.src_agent_local = (it == t->id2local.end()) ? -1 : it->second;
    d.src_frame = nf;
    d.src_robot = nr;
    se_coef(m, d);
    se.push_back(d);
  }
  for (size_t e = 0; e < se.size(); ++e) {
    if (e == 0 || se[e].lpose != se[e - 1].lpose) { pub_pose.push_back(se[e].lpose); pub_ptr.push_back((int)e); }
  }
  pub_ptr.push_back((int)se.size());
  a.npub = (int)pub_pose.size();
  pose_eptr.assign(n + 1, 0);
  for (const auto &d : se) pose_eptr[d.lpose + 1] += 1;
  for (int j = 0; j < n; ++j) pose_eptr[j + 1] += pose_eptr[j];
  {
    const int ppb = 64 / t->prm.r;
    a.max_pose_edges = a.max_tile_edges = 0;
    for (int j = 0; j < n; ++j) a.max_pose_edges = std::max(a.max_pose_edges, pose_eptr[j + 1] - pose_eptr[j]);
    for (int j = 0; j < n; j += ppb) a.max_tile_edges = std::max(a.max_tile_edges, pose_eptr[std::min(n, j + ppb)] - pose_eptr[j]);
  }
  for (int g = 0; g < 5; ++g) a.dev.fe_eptr[g] = pub_ptr[std::min(64 * g, a.npub)];
  a.dev.fe_code_ok = 0;
  {
    // order of the 64-row chunks of the dense stream (dpgo_dev.h, fe_ord): private chunks first
    bool pub_chunk[32];
    for (int m = 0; m < 32; ++m) pub_chunk[m] = 16 * (m + 1) > n;  // (the chunk that holds the last rows, and every one beyond it)
    for (int p : pub_pose) if (p / 16 < 32) pub_chunk[p / 16] = true;
    int q = 0;
    for (int m = 0; m < 32; ++m) if (!pub_chunk[m]) a.dev.fe_ord[q++] = (unsigned char)m;
    a.dev.fe_npriv = q;
    for (int m = 0; m < 32; ++m) if (pub_chunk[m]) a.dev.fe_ord[q++] = (unsigned char)m;
    if (n > 512) { for (int m = 0; m < 32; ++m) a.dev.fe_ord[m] = (unsigned char)m; a.dev.fe_npriv = 0; }
  }
  // edge records for residual / cost evaluation
  edges.clear();
  auto push_edge = [&](const dpgo_measurement_t &m) {
    EdgeDev e{};
    e.i_local = (m.r1 == a.id) ? m.p1 : -1;
    e.j_local = (m.r2 == a.id) ? m.p2 : -1;
    e.i_slot = (m.r1 == a.id) ? -1 : find_np(a, m.r1, m.p1);
    e.j_slot = (m.r2 == a.id) ? -1 : find_np(a, m.r2, m.p2);
    edge_values(m, e);
    e.count_in_cost = (m.r1 == m.r2) ? 1 : (std::min(m.r1, m.r2) == a.id);
    edges.push_back(e);
  };
  for (auto &m : a.odom) push_edge(m);
  for (auto &m : a.priv) push_edge(m);
  for (auto &m : a.shared) push_edge(m);
  }
  a.nedges = (int)edges.size();

  // ELL (slot-major, width <= 8) + CSR tail copy of Q for the SpMM kernels
  int maxlen = 0;
  for (int j = 0; j < n; ++j) maxlen = std::max(maxlen, a.rowptr[j + 1] - a.rowptr[j]);
  const int EW = std::min(maxlen, 8);
  // A weight update (same pattern: struct_uploaded stands, build_Q cleared it otherwise) refreshes the VALUES of these
  // layouts on the device from the block-CSR values (k_q_layouts): the host laid out and uploaded 1.2 MB per agent and
  // round for what is a gather of the 0.3 MB it uploads anyway.
  const bool values_only = a.struct_uploaded && a.d_rowptr.p && a.d_trowptr.p && a.d_ell_val.p &&
                           a.d_ell_val.n >= (size_t)EW * n * 16 && a.d_qval.n >= a.qval.size() &&
                           std::getenv("DPGO_HOST_LAYOUTS") == nullptr;
  std::vector<int> ell_col, trowptr, tcol;
  std::vector<double> ell_val, tval;
  if (!values_only) {
  ell_col.assign((size_t)EW * n, 0); trowptr.assign(n + 1, 0);
  ell_val.assign((size_t)EW * n * 16, 0.0);
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
  }
  // rows without a tail also get the blocks in structure-of-arrays order, [slot][16-byte chunk][pose]: the one-launch
  // iteration (step_fused.hip) runs one lane per pose, and this way every load of its 64 lanes is one contiguous KB
  std::vector<double> soa_val;
  std::vector<int> soa_col;
  const int SW = std::max(EW, 5);
  const int tiles = std::max((n + 63) / 64, 8);  // (the one-launch iteration runs 8 waves whatever the agent's size)
  if (!values_only && maxlen <= 8 && maxlen > 0) {
    // [tile of 64 poses][slot][chunk][lane]: a wave's loads differ by compile-time offsets only
    soa_val.assign((size_t)tiles * SW * 8 * 64 * 2, 0.0);
    soa_col.resize((size_t)tiles * SW * 64);
    for (int jt = 0; jt < tiles * 64; ++jt) {
      const int tile = jt / 64, lane = jt % 64, j = std::min(jt, n - 1);
      const int p0 = a.rowptr[j], p1 = (jt < n) ? a.rowptr[j + 1] : p0;  // (lanes beyond the agent: zero blocks)
      for (int u = 0; u < SW; ++u) {
        soa_col[((size_t)tile * SW + u) * 64 + lane] = (p0 + u < p1) ? a.col[p0 + u] : j;
        if (p0 + u >= p1) continue;
        for (int q = 0; q < 8; ++q) {
          const size_t o = ((((size_t)tile * SW + u) * 8 + q) * 64 + lane) * 2;
          soa_val[o] = a.qval[(size_t)16 * (p0 + u) + 2 * q];
          soa_val[o + 1] = a.qval[(size_t)16 * (p0 + u) + 2 * q + 1];
        }
      }
    }
  }
  if (!values_only) a.has_soa = !soa_val.empty();
  std::vector<int> pub_index(n, -1);
  for (size_t q = 0; q < pub_pose.size(); ++q) pub_index[pub_pose[q]] = (int)q;
  // index arrays depend on the measurement STRUCTURE only: a weight update (same edges, new weights) re-sends values
  const bool idx = !a.struct_uploaded;
  if ((idx && (a.d_ell_col.upload(ell_col, s) || a.d_trowptr.upload(trowptr, s) || a.d_tcol.upload(tcol, s) ||
               (a.has_soa && a.d_soa_col.upload(soa_col, s)) ||
               a.d_pub_index.upload(pub_index, s) || a.d_pose_eptr.upload(pose_eptr, s))) ||
      (!values_only && (a.d_ell_val.upload(ell_val, s) || a.d_tval.upload(tval, s) || (a.has_soa && a.d_soa_val.upload(soa_val, s))))) {
    set_err("device allocation/upload failed");
    return DPGO_ERR;
  }
  const bool dense = a.precond == DPGO_PRECOND_DENSE;
  const bool fresh_vec = a.d_vec.n < len * NBUF;
  if (fresh_vec && a.exported && a.d_vec.p) {
    // other processes read this agent's poses in place through an IPC mapping of these arrays (dpgo_agent_export_state):
    // re-allocating them would leave the importers with a dangling mapping
    set_err("agent " + std::to_string(a.id) + ": its pose count grew after its arrays were exported over IPC; create the team "
            "with all measurements before dpgo_agent_export_state");
    return DPGO_ERR;
  }
  if ((idx && (a.d_rowptr.upload(a.rowptr, s) || a.d_col.upload(a.col, s) || a.d_pub_pose.upload(pub_pose, s) ||
               a.d_pub_ptr.upload(pub_ptr, s))) ||
      a.d_qval.upload(a.qval, s) || a.d_se.upload(se, s) ||
      a.d_edges.upload(edges, s) || a.d_vec.alloc(len * NBUF) || a.d_nbr.alloc(2 * a.np.size() * 4 * r) ||
      a.d_part.alloc(PART_TOTAL) || a.d_scal.alloc(16) || a.d_resid.alloc(edges.size()) || a.d_st.alloc(2) ||
      (dense && a.d_M.alloc((size_t)N4 * N4))) {
    set_err("device allocation/upload failed");
    return DPGO_ERR;
  }
  if (values_only)
    launch_q_layouts(s, a.d_rowptr.p, a.d_qval.p, n, EW, a.d_ell_val.p, a.d_trowptr.p, a.d_tval.p, SW, tiles,
                     a.has_soa ? a.d_soa_val.p : nullptr);
  if (fresh_vec) {
    HIPC(hipMemsetAsync(a.d_vec.p, 0, sizeof(double) * len * NBUF, s));
    HIPC(hipMemsetAsync(a.d_nbr.p, 0, sizeof(double) * a.d_nbr.n, s));
    HIPC(hipMemsetAsync(a.d_scal.p, 0, sizeof(double) * 16, s));
    HIPC(hipMemsetAsync(t->d_nest_all.p + a.local, 0, sizeof(NestState), s));
    HIPC(hipMemsetAsync(a.d_st.p, 0, sizeof(RtrState) * 2, s));
    HIPC(hipMemsetAsync(a.d_part.p, 0, sizeof(double) * PART_TOTAL, s));
  }
  if (dense) {
    // dense preconditioner  M = (Q + shift I)^-1: assembled here, inverted by sync_descs
    launch_bsr_to_dense(s, a.d_rowptr.p, a.d_col.p, a.d_qval.p, n, t->prm.precond_shift, scratch);
  } else if (a.precond == DPGO_PRECOND_BLOCK_JACOBI) {
    // block-Jacobi: invert the 4 x 4 diagonal blocks of Q + shift I on the host (Gauss-Jordan, SPD: no pivoting)
    std::vector<double> dinv((size_t)16 * n);
    for (int j = 0; j < n; ++j) {
      double m[4][8];
      for (int p = a.rowptr[j]; p < a.rowptr[j + 1]; ++p)
        if (a.col[p] == j)
          for (int i = 0; i < 4; ++i) for (int c = 0; c < 4; ++c) m[i][c] = a.qval[(size_t)16 * p + i + 4 * c] + (i == c ? t->prm.precond_shift : 0.0);
      for (int i = 0; i < 4; ++i) for (int c = 0; c < 4; ++c) m[i][4 + c] = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < 4; ++k) {
        const double pv = 1.0 / m[k][k];
        for (int c = 0; c < 8; ++c) m[k][c] *= pv;
        for (int i = 0; i < 4; ++i) if (i != k) { const double f = m[i][k]; for (int c = 0; c < 8; ++c) m[i][c] -= f * m[k][c]; }
      }
      for (int i = 0; i < 4; ++i) for (int c = 0; c < 4; ++c) dinv[(size_t)16 * j + i + 4 * c] = m[i][4 + c];
    }
    if (a.d_dinv.upload(dinv, s)) { set_err("device allocation/upload failed"); return DPGO_ERR; }
  }

  // per-neighbour index tables for the packed-slab exchange (a7)
  size_t max_xfer = 1;
  if (idx) {
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
  {
    std::vector<int> all;
    for (auto &kv : a.d_pubframes) { const std::vector<int> fr = public_ids(a, kv.first); all.insert(all.end(), fr.begin(), fr.end()); }
    a.n_pub_all = (int)all.size();
    if (a.d_pub_all.upload(all, s)) { set_err("index upload failed"); return DPGO_ERR; }
    // where each public pose goes in that packed list (k_iterate_false writes the report pose by pose)
    std::vector<int> pp_ptr(pub_pose.size() + 1, 0), pp;
    for (size_t q = 0; q < pub_pose.size(); ++q) {
      for (size_t p = 0; p < all.size(); ++p) if (all[p] == pub_pose[q]) pp.push_back((int)p);
      pp_ptr[q + 1] = (int)pp.size();
    }
    if (pp.empty()) pp.push_back(0);
    if (a.d_pubpos_ptr.upload(pp_ptr, s) || a.d_pubpos.upload(pp, s)) { set_err("index upload failed"); return DPGO_ERR; }
    max_xfer = std::max(max_xfer, 2 * all.size());
  }
  if (a.d_xfer.alloc(max_xfer * 4 * r)) { set_err("device allocation failed"); return DPGO_ERR; }
  }
  a.struct_uploaded = true;

  AgentDev &d = a.dev;
  d.id = a.id; d.n = n; d.nb = (int)a.col.size(); d.N4 = N4;
  d.npub = a.npub; d.nshared = (int)se.size(); d.nnp = (int)a.np.size(); d.nedges = a.nedges;
  d.rowptr = a.d_rowptr.p; d.col = a.d_col.p; d.qval = a.d_qval.p; d.M = dense ? a.d_M.p : nullptr;
  d.Dinv = a.precond == DPGO_PRECOND_BLOCK_JACOBI ? a.d_dinv.p : nullptr;
  if (a.precond != DPGO_PRECOND_TWO_LEVEL) d.tl = TLDev{};  // (two-level agents: filled in by tl_build)
  d.ell_w = EW; d.ell_col = a.d_ell_col.p; d.ell_val = a.d_ell_val.p;
  d.soa_val = a.has_soa ? a.d_soa_val.p : nullptr;
  d.soa_col = a.has_soa ? a.d_soa_col.p : nullptr;
  d.soa_w = a.has_soa ? std::max(EW, 5) : 0;
  d.trowptr = a.d_trowptr.p; d.tcol = a.d_tcol.p; d.tval = a.d_tval.p; d.pub_index = a.d_pub_index.p;
  d.pose_eptr = a.d_pose_eptr.p;
  d.pub_pose = a.d_pub_pose.p; d.pub_ptr = a.d_pub_ptr.p; d.se = a.d_se.p; d.edges = a.d_edges.p;
  d.nbr[0] = a.d_nbr.p; d.nbr[1] = a.d_nbr.p + a.np.size() * 4 * r;
  for (int b = 0; b < NBUF; ++b) d.buf[b] = a.d_vec.p + len * b;
  d.part = a.d_part.p; d.st = a.d_st.p; d.nest = t->d_nest_all.p + a.local; d.scal = a.d_scal.p; d.resid = a.d_resid.p;
  a.data_dirty = false;
  t->descs_dirty = true;
  t->graph_valid = false;
  return 0;
}

// staged neighbour poses (dpgo_agent_update_neighbor_poses) -> device slabs: the indices and poses of both sequences go
// into one pinned image that ONE kernel reads in place (no copy engine, no device-side staging buffer): the scatter
// kernel launched by flush_stage, or -- where nothing in front of it needs the poses -- the report kernel that closes a
// dpgo_agent_iterate (stage_to_pinned + launch_report)
int stage_to_pinned(dpgo_team *t, Agent &a, int *n0_out, int *n1_out) {
  const size_t B = (size_t)4 * t->prm.r;
  const size_t n0 = a.stage_slots[0].size(), n1 = a.stage_slots[1].size();
  *n0_out = (int)n0; *n1_out = (int)n1;
  if (n0 + n1 == 0) return 0;
  if (a.up_pending) {
    // the pinned image may still be owed to an earlier kernel (nothing has proved since that the stream moved past it):
    // wait before it is overwritten or re-allocated
    HIPC(hipStreamSynchronize(t->stream));
    for (auto &b : t->ag) b->up_pending = false;
  }
  if (a.h_up_idx.alloc(n0 + n1) || a.h_up.alloc((n0 + n1) * B)) { set_err("allocation failed"); return DPGO_ERR; }
  size_t off = 0;
  for (int aux = 0; aux < 2; ++aux) {
    const size_t cnt = a.stage_slots[aux].size();
    if (!cnt) continue;
    std::memcpy(a.h_up_idx.p + off, a.stage_slots[aux].data(), sizeof(int) * cnt);
    std::memcpy(a.h_up.p + off * B, a.stage_data[aux].data(), sizeof(double) * cnt * B);
    off += cnt;
    for (int q : a.stage_slots[aux]) a.stage_pos[aux][q] = -1;
    a.stage_slots[aux].clear();
    a.stage_data[aux].clear();
  }
  a.up_pending = true;
  return 0;
}

int flush_stage(dpgo_team *t) {
  for (auto &a : t->ag) {
    int n0 = 0, n1 = 0;
    if (stage_to_pinned(t, *a, &n0, &n1)) return DPGO_ERR;
    if (n0 + n1) launch_upload2(t->ctx(), a->dev.nbr[0], a->dev.nbr[1], a->h_up_idx.p, a->h_up.p, n0, n1);
  }
  return 0;
}

int sync_descs(dpgo_team *t) {
  const int rc = sync_descs_noflush(t);
  return rc ? rc : flush_stage(t);
}

int sync_descs_noflush(dpgo_team *t) {
  // agents whose data matrices changed: assemble each, then invert all their Q + shift I in one batch (the many
  // small dependent steps of the blocked inversions share their launches)
  {
    size_t total = 0;
    size_t free_b = 0, total_b = 0;
    // (the memory query is a driver call of several microseconds and this function opens every entry point of the
    // per-agent API: asked only when something is about to be built)
    bool will_build = false;
    for (auto &a : t->ag) will_build = will_build || a->index_dirty || a->data_dirty;
    if (will_build) (void)hipMemGetInfo(&free_b, &total_b);
    // (the scratch of an earlier pass is reused; idle pooled buffers count as used memory but are one pool_flush away from
    // free -- DevBuf::alloc gives them back when hipMalloc fails --, so the choice of preconditioner does not depend on what
    // earlier teams of the process happened to allocate)
    double budget = (double)free_b + 8.0 * (double)t->d_tmp.n + (will_build ? (double)pool_held(t->device) : 0.0);
    bool any_dirty = false;
    static const bool timing = std::getenv("DPGO_TIMING") != nullptr;
    const auto q0 = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    for (auto &a : t->ag) {
      rebuild_index(*a);
      if (!a->data_dirty) continue;
      any_dirty = true;
      build_Q(*a);
      if (choose_precond(t, *a, budget)) return DPGO_ERR;
      if (a->precond == DPGO_PRECOND_DENSE) total += 2 * (size_t)(4 * a->n) * (4 * a->n);
    }
    const double t_q = since(q0);
    const auto q1 = std::chrono::steady_clock::now();
    if (any_dirty) {
      if (total && t->d_tmp.alloc(total)) { set_err("scratch allocation failed"); return DPGO_ERR; }
      std::vector<double *> As, Ws, Ms;
      std::vector<int> Ns;
      std::vector<Agent *> tl_agents;
      size_t off = 0;
      for (auto &a : t->ag) {
        if (!a->data_dirty) continue;
        if (a->precond != DPGO_PRECOND_DENSE) {  // block-Jacobi: nothing to invert on the device; two-level: tl_build
          const int rc = finalize_agent(t, *a, nullptr);
          if (rc) return rc;
          if (a->precond == DPGO_PRECOND_TWO_LEVEL) tl_agents.push_back(a.get());
          continue;
        }
        const size_t NN = (size_t)(4 * a->n) * (4 * a->n);
        double *A = t->d_tmp.p + off, *W = A + NN;
        off += 2 * NN;
        const int rc = finalize_agent(t, *a, A);
        if (rc) return rc;
        As.push_back(A); Ws.push_back(W); Ms.push_back(a->d_M.p); Ns.push_back(4 * a->n);
      }
      const double t_fin = since(q1);
      const auto q2 = std::chrono::steady_clock::now();
      const int fail = dense_spd_inverse_batched(t->stream, (int)Ns.size(), As.data(), Ws.data(), Ms.data(), Ns.data());
      if (fail != 0) {
        set_err("dense Cholesky of Q + shift I failed at pivot " + std::to_string(fail & 0xffffff) + " (matrix " +
                std::to_string(fail >> 24) + " of the batch)");
        return DPGO_ERR;
      }
      // (the dense batch is done with the scratch: the two-level set-up reuses it)
      const double t_inv = since(q2);
      const auto q3 = std::chrono::steady_clock::now();
      if (tl_build(t, tl_agents)) return DPGO_ERR;
      if (timing)
        std::fprintf(stderr, "sync_descs: index + Q %.2f  finalize (tables, uploads) %.2f  dense inversions %.2f  two-level set-up %.2f ms\n",
                     t_q, t_fin, t_inv, since(q3));
      t->dense_max_n = 0;
      for (auto &a : t->ag) if (a->precond == DPGO_PRECOND_DENSE) t->dense_max_n = std::max(t->dense_max_n, a->n);
      // evaluations stage the shared edges' operands through LDS where a pose carries more than g_row_range's four at a
      // time (two dependent round trips per four edges there, two in all here) and the fullest tile fits
      {
        int mp = 0, mt = 0;
        for (auto &a : t->ag) { mp = std::max(mp, a->max_pose_edges); mt = std::max(mt, a->max_tile_edges); }
        const char *env = std::getenv("DPGO_STAGED_EVAL");  // (read when the structure is built: a per-team choice in the tests)
        const int min_edges = env ? std::atoi(env) : 5;  // DPGO_STAGED_EVAL=0: always, =1000000: never
        t->stage_cap = (mp >= min_edges && mt > 0 && eval_staged_lds_bytes(t->prm.r, mt) + EVS_STATIC_LDS <= (size_t)t->max_lds) ? mt : 0;
      }
      t->precond_of.clear();
      for (auto &a : t->ag) t->precond_of.push_back(a->precond);
      t->tl_max_wg = 0;
      for (auto &a : t->ag) if (a->precond == DPGO_PRECOND_TWO_LEVEL) t->tl_max_wg = std::max(t->tl_max_wg, a->tl_plan.nwg);
    }
  }
  if (!t->peers.empty())  // robots read in place stay "received" across a re-indexing of the neighbour slots
    for (auto &a : t->ag)
      for (size_t q = 0; q < a->np.size(); ++q)
        if (t->peers.count(a->np[q].first)) { a->np_has[0][q] = 1; a->np_has[1][q] = 1; }
  if (!t->descs_dirty) return 0;
  // direct pointers from every shared edge to the neighbour's pose (buffers of re-assembled agents may have moved)
  for (auto &a : t->ag) {
    if (a->se_host.empty() || !a->d_se.p) continue;
    const size_t B = (size_t)4 * t->prm.r;
    for (auto &d : a->se_host) {
      d.src[0] = d.src[1] = nullptr;
      d.src_yalt = nullptr;
      if (d.src_agent_local < 0 || t->isolated) {  // (isolated: co-resident neighbours are served by messages as well)
        // a neighbour in another process whose X / Y arrays were imported: read it in place, like a co-resident one
        auto pit = t->peers.find(d.src_robot);
        if (pit != t->peers.end() && d.src_frame >= 0 && d.src_frame < pit->second.n) {
          d.src[0] = pit->second.base + pit->second.off_x + (size_t)d.src_frame * B;
          d.src[1] = pit->second.base + pit->second.off_y + (size_t)d.src_frame * B;
        }
        continue;
      }
      const Agent &sa = *t->ag[d.src_agent_local];
      if (!sa.dev.buf[B_X] || d.src_frame < 0 || d.src_frame >= sa.n) continue;
      d.src[0] = sa.dev.buf[B_X] + (size_t)d.src_frame * B;
      d.src[1] = sa.dev.buf[B_Y] + (size_t)d.src_frame * B;
      d.src_yalt = sa.dev.buf[B_YALT] + (size_t)d.src_frame * B;
    }
    if (a->d_se.upload(a->se_host, t->stream)) { set_err("shared-edge upload failed"); return DPGO_ERR; }
    a->dev.se = a->d_se.p;
    {
      std::vector<double> packed(16 * a->se_host.size());
      for (size_t e = 0; e < a->se_host.size(); ++e) std::memcpy(&packed[16 * e], a->se_host[e].coef, 16 * sizeof(double));
      if (a->d_fe_coef.upload(packed, t->stream)) { set_err("shared-edge upload failed"); return DPGO_ERR; }
      a->dev.fe_coef = a->d_fe_coef.p;
    }
    // (carried one-launch iteration: the same pointers as 16-bit codes in the descriptor)
    a->dev.fe_code_ok = (int)a->se_host.size() <= FE_MAX_EDGES && !t->isolated;
    for (int q = 0; q < FE_MAX_EDGES / 2; ++q) a->dev.fe_code[q] = 0;
    for (size_t e = 0; e < a->se_host.size() && a->dev.fe_code_ok; ++e) {
      const auto &d = a->se_host[e];
      if (d.src_agent_local < 0 || d.src_agent_local >= LOOKAHEAD_MAX_AGENTS || d.src_frame < 0 || d.src_frame >= 4096 || !d.src_yalt) {
        a->dev.fe_code_ok = 0;
        break;
      }
      a->dev.fe_code[e >> 1] |= (unsigned)(d.src_frame | (d.src_agent_local << 12)) << (16 * (e & 1));
    }
  }
  std::vector<AgentDev> descs;
  t->max_n = 0; t->max_npub = 0;
  for (auto &a : t->ag) {
    descs.push_back(a->dev);
    t->max_n = std::max(t->max_n, a->n);
    t->max_npub = std::max(t->max_npub, a->npub);
  }
  if (t->d_agents.upload(descs, t->stream)) { set_err("descriptor upload failed"); return DPGO_ERR; }
  t->h_descs = descs;
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
  // one more class behind the colouring: every local agent (simultaneous updates, dpgo_team_run_simultaneous)
  t->all_group = (int)t->groups.size();
  for (int k = 0; k < na_; ++k) gmem.push_back(k);
  gptr.push_back((int)gmem.size());
  if (t->d_group_ptr.upload(gptr, t->stream) || t->d_group_members.upload(gmem, t->stream)) { set_err("group upload failed"); return DPGO_ERR; }
  TeamDev td{};
  td.num_agents = (int)t->ag.size(); td.sched_len = (int)t->sched.size(); td.iter = t->iter;
  td.restart_interval = t->prm.restart_interval; td.sched = t->d_sched.p;
  td.group_ptr = t->d_group_ptr.p; td.group_members = t->d_group_members.p;
  for (int k = 0, acc = 0; k <= LOOKAHEAD_MAX_AGENTS; ++k) {
    td.pose_prefix[k] = acc;
    if (k < (int)t->ag.size()) acc += t->ag[k]->n;
  }
  HIPC(hipMemcpyAsync(t->d_team.p, &td, sizeof td, hipMemcpyHostToDevice, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  t->descs_dirty = false;
  t->graph_valid = false;
  return 0;
}

}  // namespace dpgo_host
