// capi.hip -- the extern "C" entry points of libdpgo_hip.so (contract: include/dpgo_hip.h).
#include <unistd.h>

#include <atomic>
#include <array>
#include <chrono>
#include <thread>
#include <unordered_map>

#include <mutex>
#include <random>
#include "team_internal.h"

using namespace dpgo;
using namespace dpgo_host;

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
  p->tls_threshold = 10.0; p->huber_threshold = 3.0;
  p->robust_opt_num_weight_updates = 4; p->robust_opt_inner_iters = 10 * num_robots;
  p->robust_opt_min_convergence_ratio = 0.8;
  p->weights_as_float32 = 0;
  p->robust_opt_num_resets = 0;   // launch/PGOAgent.launch:33
  p->precond_mode = DPGO_PRECOND_AUTO;
  p->status_every_iterate = 0;
  p->rgd_line_search = 0; p->rgd_ls_max_backoffs = 7; p->rgd_ls_shrink = 0.5; p->rgd_ls_sigma = 1e-4;
}

// library-owned streams of finished teams are kept (drained) for the next team of the same device
namespace {
struct StreamPool { std::mutex mu; std::map<int, std::vector<hipStream_t>> kept; };
StreamPool &stream_pool() { static StreamPool *p = new StreamPool; return *p; }
hipStream_t stream_take(int device) {
  StreamPool &P = stream_pool();
  {
    std::lock_guard<std::mutex> g(P.mu);
    auto &v = P.kept[device];
    if (!v.empty()) { hipStream_t s = v.back(); v.pop_back(); return s; }
  }
  hipStream_t s = nullptr;
  return hipStreamCreate(&s) == hipSuccess ? s : nullptr;
}
void stream_give(int device, hipStream_t s) {
  StreamPool &P = stream_pool();
  std::lock_guard<std::mutex> g(P.mu);
  auto &v = P.kept[device];
  if (v.size() < 16) v.push_back(s); else (void)hipStreamDestroy(s);
}
}  // namespace

dpgo_team_t *dpgo_team_create(int device, const dpgo_params_t *p, int num_local, const int *agent_ids, void *stream) {
  if (p->d != 3 || p->r < 3 || p->r > 8) { set_err("d must be 3 and r in [3,8]"); return nullptr; }
  if (p->robust_opt_num_resets < 0) { set_err("robust_opt_num_resets must be >= 0"); return nullptr; }
  if (p->robust_cost_type < DPGO_COST_L2 || p->robust_cost_type > DPGO_COST_GNC_TLS) {
    set_err("robust_cost_type: one of DPGO_COST_L2 / L1 / HUBER / TLS / GM / GNC_TLS");
    return nullptr;
  }
  if ((p->robust_cost_type == DPGO_COST_TLS && !(p->tls_threshold > 0.0)) || (p->robust_cost_type == DPGO_COST_HUBER && !(p->huber_threshold > 0.0))) {
    set_err("tls_threshold / huber_threshold must be > 0");
    return nullptr;
  }
  if (p->rgd_line_search && (p->rgd_ls_max_backoffs < 0 || p->rgd_ls_max_backoffs >= LS_MAX_TRIALS || !(p->rgd_ls_shrink > 0.0) ||
                             !(p->rgd_ls_shrink < 1.0) || !(p->rgd_ls_sigma > 0.0) || !(p->rgd_ls_sigma < 1.0))) {
    set_err("rgd_line_search: rgd_ls_max_backoffs in [0, 7], rgd_ls_shrink and rgd_ls_sigma in (0, 1)");
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_err("no HIP device: libdpgo_hip has no CPU fallback");
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) { set_err("hipSetDevice failed"); return nullptr; }
  auto *t = new dpgo_team();
  t->device = device; t->prm = *p;
  if (stream) t->stream = (hipStream_t)stream;
  else { t->stream = stream_take(device); if (!t->stream) { delete t; set_err("hipStreamCreate failed"); return nullptr; } t->own_stream = true; }
  // (one pinned block for the team's small read-back areas, from the pool of team_internal.h)
  {
    const size_t nl = (size_t)std::max(1, num_local);
    const size_t b_states = sizeof(RtrState) * nl, b_state = sizeof(RtrState), b_scal = sizeof(double) * 16 * nl;
    auto up = [](size_t b) { return (b + 255) / 256 * 256; };
    t->h_block_bytes = pool_round(up(b_states) + up(b_state) + up(b_scal) + 256);
    t->h_block = (char *)pinned_take(t->h_block_bytes, false);
    if (!t->h_block) { delete t; set_err("pinned allocation failed"); return nullptr; }
    std::memset(t->h_block, 0, t->h_block_bytes);
    t->h_states = (RtrState *)t->h_block;
    t->h_state = (RtrState *)(t->h_block + up(b_states));
    t->h_scal = (double *)(t->h_block + up(b_states) + up(b_state));
    t->h_bar_err = (int *)(t->h_block + up(b_states) + up(b_state) + up(b_scal));
  }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) t->num_cus = cus;
    int lds = 0;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0) t->max_lds = lds;
    const char *e3 = std::getenv("DPGO_BAKE_SEL");
    if (e3) t->bake_sel = (e3[0] == '0') ? 0 : 1;
    const char *e4 = std::getenv("DPGO_BAKE_DESC");
    if (e4) t->bake_desc = (e4[0] == '0') ? 0 : 1;
    const char *e2 = std::getenv("DPGO_FUSED_RTR");
    if (e2) t->use_fused_rtr = (e2[0] == '0') ? 0 : 1;
    const char *e5 = std::getenv("DPGO_FUSED_EVAL");
    if (e5) t->use_fused_eval = (e5[0] == '0') ? 0 : 1;
    if (const char *e6 = std::getenv("DPGO_FE_MIN_N")) t->fe_min_n = std::max(32, std::atoi(e6));
    if (const char *e7 = std::getenv("DPGO_FE_CARRY")) t->use_fe_carry = (e7[0] == '0') ? 0 : 1;
    if (const char *e8 = std::getenv("DPGO_FE_DEEP")) t->use_fe_deep = (e8[0] == '0') ? 0 : 1;
    if (const char *e9 = std::getenv("DPGO_FE_PERSIST")) t->use_fe_persist = (e9[0] == '1') ? 1 : 0;
    if (const char *e10 = std::getenv("DPGO_REPORT_TAIL")) t->use_report_tail = (e10[0] == '0') ? 0 : 1;
    if (const char *e11 = std::getenv("DPGO_REPORT_PREFETCH")) t->prefetch_reports = (e11[0] == '0') ? 0 : 1;
    if (t->d_nest_all.alloc(3 * std::max(1, num_local)) ||
        hipMemset(t->d_nest_all.p, 0, sizeof(NestState) * 3 * std::max(1, num_local)) != hipSuccess) {
      delete t; set_err("hand-off state allocation failed"); return nullptr;
    }
    *t->h_bar_err = 0;
  }
  for (int k = 0; k < num_local; ++k) {
    auto a = std::make_unique<Agent>();
    a->id = agent_ids[k]; a->local = k; a->mu = p->gnc_init_mu;
    t->id2local[a->id] = k;
    t->ag.push_back(std::move(a));
  }
  return t;
}

static void close_iteration_log(dpgo_team_t *t);

void dpgo_team_destroy(dpgo_team_t *t) {
  if (!t) return;
  (void)hipSetDevice(t->device);
  (void)hipStreamSynchronize(t->stream);
  close_iteration_log(t);
  release_fused_rtr_lock(t);
  if (t->rtr_lock_fd >= 0) { ::close(t->rtr_lock_fd); t->rtr_lock_fd = -1; }
  for (auto &kv : t->graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
  for (auto &kv : t->peers) if (kv.second.base) (void)hipIpcCloseMemHandle(kv.second.base);
  for (void *p : t->mail_handles) if (p) (void)hipIpcCloseMemHandle(p);
  t->ag.clear();
  if (t->h_block) pinned_give(t->h_block, t->h_block_bytes, false);
  if (t->own_stream) stream_give(t->device, t->stream);  // (drained above)
  delete t;
}

// An in-kernel exchange that timed out (codes: 2 hand-off of the one-launch RTR solve, 3 two-level preconditioner, 4 mailbox
// of the device-side UPDATE token) leaves its code in a pinned word.  Every entry point that has just drained the team's
// stream looks at it, so a time-out is an error at the next host read-back whichever call that is (advisor, round 3: only
// dpgo_team_synchronize did, and the run_peer path never called it).
static int check_exchange_error(dpgo_team_t *t) {
  if (!t->h_bar_err || !*t->h_bar_err) return 0;
  const int code = *t->h_bar_err;
  *t->h_bar_err = 0;
  for (auto &kv : t->graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
  t->graphs.clear(); t->graph_flip.clear();
  set_err("an in-kernel exchange timed out (code " + std::to_string(code) + ": 2 hand-off of the one-launch RTR solve, 3 two-level "
          "preconditioner, 4 mailbox of the device-side UPDATE token, 5 hand-off of the persistent RGD launch): the iterates since the last "
          "successful synchronisation are invalid");
  return DPGO_ERR;
}

int dpgo_team_num_local(const dpgo_team_t *t) { return (int)t->ag.size(); }
void *dpgo_team_stream(dpgo_team_t *t) { return (void *)t->stream; }
// hipStreamSynchronize parks the calling thread and pays its wake-up (tens of microseconds, more than a 20-iteration
// graph's launch): the stream is polled first, for as long as a short run takes, and only a long wait blocks
static hipError_t stream_wait(hipStream_t s) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipStreamQuery(s);
    if (e != hipErrorNotReady) return e;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) return hipStreamSynchronize(s);
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
}

int dpgo_team_synchronize(dpgo_team_t *t) {
  HIPC(stream_wait(t->stream));
  release_fused_rtr_lock(t);
  for (auto &a : t->ag) if (a->opt_pending_rtr && refresh_rtr_result(t, *a)) return DPGO_ERR;
  return check_exchange_error(t);
}

int dpgo_agent_add_measurements(dpgo_team_t *t, int id, const dpgo_measurement_t *m, int count) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (count < 0 || (count > 0 && !m)) { set_err("add_measurements: bad arguments"); return DPGO_ERR; }
  // indices arrive in network messages in the ROS setting: reject anything that would index outside the pose
  // arrays (or size the dense preconditioner absurdly) before a single record is stored
  for (int k = 0; k < count; ++k) {
    const dpgo_measurement_t &e = m[k];
    if (e.r1 != id && e.r2 != id) continue;
    if (e.p1 < 0 || e.p2 < 0 || e.p1 > DPGO_MAX_POSE_INDEX || e.p2 > DPGO_MAX_POSE_INDEX) {
      set_err("add_measurements: pose index out of range in measurement " + std::to_string(k));
      return DPGO_ERR;
    }
    if (e.r1 < 0 || e.r2 < 0 || e.r1 >= t->prm.num_robots || e.r2 >= t->prm.num_robots) {
      set_err("add_measurements: robot id outside [0, num_robots) in measurement " + std::to_string(k));
      return DPGO_ERR;
    }
    if (e.r1 == e.r2 && e.p1 == e.p2) { set_err("add_measurements: self loop in measurement " + std::to_string(k)); return DPGO_ERR; }
  }
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
  ++t->epoch;
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
  release_fused_rtr_lock(t);  // (the stream has drained: another team of this device may take the one-launch solve)
  return check_exchange_error(t);
}

static int finish_report(dpgo_team_t *t, Agent *a);

int dpgo_agent_get_public_poses(dpgo_team_t *t, int id, int nbr, int aux, double *poses) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (!a->has_X) return DPGO_NOT_READY;
  if (sync_descs_noflush(t)) return DPGO_ERR;  // (reads this agent's own poses: staged neighbour poses stay staged)
  if (a->d_pubframes.find(nbr) == a->d_pubframes.end()) return 0;
  if (a->rep.pending && finish_report(t, a)) return DPGO_ERR;  // (the report of an iterate(false) carries these poses)
  if (a->pub_epoch != t->epoch) {
    // the wrapper asks per neighbour and per sequence (:666-668), right after an iterate: pack everything this agent
    // publishes (all neighbours, X and Y) into one buffer, ONE copy and ONE synchronisation; later calls are served
    // from the host copy until the team enqueues device work again
    const size_t B = (size_t)4 * t->prm.r;
    const size_t total = 2 * (size_t)a->n_pub_all;
    if (a->d_xfer.alloc(total * B)) { set_err("device allocation failed"); return DPGO_ERR; }
    launch_pack2(t->ctx(), a->dev.buf[B_X], a->dev.buf[B_Y], a->d_pub_all.p, a->n_pub_all, a->d_xfer.p);
    std::vector<double> host(total * B);
    HIPC(hipMemcpyAsync(host.data(), a->d_xfer.p, sizeof(double) * total * B, hipMemcpyDeviceToHost, t->stream));
    HIPC(hipStreamSynchronize(t->stream));
    size_t off = 0;
    for (auto &kv : a->d_pubframes) {
      const size_t cnt = (size_t)a->n_pubframes[kv.first];
      for (int s = 0; s < 2; ++s)
        a->pub_cache[s][kv.first].assign(host.begin() + ((size_t)s * a->n_pub_all + off) * B,
                                         host.begin() + ((size_t)s * a->n_pub_all + off + cnt) * B);
      off += cnt;
    }
    a->pub_epoch = t->epoch;
    a->pub_pinned = false;
  }
  if (a->pub_pinned) {
    // layout of k_report / k_iterate_false: [8 scalars][X of every neighbour's public frames, neighbour order][Y likewise]
    const size_t B = (size_t)4 * t->prm.r;
    size_t off = 0;
    for (auto &kv : a->d_pubframes) {
      if (kv.first == nbr) break;
      off += (size_t)a->n_pubframes[kv.first];
    }
    const double *src = a->h_down.p + 8 + ((size_t)((aux && !a->pub_one_seq) ? 1 : 0) * a->n_pub_all + off) * B;
    std::memcpy(poses, src, sizeof(double) * (size_t)a->n_pubframes[nbr] * B);
    return 0;
  }
  const std::vector<double> &c = a->pub_cache[aux ? 1 : 0][nbr];
  std::copy(c.begin(), c.end(), poses);
  return 0;
}

int dpgo_agent_update_neighbor_poses(dpgo_team_t *t, int id, int nbr, int aux, int count, const int *frames,
                                     const double *poses) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (count < 0 || (count > 0 && (!frames || !poses))) { set_err("update_neighbor_poses: bad arguments"); return DPGO_ERR; }
  if (sync_descs_noflush(t)) return DPGO_ERR;
  const size_t B = (size_t)4 * t->prm.r;
  // staged on the host: the wrapper delivers one message per neighbour and sequence (:1276-1278); everything that
  // arrived is uploaded with one copy and one scatter kernel when the poses are next used (flush_stage)
  const int s = aux ? 1 : 0;
  for (int k = 0; k < count; ++k) {
    const int q = find_np(*a, nbr, frames[k]);
    if (q < 0) continue;  // not an endpoint of any shared edge: dropped
    // a pose delivered twice before it is used keeps its latest value (one scatter target per slot)
    if (a->stage_pos[s].size() != a->np.size()) a->stage_pos[s].assign(a->np.size(), -1);
    size_t at;
    if (a->stage_pos[s][q] >= 0) at = (size_t)a->stage_pos[s][q];
    else {
      at = a->stage_slots[s].size();
      a->stage_pos[s][q] = (int)at;
      a->stage_slots[s].push_back(q);
      a->stage_data[s].resize((at + 1) * B);
    }
    std::copy(poses + k * B, poses + (k + 1) * B, a->stage_data[s].begin() + at * B);
    a->np_has[s][q] = 1;
  }
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

// finish_report: wait for the sequence word of the agent's outstanding report and move what it carries into the host copies
static int finish_report(dpgo_team_t *t, Agent *a) {
  if (!a->rep.pending) return DPGO_OK;
  const Agent::PendingReport rp = a->rep;
  a->rep.pending = false;
  const size_t npub = 2 * (size_t)a->n_pub_all * 4 * t->prm.r;
  {
    volatile unsigned long long *flag = reinterpret_cast<volatile unsigned long long *>(a->h_down.p);
    unsigned long long spins = 0;
    auto resync = [&]() {  // a report that never arrived must not leave the host one ahead of the device for good
      a->report_seq = 0;
      *flag = 0ull;
      (void)hipMemsetAsync(a->d_report_seq.p, 0, 2 * sizeof(unsigned long long), t->stream);
    };
    while (*flag < rp.expect) {
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#else
      std::this_thread::yield();
#endif
      if (++spins > (1ull << 26)) {  // (seconds: something is wrong -- let the runtime say what)
        const hipError_t se = hipStreamSynchronize(t->stream);
        if (se != hipSuccess || *flag < rp.expect) {
          resync();
          set_err(se != hipSuccess ? std::string("report kernel failed: ") + hipGetErrorString(se)
                                   : std::string("report kernel did not deliver (sequence word not written)"));
          return DPGO_ERR;
        }
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    // the image was written over PCIe: none of it is in this core's caches, and the getters that follow copy it 8 KB at a
    // time (one dependent stream of misses each).  Ask for all of its lines now, side by side.
    if (t->prefetch_reports) {
      const char *img = reinterpret_cast<const char *>(a->h_down.p);
      const size_t bytes = sizeof(double) * (8 + (rp.one_seq ? npub / 2 : npub));
      for (size_t o = 64; o < bytes; o += 64) __builtin_prefetch(img + o, 0, 2);
    }
  }
  // diagnostics (dpgo_team_get_counters [5..6]): host time between the launch of a report and its arrival (us), reports
  t->counters[5] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - rp.t_launch).count();
  t->counters[6] += 1;
  for (auto &b : t->ag) b->up_pending = false;  // this team's stream has drained past every upload enqueued before
  release_fused_rtr_lock(t);                    // ... and past any one-launch solve (also when it timed out: the device lock
                                                // must not stay with a team that is about to report an error)
  if (check_exchange_error(t)) return DPGO_ERR;
  if (a->opt_pending_rtr && refresh_rtr_result(t, *a, true)) return DPGO_ERR;  // (its record is in pinned memory already)
  const double *out = a->h_down.p;
  if (npub) {
    // the getters answer straight from the pinned image while it is current (nothing enqueued on this team since the
    // report's launch; a later report would rewrite it): no second host copy
    a->pub_epoch = rp.epoch;
    a->pub_pinned = true;
    a->pub_one_seq = rp.one_seq;
  }
  if (rp.want_status) {
    a->opt_rel_change = std::sqrt(out[1] / a->n);
    a->opt_cached = true;
  }
  if (rp.want_opt) {
    a->opt.success = 1;
    a->opt.f_init = out[2]; a->opt.gradnorm_init = std::sqrt(out[3]);
    a->opt.f_opt = out[4]; a->opt.gradnorm_opt = std::sqrt(out[5]);
    a->opt.rtr_outer_iters = 0; a->opt.tcg_iters_total = 0; a->opt.hessvec_count = 0;
    a->opt.precond_count = t->prm.rgd_use_preconditioner ? 1 : 0; a->opt.accepted = 1; a->opt.ls_backoffs = 0;
    a->opt_pending_rgd = false;
    if (read_ls_record(t, *a)) return DPGO_ERR;  // (line search: back-offs / accepted from the agent's scalars)
  }
  return DPGO_OK;
}

// Everything the wrapper asks for right after an iterate -- the public poses of all neighbours and both sequences
// (:666-668), the status of the block update (:616) and the result of a local RGD solve (:169-172) -- is written by ONE
// kernel behind the iterate's launches straight into pinned host memory, followed by a sequence word; the host polls
// that word (no copy engine, no stream-wide wait) and the getters then answer from the host copies.
// wait = false (iterate(false), src/PGOAgentROS.cpp:1183-1186): the report is only ENQUEUED -- nothing the wrapper reads
// right after iterate(false) (publishStatus: the status of the last iterate(true), the iteration number) comes from it;
// the first getter that needs its payload (get*SharedPoseDictWithNeighbor from runOnce, :109-113) waits for it, by which
// time it has usually landed.
// the pinned image and the sequence words of an agent's reports
static int report_prepare(dpgo_team_t *t, Agent *a) {
  const size_t npub = 2 * (size_t)a->n_pub_all * 4 * t->prm.r;
  // (an earlier report nobody asked for is superseded: this one rewrites the same pinned image behind it on the stream;
  // the image may only be re-allocated once the stream is past the earlier kernel)
  if (a->rep.pending && 8 + npub > a->h_down.n && finish_report(t, a)) return DPGO_ERR;
  a->rep.pending = false;
  if (a->h_down.alloc(8 + npub, true)) { set_err("pinned allocation failed"); return DPGO_ERR; }
  if (!a->d_report_seq.p) {
    if (a->d_report_seq.alloc(2)) { set_err("device allocation failed"); return DPGO_ERR; }
    HIPC(hipMemsetAsync(a->d_report_seq.p, 0, 2 * sizeof(unsigned long long), t->stream));
    a->report_seq = 0;
    *reinterpret_cast<volatile unsigned long long *>(a->h_down.p) = 0ull;
  }
  return DPGO_OK;
}

static int report_after_iterate(dpgo_team_t *t, Agent *a, bool did_opt, bool advance, bool upload, bool one_seq = false,
                                bool whole_iterate_false = false, bool wait = true) {
  const bool want_status = did_opt && !t->prm.status_every_iterate && (a->opt_rel_src == 1 || a->opt_rel_src == 5);
  const bool tiles = a->opt_rel_src == 5;
  const int scnt = want_status ? (tiles ? (a->n + 63) / 64 : precond_nblk(*a)) : 0;
  const int ppb = 64 / t->prm.r, nb = (a->n + ppb - 1) / ppb;
  const bool want_opt = did_opt && a->opt_pending_rgd;
  if (report_prepare(t, a)) return DPGO_ERR;
  int up0 = 0, up1 = 0;
  if (upload && stage_to_pinned(t, *a, &up0, &up1)) return DPGO_ERR;
  const unsigned long long expect = ++a->report_seq;
  const auto tq0 = std::chrono::steady_clock::now();
  if (whole_iterate_false) {
    launch_iterate_false(t->ctx(), a->local, a->n, t->prm.num_robots, t->prm.restart_interval, a->d_pubpos_ptr.p, a->d_pubpos.p,
                         a->h_down.p, a->d_report_seq.p, a->d_report_seq.p + 1, a->h_up_idx.p, a->h_up.p, up0, up1);
  } else
  launch_report(t->ctx(), a->local, a->d_pub_all.p, a->n_pub_all, a->h_down.p, tiles ? PART_E : PART_B + 2, scnt, PART_STRIDE,
                want_opt ? nb : 0, a->d_report_seq.p, advance ? 1 : 0, t->prm.acceleration, t->prm.num_robots,
                t->prm.restart_interval, a->h_up_idx.p, a->h_up.p, up0, up1, one_seq ? 1 : 0);
  a->rep.pending = true; a->rep.expect = expect; a->rep.epoch = t->epoch; a->rep.one_seq = one_seq;
  a->rep.want_status = want_status; a->rep.want_opt = want_opt; a->rep.t_launch = tq0;
  return wait ? finish_report(t, a) : DPGO_OK;
}

int dpgo_agent_iterate(dpgo_team_t *t, int id, int do_optimization) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (a->state != DPGO_INITIALIZED || !a->has_X) { a->iter++; return DPGO_NOT_READY; }
  // iterate(false) reads no neighbour pose: what is staged on the host is not uploaded in front of it
  const bool defer_upload = !do_optimization && t->prm.acceleration && t->peers.empty();
  // an accelerated iterate(true) opens with k_nest_pre, which scatters this agent's staged poses itself
  const bool upload_in_first = do_optimization && t->prm.acceleration && t->peers.empty();
  if ((defer_upload || upload_in_first) ? sync_descs_noflush(t) : sync_descs(t)) return DPGO_ERR;
  if (upload_in_first) {
    int n0 = 0, n1 = 0;
    if (stage_to_pinned(t, *a, &n0, &n1)) return DPGO_ERR;
    t->pend_up.slots = a->h_up_idx.p; t->pend_up.in = a->h_up.p; t->pend_up.n0 = n0; t->pend_up.n1 = n1;
  }
  if (t->prm.robust_cost_type != DPGO_COST_L2) a->robust_inner_iter++;
  bool opt = do_optimization != 0;
  a->last_success = true;
  if (opt && !neighbor_poses_ready(*a, t->prm.acceleration ? 1 : 0)) { opt = false; a->last_success = false; }
  // a report closes this call whenever there is something to publish or a result to read: its kernel then also takes
  // the end-of-iterate bookkeeping (one launch less)
  const bool will_report = t->prm.acceleration || opt || a->publish_requested;
  // an accelerated iterate(false) is ONE launch (k_iterate_false): Nesterov step, staged poses, bookkeeping and report
  const bool one_launch = defer_upload && !t->prm.status_every_iterate;
  if (one_launch) a->rel_src = 0;
  // iterate(true) with a local solve: the report is offered to the call's last launch (solve.hip takes it when that is the
  // closing statistics evaluation of a fused RGD step; anything else -- RTR, a restart iteration, the line search -- leaves it,
  // and k_report follows as a launch of its own)
  const bool offer = opt && will_report && !t->prm.status_every_iterate && t->use_report_tail;
  std::chrono::steady_clock::time_point tq0{};
  if (offer) {
    if (report_prepare(t, a)) return DPGO_ERR;
    t->rep_offer = dpgo_team::ReportOffer();
    t->rep_offer.valid = true;
    dpgo::ReportTail &rt = t->rep_offer.rt;
    rt.out = a->h_down.p; rt.frames = a->d_pub_all.p; rt.count = a->n_pub_all;
    rt.seq = a->d_report_seq.p; rt.ticket = a->d_report_seq.p + 1; rt.expect = a->report_seq + 1;
    tq0 = std::chrono::steady_clock::now();
  }
  const int rc = one_launch ? 0 : enqueue_iterate(t, a->local, opt ? 1 : (do_optimization ? 2 : 0), will_report);
  const bool folded = offer && t->rep_offer.taken;
  t->rep_offer = dpgo_team::ReportOffer();
  if (rc) return rc;
  if (do_optimization) mark_optimized(t, *a, opt ? (a->rel_src == 1 ? 1 : 5) : 2, opt);
  a->iter++;
  if (t->prm.acceleration || opt) a->publish_requested = true;
  if (folded) {
    // (what report_after_iterate would have asked k_report for: the status of a fused step -- PART_B partials -- and the
    // result of an RGD solve; an RTR solve leaves its record in pinned memory itself)
    if (a->opt_rel_src != 1) { set_err("internal: folded report on a path without the fused status partials"); return DPGO_ERR; }
    a->rep.pending = true; a->rep.expect = ++a->report_seq; a->rep.epoch = t->epoch; a->rep.one_seq = false;
    a->rep.want_status = true; a->rep.want_opt = a->opt_pending_rgd; a->rep.t_launch = tq0;
    t->counters[10] += 1;
    const int rr = finish_report(t, a);
    if (rr) return rr;
  } else if (will_report) {
    // (an accelerated iterate(false) leaves X = Y at every pose: one sequence crosses the bus)
    // (what updateNeighborPoses staged stays on the host across iterate(false) calls -- nothing reads the slabs until
    // this agent's next iterate(true), whose first launch scatters the latest value of every slot)
    // iterate(false) only ENQUEUES its report (see report_after_iterate); iterate(true) waits: the wrapper reads
    // mLocalOptResult and the status right behind it (:160-172)
    const int rr = report_after_iterate(t, a, opt, true, defer_upload && !one_launch, !do_optimization && t->prm.acceleration, one_launch,
                                        do_optimization != 0);
    if (rr) return rr;
  }
  return a->last_success ? DPGO_OK : DPGO_NOT_READY;
}

int dpgo_agent_get_status(dpgo_team_t *t, int id, dpgo_status_t *s) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  s->agent_id = a->id; s->state = a->state; s->instance_number = a->instance; s->iteration_number = a->iter;
  s->relative_change = 0; s->ready_to_terminate = 0;
  const bool robust = t->prm.robust_cost_type != DPGO_COST_L2;
  if (!t->prm.status_every_iterate) {
    // the status block describes the agent's last iterate(true)
    if (!a->has_X || a->opt_rel_src < 0) return DPGO_OK;
    if (!a->opt_cached) {
      double sum = 0;
      if (a->opt_rel_src != 2) {
        const bool tiles = a->opt_rel_src == 5;
        const int cnt = tiles ? (a->n + 63) / 64 : precond_nblk(*a);
        const int off = tiles ? PART_E : PART_B + 2;
        std::vector<double> part((size_t)cnt * PART_STRIDE);
        HIPC(hipMemcpyAsync(part.data(), a->dev.part + off, sizeof(double) * ((size_t)(cnt - 1) * PART_STRIDE + 1),
                            hipMemcpyDeviceToHost, t->stream));
        HIPC(hipStreamSynchronize(t->stream));
        if (check_exchange_error(t)) return DPGO_ERR;
        for (int k = 0; k < cnt; ++k) sum += part[(size_t)k * PART_STRIDE];
      }
      a->opt_rel_change = std::sqrt(sum / a->n);
      a->opt_cached = true;
    }
    s->relative_change = a->opt_rel_change;
    s->ready_to_terminate = a->opt_success && (s->relative_change <= t->prm.rel_change_tol) &&
                            (!robust || a->opt_ratio >= t->prm.robust_opt_min_convergence_ratio);
    return DPGO_OK;
  }
  if (!a->has_X || a->iter == 0 || a->rel_src == 2) { s->ready_to_terminate = a->has_X && a->iter > 0 && a->last_success; return DPGO_OK; }
  // |X - XPrev|^2 partials were left by the last kernel that moved X (fixed summation order)
  const int cnt = a->rel_src == 4 ? a->n : (a->rel_src ? precond_nblk(*a) : (a->n + 63) / 64);
  const int stride = a->rel_src == 4 ? 1 : PART_STRIDE;
  const int off = a->rel_src == 1 ? PART_B + 2 : PART_D;
  std::vector<double> part((size_t)cnt * stride);
  HIPC(hipMemcpyAsync(part.data(), a->dev.part + off, sizeof(double) * ((size_t)(cnt - 1) * stride + 1),
                      hipMemcpyDeviceToHost, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  double sum = 0;
  for (int k = 0; k < cnt; ++k) sum += part[(size_t)k * stride];
  s->relative_change = std::sqrt(sum / a->n);
  s->ready_to_terminate = a->last_success && (s->relative_change <= t->prm.rel_change_tol) &&
                          (!robust || converged_ratio(*a) >= t->prm.robust_opt_min_convergence_ratio);
  return DPGO_OK;
}

int dpgo_agent_get_opt_result(dpgo_team_t *t, int id, dpgo_opt_result_t *r) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (refresh_rtr_result(t, *a) || refresh_rgd_result(t, *a)) return DPGO_ERR;
  *r = a->opt;
  return DPGO_OK;
}

int dpgo_agent_iteration_number(dpgo_team_t *t, int id) {
  Agent *a = find_agent(t, id);
  return a ? a->iter : DPGO_ERR;
}

int dpgo_agent_preconditioner(dpgo_team_t *t, int id) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  return a->precond;
}

int dpgo_agent_preconditioner_info(dpgo_team_t *t, int id, double *out) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  const double N4 = 4.0 * a->n;
  const bool tl = a->precond == DPGO_PRECOND_TWO_LEVEL;
  out[0] = a->precond;
  out[1] = tl ? (double)a->tl_plan.sub.size() : 0;
  out[2] = tl ? a->tl_plan.ns : 0;
  out[3] = tl ? a->tl_plan.nwg : (4 * a->n + 7) / 8;
  out[4] = tl ? a->tl_plan.nA : 0;
  out[5] = tl ? a->tl_plan.bytes : (a->precond == DPGO_PRECOND_DENSE ? 8.0 * N4 * N4 : 128.0 * a->n);
  out[6] = 8.0 * N4 * N4;
  size_t mx = 0;
  if (tl) for (const auto &s : a->tl_plan.sub) mx = std::max(mx, s.size());
  out[7] = (double)mx;
  return DPGO_OK;
}

int dpgo_two_level_plan(int n, const int *rowptr, const int *col, int max_sub, int *sub_of, double *info) {
  if (n < 1 || !rowptr || !col) { set_err("two_level_plan: n >= 1, rowptr and col required"); return DPGO_ERR; }
  if (rowptr[0] != 0) { set_err("two_level_plan: rowptr[0] must be 0"); return DPGO_ERR; }
  for (int i = 0; i < n; ++i)
    if (rowptr[i + 1] < rowptr[i]) { set_err("two_level_plan: rowptr must be non-decreasing"); return DPGO_ERR; }
  for (int p = 0; p < rowptr[n]; ++p)
    if (col[p] < 0 || col[p] >= n) { set_err("two_level_plan: column index out of range"); return DPGO_ERR; }
  const std::vector<int> rp(rowptr, rowptr + n + 1), cl(col, col + rowptr[n]);
  const dpgo_host::TLPlan pl = dpgo_host::tl_make_plan(n, rp, cl, max_sub);
  if (sub_of) for (int i = 0; i < n; ++i) sub_of[i] = pl.sub_of[i];
  if (info) {
    info[0] = (double)pl.sub.size(); info[1] = pl.ns; info[2] = pl.nwg; info[3] = pl.nA; info[4] = pl.bytes;
    info[5] = dpgo_host::tl_worthwhile(pl) ? 1.0 : 0.0;
  }
  return DPGO_OK;
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
  const int ppb = 64 / t->prm.r, nb = (a->n + ppb - 1) / ppb;
  std::vector<double> part((size_t)PART_STRIDE * nb);
  HIPC(hipMemcpyAsync(part.data(), a->dev.part + PART_A, sizeof(double) * part.size(), hipMemcpyDeviceToHost, t->stream));
  if (egrad) HIPC(hipMemcpyAsync(egrad, a->dev.buf[B_T1], bytes, hipMemcpyDeviceToHost, t->stream));
  if (rgrad) HIPC(hipMemcpyAsync(rgrad, a->dev.buf[B_T2], bytes, hipMemcpyDeviceToHost, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
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

// how well the operator the kernels apply inverts Q + shift I:  |z (Q + shift I) - v| / |v|  for a fixed pseudo-random v,
// z from the device (the plain apply at X = 0, where the tangent projection is the identity), the product with the sparse
// matrix on the host.  What the reference's backward-stable Cholesky solve would leave is ~1e-16 x cond.
int dpgo_agent_preconditioner_residual(dpgo_team_t *t, int id, double *rel) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  const int r = t->prm.r, n = a->n;
  const size_t len = (size_t)r * 4 * n;
  std::vector<double> v(len), z(len), zero(len, 0.0);
  unsigned long long s = 0x9E3779B97F4A7C15ull;
  for (size_t i = 0; i < len; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    v[i] = (double)(s >> 11) / 9007199254740992.0 - 0.5;
  }
  if (dpgo_agent_precondition(t, id, zero.data(), v.data(), z.data())) return DPGO_ERR;
  double num = 0, den = 0;
  std::vector<double> w(4 * (size_t)r);
  for (int j = 0; j < n; ++j) {
    std::fill(w.begin(), w.end(), 0.0);
    for (int p = a->rowptr[j]; p < a->rowptr[j + 1]; ++p) {
      const int i = a->col[p];
      const double *val = a->qval.data() + (size_t)16 * p;
      for (int c = 0; c < 4; ++c)
        for (int cp = 0; cp < 4; ++cp) {
          const double q = val[cp + 4 * c] + ((i == j && cp == c) ? t->prm.precond_shift : 0.0);
          for (int b = 0; b < r; ++b) w[(size_t)c * r + b] += z[((size_t)4 * i + cp) * r + b] * q;
        }
    }
    for (int e = 0; e < 4 * r; ++e) {
      const double d = w[e] - v[(size_t)j * 4 * r + e];
      num += d * d;
      den += v[(size_t)j * 4 * r + e] * v[(size_t)j * 4 * r + e];
    }
  }
  *rel = std::sqrt(num / den);
  return DPGO_OK;
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

// GNC-TLS weights of one agent from the residuals of its current iterate (it owns a shared edge's weight when it is
// the lower-ID endpoint); marks its data matrices dirty but rebuilds nothing
// `pre`: the agent's residuals, already on the host (dpgo_team_update_weights fetches every agent's behind ONE wait)
static int update_weights_of(dpgo_team_t *t, Agent *a, const std::vector<double> *pre = nullptr) {
  std::vector<double> own;
  if (a->has_X && !pre && compute_residuals(t, *a, own)) return DPGO_ERR;
  const std::vector<double> &res = pre ? *pre : own;
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
  return DPGO_OK;
}

// after the rebuild that follows a weight update: the acceleration restarts from X (V = Y = X, gamma = alpha = 0)
static int reset_acceleration_of(dpgo_team_t *t, Agent *a) {
  if (!(t->prm.acceleration && a->has_X)) return DPGO_OK;
  launch_nest_reset(t->ctx(), a->local, a->n);
  NestState ns{}; ns.iter = a->iter;
  HIPC(hipMemcpyAsync(a->dev.nest, &ns, sizeof ns, hipMemcpyHostToDevice, t->stream));
  return DPGO_OK;
}

int dpgo_agent_update_measurement_weights(dpgo_team_t *t, int id) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  if (update_weights_of(t, a)) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  if (reset_acceleration_of(t, a)) return DPGO_ERR;
  HIPC(hipStreamSynchronize(t->stream));
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

// residuals of every stored measurement (order of dpgo_agent_get_measurements: odometry, private, shared) from ONE
// launch of the residual kernel; available[k] = 0 where a neighbour pose has not arrived yet
int dpgo_agent_compute_residuals(dpgo_team_t *t, int id, double *residuals, int *available) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (!a->has_X) return DPGO_NOT_READY;
  if (sync_descs(t)) return DPGO_ERR;
  std::vector<double> res;
  if (compute_residuals(t, *a, res)) return DPGO_ERR;
  int e = 0;
  for (auto *vec : {&a->odom, &a->priv, &a->shared})
    for (auto &m : *vec) {
      int ok = 1;
      if (m.r1 != a->id) { const int q = find_np(*a, m.r1, m.p1); if (q < 0 || !a->np_has[0][q]) ok = 0; }
      if (m.r2 != a->id) { const int q = find_np(*a, m.r2, m.p2); if (q < 0 || !a->np_has[0][q]) ok = 0; }
      if (residuals) residuals[e] = ok ? res[e] : 0.0;
      if (available) available[e] = ok;
      ++e;
    }
  return e;
}

// weights / fixed flags of every stored measurement at once (same order); marks the data matrices dirty
int dpgo_agent_set_measurement_weights(dpgo_team_t *t, int id, const double *weights, const int *fixed, int count) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (count != (int)(a->odom.size() + a->priv.size() + a->shared.size())) { set_err("set_measurement_weights: count mismatch"); return DPGO_ERR; }
  int e = 0;
  for (auto *vec : {&a->odom, &a->priv, &a->shared})
    for (auto &m : *vec) { m.weight = weights[e]; if (fixed) m.fixed_weight = fixed[e]; ++e; }
  a->data_dirty = true;
  return DPGO_OK;
}

// restartNesterovAcceleration: V = Y = X, gamma = alpha = 0 (what PGOAgent::updateMeasurementWeights does after the
// weights changed; the facade calls it from its own updateMeasurementWeights)
int dpgo_agent_reset_acceleration(dpgo_team_t *t, int id) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  if (reset_acceleration_of(t, a)) return DPGO_ERR;
  HIPC(hipStreamSynchronize(t->stream));
  return DPGO_OK;
}

// Robust local initialisation (InitializationMethod::GNC_TLS, src/PGOAgentROSNode.cpp:111-112): a single-robot
// GNC-TLS pose-graph solve on the device.  Odometry is trusted (fixed weight 1) and supplies the initial guess;
// the private loop closures are re-weighted between blocks of RTR iterations on the un-lifted problem (r = d = 3).
int dpgo_robust_local_init(int device, const dpgo_measurement_t *m, int nm, int num_poses, const dpgo_params_t *gnc,
                           double *T_out, double *weights_out) {
  if (!m || nm <= 0 || num_poses <= 0 || !gnc || !T_out) { set_err("robust_local_init: bad arguments"); return DPGO_ERR; }
  dpgo_params_t prm = *gnc;
  prm.d = 3; prm.r = 3; prm.num_robots = 1;
  prm.method = DPGO_METHOD_RTR;
  prm.acceleration = 0;
  prm.robust_cost_type = DPGO_COST_GNC_TLS;
  std::vector<dpgo_measurement_t> loc(m, m + nm);
  for (auto &e : loc) {
    if (e.r1 != e.r2) { set_err("robust_local_init: single-robot measurements expected"); return DPGO_ERR; }
    if (e.p1 < 0 || e.p1 >= num_poses || e.p2 < 0 || e.p2 >= num_poses) { set_err("robust_local_init: pose index out of range"); return DPGO_ERR; }
    e.r1 = e.r2 = 0;
    if (e.p1 + 1 == e.p2) { e.fixed_weight = 1; e.weight = 1.0; }
    else if (!e.fixed_weight) e.weight = 1.0;
  }
  const int id = 0;
  dpgo_team_t *t = dpgo_team_create(device, &prm, 1, &id, nullptr);
  if (!t) return DPGO_ERR;
  int rc = DPGO_ERR;
  do {
    if (dpgo_agent_add_measurements(t, id, loc.data(), nm)) break;
    std::vector<double> T0((size_t)12 * num_poses);
    dpgo_odometry_init(loc.data(), nm, num_poses, T0.data());
    if (dpgo_agent_num_poses(t, id) != num_poses) { set_err("robust_local_init: pose count mismatch"); break; }
    if (dpgo_agent_set_X(t, id, T0.data())) break;  // r = 3: the lifted iterate IS the 3 x 4n trajectory
    bool failed = false;
    for (int u = 0; u <= prm.robust_opt_num_weight_updates && !failed; ++u) {
      for (int k = 0; k < prm.robust_opt_inner_iters; ++k) {
        // a weighting that is already solved costs one gradient evaluation per call (RTR stops at its first test)
        if (dpgo_agent_iterate(t, id, 1) < 0) { failed = true; break; }
      }
      if (!failed && u < prm.robust_opt_num_weight_updates && dpgo_agent_update_measurement_weights(t, id)) failed = true;
    }
    if (failed) break;
    if (dpgo_agent_get_X(t, id, 0, T_out)) break;
    if (weights_out) {
      std::vector<dpgo_measurement_t> cur(nm);
      if (dpgo_agent_get_measurements(t, id, cur.data()) != nm) break;
      // stored as [odometry..., private...] in insertion order within each class
      int io = 0, ip = 0, nodo = 0;
      for (int k = 0; k < nm; ++k) nodo += (loc[k].p1 + 1 == loc[k].p2);
      for (int k = 0; k < nm; ++k) weights_out[k] = (loc[k].p1 + 1 == loc[k].p2) ? cur[io++].weight : cur[nodo + ip++].weight;
    }
    rc = DPGO_OK;
  } while (false);
  dpgo_team_destroy(t);
  return rc;
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
  t->graph_valid = false;  // (captured runs bake the schedule into their launches)
  return 0;
}

// PGOAgentROSParameters::UpdateRule::Uniform (include/dpgo_ros/PGOAgentROS.h:35-41, the struct's default :76; selected at
// src/PGOAgentROS.cpp:446-463): after every iteration the next token holder is drawn uniformly -- std::discrete_distribution
// with equal weights over the active, initialised robots, a std::mt19937 -- with replacement, so a robot may follow itself
// (:476 only warns).  The wrapper seeds from std::random_device; here the seed is given, and the draws are those the
// wrapper's own lines make from an engine of that seed.  The order is installed as the team's schedule (period `length`).
int dpgo_team_set_uniform_schedule(dpgo_team_t *t, unsigned seed, int length, int *order_out) {
  if (length <= 0) { set_err("set_uniform_schedule: length must be positive"); return DPGO_ERR; }
  std::vector<unsigned> active;
  for (auto &a : t->ag) active.push_back((unsigned)a->id);
  std::sort(active.begin(), active.end());
  if (active.empty()) { set_err("set_uniform_schedule: the team holds no robot"); return DPGO_ERR; }
  std::vector<double> weights(active.size(), 1.0);
  std::discrete_distribution<int> distribution(weights.begin(), weights.end());
  std::mt19937 gen(seed);
  std::vector<int> order((size_t)length);
  for (int k = 0; k < length; ++k) order[k] = (int)active[distribution(gen)];
  if (order_out) std::copy(order.begin(), order.end(), order_out);
  return dpgo_team_set_schedule(t, order.data(), length);
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
      if (t->id2local.count(a->np[q].first) && !t->isolated) { a->np_has[0][q] = 1; a->np_has[1][q] = 1; }
  }
  HIPC(hipStreamSynchronize(t->stream));
  return 0;
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
  account_iteration(t, sel, fused || t->last_iteration_folded);
  return 0;
}

// Carried rows of the one-launch iterations (step_fused.hip): launch `rep` of a run of nfe one-launch iterations finds the
// row products of its agent formed by launch rep - 1, from the evaluation point launch rep - 2 left -- which takes the
// agents of the three iterations to be three different ones (the point is formed while the agent rests) and both earlier
// launches to be part of the same run.  The first two launches of a run form their row products themselves.
static int fe_carry_flags(dpgo_team_t *t, int rep, int nfe, const std::function<int(int)> &sel_at) {
  if (!t->use_fe_carry) return 0;
  auto consumes = [&](int q) {
    if (q < 2 || q >= nfe) return false;
    const int a = sel_at(q - 2), b = sel_at(q - 1), c = sel_at(q);
    if (a < 0 || a == b || a == c || b == c) return false;
    // (the poses of agent c are spread over the workgroups of the launch of agent b)
    const int nblk = precond_nblk(*t->ag[b]);
    if ((t->ag[c]->n + nblk - 1) / nblk > step_fe_carry_max_poses()) return false;
    // (a gradient wave of the launch of agent c finishes 64 public poses and fetches their shared edges, two per lane)
    if (t->ag[c]->npub < 1 || t->ag[c]->npub > 256 || !t->ag[c]->dev.fe_code_ok) return false;  // (no public pose: no table to read)
    for (int g = 0; g < 4; ++g)
      if (t->ag[c]->dev.fe_eptr[g + 1] - t->ag[c]->dev.fe_eptr[g] > 128) return false;
    return true;
  };
  return (consumes(rep) ? FE_CARRY_IN : 0) | (consumes(rep + 1) ? FE_CARRY_W : 0) | (consumes(rep + 2) ? FE_CARRY_Y : 0);
}

// every one-launch iteration of a long run finds carried rows (every three consecutive agents of the schedule differ)
static bool fe_carry_everywhere(dpgo_team_t *t) {
  const int P = (int)t->sched.size();
  if (P < 3) return false;
  const std::function<int(int)> sel_at = [&](int rep) { return t->sched[(size_t)(rep % P)]; };
  for (int q = 2; q < P + 2; ++q)
    if (!(fe_carry_flags(t, q, 1 << 30, sel_at) & FE_CARRY_IN)) return false;
  return true;
}

// prepare_only: capture and instantiate every graph a run of `iters` iterations from the current state would replay
// (both alternating instances of each), execute nothing
// the one-launch iteration (step_fused.hip) may serve this team: dense agents of fe_min_n (449: where it is faster) .. 512 poses whose rows fit the ELL
// part, few enough public poses / shared edges for its LDS tables, the schedule and the descriptors baked into the
// launches (period <= 8), every neighbour co-resident (the twins of its poses are addressed through the shared-edge table)
static bool fused_eval_eligible(dpgo_team_t *t) {
  const dpgo_params_t &p = t->prm;
  const int P = (int)t->sched.size();
  if (!(t->use_fused_eval && t->bake_sel && t->bake_desc && P >= 1 && P <= 8 && step_fe_supported(p.r) && p.acceleration &&
        p.method == DPGO_METHOD_RGD && p.rgd_use_preconditioner && (int)t->ag.size() <= LOOKAHEAD_MAX_AGENTS &&
        t->h_descs.size() == t->ag.size() && t->precond_of.size() == t->ag.size() && t->peers.empty() && !t->isolated &&
        (int)t->ag.size() == p.num_robots))
    return false;
  // (with carried rows the one-launch form is the faster one at every size it was measured at, 41 .. 500 poses; without them
  // only from about 450 poses up -- profiles/experiments/fe_small.py)
  const int min_n = t->fe_min_n > 0 ? t->fe_min_n : (fe_carry_everywhere(t) ? 32 : 449);
  for (size_t k = 0; k < t->ag.size(); ++k) {
    const int n = t->ag[k]->n;
    if (t->precond_of[k] != DPGO_PRECOND_DENSE || n < min_n || n > 512 || !t->ag[k]->has_soa ||
        t->h_descs[k].nshared > step_fe_max_edges())
      return false;
  }
  return true;
}

// The deep-carried form (step_deep.hip) may serve this team: the private part of every agent's product is formed one launch
// early, its row products two, its evaluation point three -- so every FOUR consecutive agents of the schedule differ; the
// first 24 chunks of every agent's order are private (24 is returned; 0: not this team); an agent's
// public poses fit two waves, its shared edges three edge slots of 64, and the partial sums have their buffers.
static int fe_deep_m0(dpgo_team_t *t) {
  if (!t->use_fe_deep || !t->use_fe_carry || !fused_eval_eligible(t)) return 0;
  const int P = (int)t->sched.size();
  if (P < 4) return 0;
  for (int q = 0; q < P; ++q)
    for (int u = 1; u < 4; ++u)
      if (t->sched[(size_t)q] == t->sched[(size_t)((q + u) % P)]) return 0;
  int min_priv = 32, nblk_all = 0, total = 0;
  for (auto &a : t->ag) {
    if (a->npub < 1 || a->npub > 128 || !a->dev.fe_code_ok || (int)a->se_host.size() > FE_MAX_EDGES) return 0;
    min_priv = std::min(min_priv, a->dev.fe_npriv);
    nblk_all = std::max(nblk_all, (4 * a->n + 7) / 8);
    total += a->n;
  }
  for (auto &a : t->ag)
    if ((total - a->n + nblk_all - 1) / nblk_all > 64) return 0;
  int m0 = step_fd_pick_m0(min_priv);
  if (const char *e = std::getenv("DPGO_FD_M0")) { const int f = std::atoi(e); if (f > 0 && f <= min_priv && step_fd_pick_m0(f) == f) m0 = f; }  // (experiments)
  if (m0 == 0) return 0;
  const size_t want = (size_t)2 * nblk_all * t->prm.r * 256;
  if (t->d_fd_pacc.n < want && t->d_fd_pacc.alloc(want)) return 0;
  if (t->use_fe_persist && !t->d_pd_bar.p && t->d_pd_bar.alloc(PD_BAR_WORDS)) return 0;
  return m0;
}

// one run of nfe deep-carried one-launch iterations from the state k_nest_pre leaves: the points of the first three agents,
// two launches that only produce (the row products of sel(0); then its private partial sums and the row products of
// sel(1)), then the iterations -- each consuming what the three launches before it left
// how many iterations of a graph of B run as one launch each.  Round 5's form leaves the last L + 1 (L = schedule period:
// every agent's last block update of the run, whose statistics a status query reads) to the two-launch sequence; the
// deep-carried form leaves those statistics itself (FD_STATS / FD_LASTAT) and hands over only the last iteration, which does
// not look ahead.  An even number either way: the launches alternate between the two copies of the poses.
static int fe_run_length(bool fe, int fd_m0, int B, int L) {
  if (!fe) return 0;
  if (fd_m0 > 0 && ((B - 1) & ~1) >= 4) return (B - 1) & ~1;
  return std::max(0, B - L - 1) & ~1;
}
static bool fe_run_is_deep(bool fe, int fd_m0, int B) { return fe && fd_m0 > 0 && ((B - 1) & ~1) >= 4; }

static void enqueue_fe_deep(dpgo_team_t *t, const LaunchCtx &c, int m0, int nfe, int B, int L, int iter0, const std::function<int(int)> &sel_at,
                            NestState *nest_own, NestState *const nest_fe[2]) {
  const dpgo_params_t &p = t->prm;
  if (t->use_fe_persist && t->d_pd_bar.p) {
    // the same run as ONE persistent launch (step_persist.hip): the points of the first three agents, the counters zeroed,
    // then the two producing iterations and the nfe real ones behind grid hand-offs inside the kernel
    launch_fd_prime(c, sel_at(0), sel_at(1), sel_at(2), t->max_n, p.num_robots, p.restart_interval, nest_own);
    (void)hipMemsetAsync(t->d_pd_bar.p, 0, sizeof(unsigned long long) * PD_BAR_WORDS, c.stream);
    launch_step_pd(c, m0, t->d_sched.p, (int)t->sched.size(), iter0 % (int)t->sched.size(), nfe, B, L, p.rgd_stepsize, p.num_robots,
                   p.restart_interval, nest_own, nest_fe[nfe & 1], t->d_pd_bar.p, t->h_bar_err);
    return;
  }
  int nblk_all = 0;
  for (auto &a : t->ag) nblk_all = std::max(nblk_all, (4 * a->n + 7) / 8);
  double *pacc[2] = {t->d_fd_pacc.p, t->d_fd_pacc.p + (size_t)nblk_all * p.r * 256};
  const int s0 = sel_at(0), s1 = sel_at(1), s2 = sel_at(2);
  launch_fd_prime(c, s0, s1, s2, t->max_n, p.num_robots, p.restart_interval, nest_own);
  launch_fd_open(c, m0, s0, s1, pacc[0]);
  for (int rep = 0; rep < nfe; ++rep) {
    const int flags = FD_IN | (rep + 1 < nfe ? FD_P : 0) | (rep + 2 < nfe ? FD_W : 0) | (rep + 3 < nfe ? FD_Y : 0) |
                      (rep >= B - L ? FD_STATS : 0) | ((rep + 1 < B && rep + 1 >= B - L) ? FD_LASTAT : 0);
    launch_step_fd(c, m0, sel_at(rep), sel_at(rep + 1), sel_at(rep + 2), sel_at(rep + 3), p.rgd_stepsize, p.num_robots,
                   p.restart_interval, rep == 0 ? nest_own : nest_fe[rep & 1], nest_fe[(rep + 1) & 1], rep & 1, flags,
                   pacc[rep & 1], pacc[(rep + 1) & 1]);
  }
}

static int team_run_impl(dpgo_team_t *t, int iters, bool prepare_only) {
  if (sync_descs(t)) return DPGO_ERR;
  const dpgo_params_t &p = t->prm;
  for (auto &a : t->ag) if (!a->has_X) { set_err("team_run before set_initial"); return DPGO_NOT_READY; }
  const bool graphable = (p.method == DPGO_METHOD_RGD) && p.rgd_use_preconditioner;
  if (graphable && !t->graph_valid) {
    for (auto &kv : t->graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
    t->graphs.clear();
    t->graph_flip.clear();
    t->graph_valid = true;
  }
  // look-ahead Nesterov steps need every workgroup's share of the other agents' poses to fit one wave, and one
  // double per pose in the PART_D region
  // (a line search decides the step after the whole agent's trial costs are known: its iterations are the un-fused
  // launch sequence of enqueue_team_iteration, captured as it is)
  const bool ls = p.rgd_line_search != 0;
  bool pipelined = p.acceleration != 0 && (int)t->ag.size() <= LOOKAHEAD_MAX_AGENTS && !ls;
  {
    int total = 0;
    for (auto &a : t->ag) total += a->n;
    for (auto &a : t->ag) {
      const int nblk = precond_nblk(*a);
      if ((total - a->n + nblk - 1) / nblk > 64 || a->n > MAX_PART * PART_STRIDE) pipelined = false;
    }
  }
  // the schedule, the counters and the Nesterov scalars live on the device, so a run of B iterations is one
  // fixed launch sequence: captured once per B and replayed
  // lead: the window opens with a restart iteration (un-fused kernels, same launch sequence every time); B: fused
  // iterations that follow.  Pipelined teams have no windows: restart iterations are part of the uniform sequence
  // (lead is never set).  Two instances per key alternate, so that a launch never has to wait for the previous
  // replay of the same executable graph.
  // Pipelined teams with a short schedule period bake the agent of every iteration into its launches (one graph per
  // phase of the schedule): the kernels then address the agent's descriptor from a kernel argument instead of through
  // team->cur_sel / next_sel, one dependent round trip less in each prologue.
  const int P = (int)t->sched.size();
  const bool bake = t->bake_sel && P >= 1 && P <= 8;
  // One-launch iterations (step_fused.hip) for the mid-run part of a pipelined graph: dense agents of 449 .. 512 poses,
  // the schedule and the descriptors baked in.  Their launches alternate between the two copies of the poses (parity),
  // so they need neither each other's company on the device nor its lock
  const bool fe_ok = pipelined && graphable && fused_eval_eligible(t);
  const int fd_m0 = fe_ok ? fe_deep_m0(t) : 0;  // > 0: runs of one-launch iterations take the deep-carried form (step_deep.hip)
  if (t->use_fe_persist && fd_m0 > 0 && !prepare_only && !acquire_fused_rtr_lock(t)) {
    // the persistent form waits for its own workgroups: like the one-launch RTR solve it runs only under the device's lock
    // (given back wherever the stream is known to have drained); without it the team keeps the per-launch form for good
    t->use_fe_persist = 0;
    for (auto &kv : t->graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
    t->graphs.clear();
    t->graph_flip.clear();
  }
  auto graph_for = [&](bool lead, int B, int iter0, bool fe, hipGraphExec_t *out) -> int {
    const int phase = bake ? iter0 % P : -1;
    const int base = ((((lead ? 1 : 0) + 2 * B) * 16 + phase + 1) * 2 + (fe ? 1 : 0)) * 2;
    const int key = base + (t->graph_flip[base / 2] ^= 1);
    auto sel_at = [&](int rep) { return bake ? t->sched[(size_t)((iter0 + rep) % P)] : -1; };
    auto it = t->graphs.find(key);
    if (it != t->graphs.end()) { *out = it->second; return 0; }
    hipGraph_t g = nullptr;
    HIPC(hipStreamBeginCapture(t->stream, hipStreamCaptureModeThreadLocal));
    int rc = 0;
    if (lead) rc = enqueue_team_iteration(t, true, true, -1, 0);
    if (rc == 0 && B > 0 && p.acceleration && pipelined) {
      // pipelined: 2 launches per iteration (see k_eval_stats), restart iterations included.  The Nesterov step of
      // the first iteration is a launch of its own, the last iteration does not look ahead, and its statistics /
      // bookkeeping close the run.
      LaunchCtx c = t->ctx();
      c.bake_desc = bake && t->bake_desc;  // the agent's descriptor by value in the launches that name their agent
      const int na = (int)t->ag.size(), mn = t->max_n;
      launch_nest_pre(c, -1, -1, na, mn, p.num_robots, p.restart_interval, 1);
      // the last L = min(B, schedule period) steps leave their statistics (X2 snapshot, |X - XPrev|^2: every agent's
      // last block update of the run lies among them, and a status query reads it, a9), the look-aheads in front of
      // them leave XPrev and |Y' - X|^2; nothing reads these values earlier in the run
      const int L = std::min(B, (int)t->sched.size());
      // iterations [0, nfe): one launch each (they are the ones that leave nothing behind: ahead == 3); an even number,
      // so that the poses end in the primary arrays
      const int nfe = fe_run_length(fe, fd_m0, B, L);
      NestState *nest_own = t->d_nest_all.p, *nest_fe[2] = {t->d_nest_all.p + na, t->d_nest_all.p + 2 * na};
      const bool deep = fe_run_is_deep(fe, fd_m0, B);
      if (deep) enqueue_fe_deep(t, c, fd_m0, nfe, B, L, iter0, sel_at, nest_own, nest_fe);
      for (int rep = deep ? nfe : 0; rep < B; ++rep) {
        const int ahead = (rep + 1 < B ? 3 : 0) | ((rep + 1 < B && rep + 1 >= B - L) ? 4 : 0) | (rep >= B - L ? 8 : 0);
        if (rep < nfe) {
          launch_step_fe(c, sel_at(rep), sel_at(rep + 1), p.rgd_stepsize, p.num_robots, p.restart_interval,
                         rep == 0 ? nest_own : nest_fe[rep & 1], nest_fe[(rep + 1) & 1], rep & 1, sel_at(rep + 2),
                         fe_carry_flags(t, rep, nfe, sel_at));
          continue;
        }
        launch_eval_stats(c, mn, rep == 0, 1, 0, p.num_robots, p.restart_interval, sel_at(rep), -1,
                          (rep == nfe && nfe > 0) ? nest_fe[nfe & 1] : nullptr);
        launch_precond(c, sel_at(rep), mn, PM_RGD_, B_X, B_GF, B_Z, 0, 0, p.rgd_stepsize, 1, p.num_robots, 2, p.restart_interval,
                       ahead);
      }
      launch_eval_stats(c, mn, 0, 0, 1, p.num_robots, p.restart_interval, -1, sel_at(B - 1));
    } else if (rc == 0 && B > 0 && p.acceleration && !ls) {
      // 3 launches per iteration: [statistics of iteration k-1 + Nesterov step of iteration k] in one
      // heterogeneous kernel, cost/gradient (+ G from the neighbours' Y), preconditioner + RGD step +
      // Nesterov V + bookkeeping
      LaunchCtx c = t->ctx();
      const int na = (int)t->ag.size(), mn = t->max_n;
      for (int rep = 0; rep < B; ++rep) {
        if (rep == 0) launch_nest_pre(c, -1, -1, na, mn, p.num_robots, p.restart_interval);
        else launch_stats_nest(c, na, mn, p.num_robots, p.restart_interval);
        launch_eval(c, -1, mn, B_X, B_EGRAD, B_GF, PART_C, eval_opts(t, 2, 1, 0));
        launch_precond(c, -1, mn, PM_RGD_, B_X, B_GF, B_Z, 0, 0, p.rgd_stepsize, 1, p.num_robots, 1, p.restart_interval);
      }
      launch_eval(c, -5, mn, B_X2, B_EGRAD2, B_GF2, PART_A, eval_opts(t, 0, 0, 0));
    } else if (rc == 0) {
      for (int rep = 0; rep < B && !rc; ++rep) rc = enqueue_team_iteration(t, true, false, -1, 0, rep + 1 < B);
    }
    HIPC(hipStreamEndCapture(t->stream, &g));
    if (rc) { (void)hipGraphDestroy(g); return rc; }
    hipGraphExec_t ge = nullptr;
    HIPC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    (void)hipGraphUpload(ge, t->stream);  // (the first replay of a prepared graph then costs what the later ones do)
    t->graphs[key] = ge;
    *out = ge;
    return 0;
  };
  // evals: sparse evaluations of the iteration -- the gradient and the closing statistics of the two-launch form, the
  // gradient alone in a one-launch iteration (it leaves no statistics); a line-search iteration adds its trial passes
  auto account = [&](int sel, int evals) {
    t->counters[0] += 1; t->counters[1] += precond_operator_bytes(*t->ag[sel]);
    t->counters[2] += evals; t->counters[3] += evals * spmm_bytes_of(t, *t->ag[sel]);
  };
  int k = 0;
  int cur_iter = t->iter;
  if (prepare_only && !graphable) return 0;
  while (k < iters) {
    const bool uniform = graphable && p.acceleration && pipelined;  // restart iterations are ordinary iterations of the sequence
    const bool restart = !uniform && p.acceleration && ((cur_iter + 2) % p.restart_interval) == 0;
    int batch = 1;
    if (graphable) {
      // one graph per window: [the restart iteration, if the window opens with one] + the fused iterations up to
      // the next restart iteration
      int fusedn = iters - k - (restart ? 1 : 0);
      if (p.acceleration && !uniform) {
        const int it0 = cur_iter + (restart ? 1 : 0);
        const int to_restart = (p.restart_interval - ((it0 + 2) % p.restart_interval)) % p.restart_interval;
        fusedn = std::min(fusedn, to_restart);
      }
      // (the uniform pipelined sequence pays two extra launches per graph -- the first Nesterov step, the closing
      // statistics -- and six two-launch iterations at its end: longer graphs)
      fusedn = std::max(0, std::min(fusedn, uniform ? dpgo_team::MAX_PIPELINED_GRAPH_ITERS : dpgo_team::MAX_GRAPH_ITERS));
      batch = fusedn + (restart ? 1 : 0);
      hipGraphExec_t ge = nullptr;
      // (prepared graphs are the ones a run will ask for: with the one-launch iterations if this team may take the lock)
      const bool fe = fe_ok && (fd_m0 > 0 ? fusedn >= 6 : fusedn > (int)t->sched.size() + 2);
      const int grc = graph_for(restart, fusedn, cur_iter, fe, &ge);
      if (grc) return grc;
      if (prepare_only) {
        const int grc2 = graph_for(restart, fusedn, cur_iter, fe, &ge);  // the other instance; leaves the alternation where it was
        if (grc2) return grc2;
        cur_iter += batch;
        k += batch;
        continue;
      }
      ++t->epoch;
      HIPC(hipGraphLaunch(ge, t->stream));
      const int nfe_run = fe_run_length(fe, fd_m0, fusedn, std::min(fusedn, (int)t->sched.size()));
      t->counters[7] += nfe_run;  // one-launch iterations
      {
        const int P_ = (int)t->sched.size(), it0 = t->iter;
        const std::function<int(int)> sel_run = [&](int rep) { return t->sched[(size_t)((it0 + rep) % P_)]; };
        if (fe_run_is_deep(fe, fd_m0, fusedn)) { t->counters[8] += nfe_run; t->counters[9] += nfe_run; }  // ... deep-carried: every one of them
        else
        for (int q = 0; q < nfe_run; ++q) t->counters[8] += (fe_carry_flags(t, q, nfe_run, sel_run) & FE_CARRY_IN) ? 1 : 0;  // ... with carried rows
      }
      // after >= 2 pipelined iterations every agent took its last Nesterov step as a look-ahead (per-pose partials)
      for (auto &a : t->ag) a->rel_src = p.acceleration ? ((pipelined && fusedn >= 2) ? 4 : 0) : 2;
      for (int q = 0; q < batch; ++q) {
        const int sel = t->sched[(t->iter + q) % t->sched.size()];
        const int evals = ls ? 2 + (ls_trials(p) + 3) / 4 : ((q - (restart ? 1 : 0) < nfe_run && !(restart && q == 0)) ? 1 : 2);
        account(sel, evals);
        if (restart && q == 0) account(sel, evals);  // the restart iteration solves twice (from Y, then from XPrev)
        // status of this block update: the fused step leaves PART_B[2], the un-fused restart iteration k_status tiles
        mark_optimized(t, *t->ag[sel], ((restart && q == 0) || ls) ? 5 : 1, true);
        if (q == batch - 1) {
          t->ag[sel]->opt_pending_rgd = true;
          t->ag[sel]->rel_src = (fusedn > 0 && !ls) ? 1 : 0;  // a lone restart iteration ends with k_status (PART_D tiles)
        }
      }
    } else {
      const int sel = t->sched[t->iter % t->sched.size()];
      for (auto &a : t->ag) a->rel_src = p.acceleration ? 0 : 2;
      const int rc = enqueue_team_iteration(t, false, restart, sel, 0);
      if (rc) return rc;
      // status of the block update: k_status tiles, or -- where the one-launch RTR solve took the iteration's tail --
      // one partial per pose pair in PART_B[2], the fused RGD step's layout
      t->ag[sel]->rel_src = t->last_iteration_folded ? 1 : 0;
      mark_optimized(t, *t->ag[sel], t->last_iteration_folded ? 1 : 5, true);
    }
    t->iter += batch;
    cur_iter = t->iter;
    for (auto &a : t->ag) { a->iter += batch; if (p.robust_cost_type != DPGO_COST_L2) a->robust_inner_iter += batch; }
    t->counters[4] += batch;
    k += batch;
  }
  return 0;
}

int dpgo_team_run(dpgo_team_t *t, int iters) { return team_run_impl(t, iters, false); }
int dpgo_team_prepare(dpgo_team_t *t, int iters) { return team_run_impl(t, iters, true); }

int dpgo_team_get_coloring(dpgo_team_t *t, int *color_of_agent) {
  if (sync_descs(t)) return DPGO_ERR;
  for (size_t k = 0; k < t->ag.size(); ++k) color_of_agent[k] = t->color_of[k];
  return (int)t->groups.size();
}

// explicit colour classes (global robot ids; ids that are not local are ignored) for teams that hold only part
// of the problem: the colouring must be global so that no two agents updated together share an edge
int dpgo_team_set_groups(dpgo_team_t *t, int num_groups, const int *group_ptr, const int *member_ids) {
  t->groups.assign(num_groups, {});
  t->group_ids.assign(num_groups, {});
  t->color_of.assign(t->ag.size(), -1);
  for (int g = 0; g < num_groups; ++g)
    for (int q = group_ptr[g]; q < group_ptr[g + 1]; ++q) {
      t->group_ids[g].push_back(member_ids[q]);  // (global ids, non-local members included: rank_exchange.cpp sends to them)
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
  if (!t->groups[g].empty()) {
    LaunchCtx cg = c;
    cg.ny = (int)t->groups[g].size();
    int gmn = 0;
    for (int k : t->groups[g]) gmn = std::max(gmn, t->ag[k]->n);
    launch_status(cg, SEL_GROUP0 - g, -1, cg.ny, gmn, 1);
  }
  launch_advance(c, -1, na, 0, p.num_robots, p.restart_interval, 1, count);
  for (auto &a : t->ag) { a->rel_src = 0; a->iter += count; if (p.robust_cost_type != DPGO_COST_L2) a->robust_inner_iter += count; }
  for (int k : t->groups[g]) { t->ag[k]->publish_requested = true; mark_optimized(t, *t->ag[k], 5, true); }
  t->iter += count;
  t->counters[4] += count;
  return 0;
}

// Simultaneous updates: every local agent takes one preconditioned RGD step per tick, all in the same launches
// (blockIdx.y = agent), each from the neighbour poses as they were when the tick began.  This is the deterministic
// instance of the asynchronous (ASAPP) mode in which all Poisson clocks fire together (src/PGOAgentROS.cpp:119-127
// runs the same RGD step from whatever neighbour poses have arrived); one graph replay per call.
int dpgo_team_run_simultaneous(dpgo_team_t *t, int ticks) {
  if (sync_descs(t)) return DPGO_ERR;
  const dpgo_params_t &p = t->prm;
  if (p.method != DPGO_METHOD_RGD || !p.rgd_use_preconditioner || p.acceleration || p.rgd_line_search) {
    set_err("simultaneous updates: preconditioned RGD with the fixed step, without acceleration (the ASAPP configuration)");
    return DPGO_ERR;
  }
  for (auto &a : t->ag) if (!a->has_X) { set_err("run_simultaneous before set_initial"); return DPGO_NOT_READY; }
  const int na = (int)t->ag.size();
  if (na == 0 || ticks <= 0) return 0;
  if (!t->graph_valid) {
    for (auto &kv : t->graphs) if (kv.second) (void)hipGraphExecDestroy(kv.second);
    t->graphs.clear();
    t->graph_flip.clear();
    t->graph_valid = true;
  }
  LaunchCtx c = t->ctx();
  c.ny = na;
  const int sel = SEL_ALL, mn = t->max_n;  // (= the class t->all_group: the local agents in index order)
  auto body = [&](int reps) {
    for (int rep = 0; rep < reps; ++rep) {
      // XPrev only feeds the status of the LAST tick of a run (|X - XPrev|^2 left by its step kernel): the copy is
      // taken in the last tick of every graph, two launches per tick otherwise
      if (rep == reps - 1) launch_copy(c, -3, -1, na, mn, B_X, B_XPREV, 0);
      launch_eval(c, sel, mn, B_X, B_EGRAD, B_GF, PART_C, eval_opts(t, 2, 0, 0));
      // (statistics -- X2 snapshot, |X - XPrev|^2 -- only from the last tick of a graph: nothing reads the others')
      launch_precond(c, sel, mn, PM_RGD_, B_X, B_GF, B_Z, 0, 0, p.rgd_stepsize, 0, p.num_robots, 0, p.restart_interval,
                     rep == reps - 1 ? 0 : 16);
    }
  };
  int left = ticks;
  while (left > 0) {
    const int B = std::min(left, dpgo_team::MAX_GRAPH_ITERS);
    const int key = -(B + 1);  // negative keys: simultaneous-update graphs
    hipGraphExec_t ge = nullptr;
    auto it = t->graphs.find(key);
    if (it != t->graphs.end()) ge = it->second;
    else {
      hipGraph_t g = nullptr;
      HIPC(hipStreamBeginCapture(t->stream, hipStreamCaptureModeThreadLocal));
      body(B);
      HIPC(hipStreamEndCapture(t->stream, &g));
      HIPC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      (void)hipGraphDestroy(g);
      t->graphs[key] = ge;
    }
    ++t->epoch;
    HIPC(hipGraphLaunch(ge, t->stream));
    left -= B;
  }
  // f_opt / gradnorm_opt of every agent on the snapshot of its last step, then the counters
  launch_eval(c, sel, mn, B_X2, B_EGRAD2, B_GF2, PART_A, eval_opts(t, 0, 0, 0));
  LaunchCtx c1 = t->ctx();
  launch_advance(c1, -1, na, 0, p.num_robots, p.restart_interval, 1, ticks, ticks * na);
  for (auto &a : t->ag) {
    t->counters[0] += ticks; t->counters[1] += ticks * precond_operator_bytes(*a);
    t->counters[2] += ticks + 1; t->counters[3] += (ticks + 1) * spmm_bytes_of(t, *a);
    a->rel_src = 1; a->iter += ticks; a->opt_pending_rgd = true; a->publish_requested = true;
    mark_optimized(t, *a, 1, true);
    if (p.robust_cost_type != DPGO_COST_L2) a->robust_inner_iter += ticks;
  }
  t->iter += ticks * na;
  t->counters[4] += ticks * na;
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
      if (gs > 0) {
        LaunchCtx cg = c;
        cg.ny = gs;
        int gmn = 0;
        for (int k : t->groups[g]) gmn = std::max(gmn, t->ag[k]->n);
        launch_status(cg, SEL_GROUP0 - (int)g, -1, gs, gmn, 1);
      }
      launch_advance(c, -1, na, 0, p.num_robots, p.restart_interval, 1, gs);
      for (auto &a : t->ag) { a->rel_src = 0; a->iter += gs; if (p.robust_cost_type != DPGO_COST_L2) a->robust_inner_iter += gs; }
      for (int k : t->groups[g]) mark_optimized(t, *t->ag[k], 5, true);
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
  // every agent's scalars into its own 16 doubles of the pinned buffer, ONE wait for all of them (a wait per agent was
  // most of this call on an 8-agent team)
  for (auto &a : t->ag)
    HIPC(hipMemcpyAsync(t->h_scal + 16 * (size_t)a->local, a->dev.scal, sizeof(double) * 16, hipMemcpyDeviceToHost, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  release_fused_rtr_lock(t);
  if (check_exchange_error(t)) return DPGO_ERR;
  for (auto &a : t->ag) total += t->h_scal[16 * (size_t)a->local + 5];  // (agent order: the sum is what it was)
  *f = total;
  return 0;
}

int dpgo_team_update_weights(dpgo_team_t *t) {
  static const bool timing = std::getenv("DPGO_TIMING") != nullptr;  // stage times of a round on stderr (profiles/experiments)
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  const auto q0 = now();
  if (sync_descs(t)) return DPGO_ERR;
  if (dpgo_team_exchange_all(t)) return DPGO_ERR;
  const auto q1 = now();
  int changed = 0;
  // all weights first (every agent's residuals come from the current iterate), then ONE rebuild of the data
  // matrices and preconditioners of the whole team (batched dense inversions)
  {
    // every agent's residuals come from the current iterate: all the launches and copies first, ONE wait (a wait per agent
    // was 0.25 ms of a round on 8 agents)
    // (into ONE pinned buffer: a copy into pageable memory stages, and the host waited 18 us per agent between a kernel and
    // the next, profiles/r06_update_weight_timeline.txt)
    std::vector<std::vector<double>> res(t->ag.size());
    std::vector<size_t> at(t->ag.size(), 0);
    size_t total = 0;
    for (size_t k = 0; k < t->ag.size(); ++k) { at[k] = total; if (t->ag[k]->has_X) total += (size_t)t->ag[k]->nedges; }
    if (t->h_resid.alloc(std::max<size_t>(total, 1))) { set_err("pinned allocation failed"); return DPGO_ERR; }
    LaunchCtx c = t->ctx();
    for (size_t k = 0; k < t->ag.size(); ++k) {
      Agent &a = *t->ag[k];
      if (!a.has_X) continue;
      launch_residuals(c, a.local, a.nedges);
      if (a.nedges) HIPC(hipMemcpyAsync(t->h_resid.p + at[k], a.dev.resid, sizeof(double) * a.nedges, hipMemcpyDeviceToHost, t->stream));
    }
    HIPC(hipStreamSynchronize(t->stream));
    for (size_t k = 0; k < t->ag.size(); ++k)
      if (t->ag[k]->has_X) res[k].assign(t->h_resid.p + at[k], t->h_resid.p + at[k] + t->ag[k]->nedges);
    for (size_t k = 0; k < t->ag.size(); ++k) if (update_weights_of(t, t->ag[k].get(), &res[k])) return DPGO_ERR;
  }
  const auto q2 = now();
  // the owner of a shared edge (the robot with the smaller id) hands its weight to the other end point's copy: one index
  // over every agent's shared edges (a scan of the receiver's measurements per edge was 0.9 ms of a round) -- built when the
  // edge lists change and kept as (owner's copy, receiver's copy) pairs: rebuilding it was 0.23 ms of every round.
  // Parallel edges between the same two poses: the FIRST stored copy receives every one of them, as the scan did (and
  // the oracle does).
  {
    std::vector<std::pair<const void *, size_t>> key;
    for (auto &a : t->ag) key.emplace_back((const void *)a->shared.data(), a->shared.size());
    if (key != t->shared_links_key) {
      typedef std::array<int, 4> EdgeKey;
      struct EdgeHash {
        size_t operator()(const EdgeKey &k) const {
          unsigned long long h = (unsigned long long)(unsigned)k[0] * 0x9E3779B97F4A7C15ull;
          h = (h ^ (unsigned)k[1]) * 0xC2B2AE3D27D4EB4Full;
          h = (h ^ (unsigned)k[2]) * 0x165667B19E3779F9ull;
          return (size_t)((h ^ (unsigned)k[3]) * 0x9E3779B97F4A7C15ull);
        }
      };
      std::vector<std::unordered_map<EdgeKey, dpgo_measurement_t *, EdgeHash>> index(t->ag.size());
      for (size_t k = 0; k < t->ag.size(); ++k) {
        index[k].reserve(2 * t->ag[k]->shared.size());
        for (auto &m : t->ag[k]->shared) index[k].emplace(EdgeKey{m.r1, m.p1, m.r2, m.p2}, &m);  // (keeps the first)
      }
      t->shared_links.clear();
      for (auto &a : t->ag)
        for (auto &m : a->shared) {
          const int other = (m.r1 == a->id) ? m.r2 : m.r1;
          auto ol = t->id2local.find(other);
          if (other < a->id || ol == t->id2local.end()) continue;
          auto it = index[ol->second].find(EdgeKey{m.r1, m.p1, m.r2, m.p2});
          t->shared_links.push_back({&m, it != index[ol->second].end() ? it->second : nullptr, ol->second});
        }
      t->shared_links_key = key;
    }
    for (const auto &l : t->shared_links) {
      double w = l.from->weight;
      if (t->prm.weights_as_float32) w = (double)(float)w;
      if (l.to) { l.to->weight = w; l.to->fixed_weight = l.from->fixed_weight; ++changed; }
      t->ag[l.to_local]->data_dirty = true;
    }
  }
  const auto q3 = now();
  if (sync_descs(t)) return DPGO_ERR;
  const auto q4 = now();
  for (auto &a : t->ag) if (reset_acceleration_of(t, a.get())) return DPGO_ERR;
  HIPC(hipStreamSynchronize(t->stream));
  if (timing)
    std::fprintf(stderr, "update_weights: exchange %.2f  residuals+weights %.2f  shared weights %.2f  rebuild (enqueue) %.2f  "
                         "reset + drain %.2f ms\n", ms(q0, q1), ms(q1, q2), ms(q2, q3), ms(q3, q4), ms(q4, now()));
  return changed;
}

// PGOAgent::shouldTerminate() from the leader's point of view [UPSTREAM-RECALL for the body; in tree: only the leader
// evaluates it, right after its own iterate(true), src/PGOAgentROS.cpp:206-214; max_num_iters rule Node.cpp:228-232]
int dpgo_team_should_terminate(dpgo_team_t *t) {
  Agent *lead = find_agent(t, 0);
  if (!lead) { set_err("should_terminate: robot 0 (the leader) does not live in this team"); return DPGO_ERR; }
  if ((int)t->ag.size() != t->prm.num_robots) { set_err("should_terminate: the team must hold every robot"); return DPGO_ERR; }
  if (lead->iter > t->prm.max_num_iters) return 1;
  if (t->prm.robust_cost_type != DPGO_COST_L2 && lead->weight_update_count < t->prm.robust_opt_num_weight_updates) return 0;
  for (auto &a : t->ag) {
    dpgo_status_t s;
    if (dpgo_agent_get_status(t, a->id, &s) != DPGO_OK) return DPGO_ERR;
    if (s.state != DPGO_INITIALIZED || !s.ready_to_terminate) return 0;
  }
  return 1;
}

// ---- per-iteration log.  The reference's wrapper writes one CSV per robot -- a header (createIterationLog,
// src/PGOAgentROS.cpp:853-867), one row after every iterate(true) of the robot (logIteration, :869-894, called at :189),
// and the strings UPDATE_WEIGHT (:1217) / TERMINATE (:1042) when those commands arrive.  Same columns in the same order here,
// followed by the global cost the reference lacks.
static void close_iteration_log(dpgo_team_t *t) {
  for (FILE *f : t->ilog.f) if (f) std::fclose(f);
  t->ilog.f.clear();
  t->ilog.bytes_received.clear();
}

int dpgo_team_set_iteration_log(dpgo_team_t *t, const char *directory) {
  close_iteration_log(t);
  if (!directory) return DPGO_OK;
  for (auto &a : t->ag) {
    const std::string path = std::string(directory) + "/dpgo_log_robot" + std::to_string(a->id) + ".csv";
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f) { close_iteration_log(t); set_err("cannot open " + path); return DPGO_ERR; }
    std::fputs("robot_id, cluster_id, num_active_robots, iteration, num_poses, bytes_received, "
               "iter_time_sec, total_time_sec, rel_change, global_cost \n", f);
    std::fflush(f);
    t->ilog.f.push_back(f);
  }
  t->ilog.bytes_received.assign(t->ag.size(), 0.0);
  t->ilog.t0 = std::chrono::steady_clock::now();
  return DPGO_OK;
}

static void log_string_all(dpgo_team_t *t, const char *s) {
  for (FILE *f : t->ilog.f) { std::fputs(s, f); std::fputc('\n', f); std::fflush(f); }
}

// one row in the log of the robot that just optimized (local index sel), after the stream has drained
static int log_iteration_row(dpgo_team_t *t, int sel, double iter_sec) {
  Agent &a = *t->ag[sel];
  dpgo_status_t st;
  if (dpgo_agent_get_status(t, a.id, &st) != DPGO_OK) return DPGO_ERR;
  double f = 0;
  if (dpgo_team_cost(t, &f)) return DPGO_ERR;
  // what a PublicPoses message per neighbour and sequence would have carried for this block update (msg/PublicPoses.msg:
  // float64[] of r x 4 per pose; the reference counts the serialized message, :1283)
  const int nseq = t->prm.acceleration ? 2 : 1;
  t->ilog.bytes_received[sel] += 8.0 * 4 * t->prm.r * (double)a.np.size() * nseq;
  const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t->ilog.t0).count();
  FILE *fp = t->ilog.f[sel];
  std::fprintf(fp, "%d,%d,%d,%d,%d,%.0f,%.9g,%.9g,%.17g,%.17g\n", a.id, 0, t->prm.num_robots, a.iter, a.n, t->ilog.bytes_received[sel],
               iter_sec, total, st.relative_change, f);
  std::fflush(fp);
  return DPGO_OK;
}

int dpgo_team_run_schedule(dpgo_team_t *t, int max_iters, int *terminated, int *weight_rounds) {
  if (sync_descs(t)) return DPGO_ERR;
  auto itl = t->id2local.find(0);
  if (itl == t->id2local.end()) { set_err("run_schedule: robot 0 (the leader) does not live in this team"); return DPGO_ERR; }
  const int lead = itl->second, len = (int)t->sched.size();
  bool in_sched = false;
  for (int s : t->sched) in_sched = in_sched || s == lead;
  if (!in_sched) { set_err("run_schedule: the leader never optimizes under this schedule"); return DPGO_ERR; }
  int done = 0, term = 0, rounds = 0;
  while (done < max_iters) {
    // iterations up to and including the leader's next block update: only then is anything decided (:206)
    int chunk = 1;
    while (t->sched[(t->iter + chunk - 1) % len] != lead) ++chunk;
    const bool reaches_leader = chunk <= max_iters - done;
    chunk = std::min(chunk, max_iters - done);
    if (t->ilog.f.empty()) {
      const int rc = dpgo_team_run(t, chunk);
      if (rc) return rc;
    } else {
      // logging: one iteration per host round trip, a row in the log of the robot that optimized
      for (int q = 0; q < chunk; ++q) {
        const int sel = t->sched[t->iter % len];
        const auto a0 = std::chrono::steady_clock::now();
        int rc = dpgo_team_run(t, 1);
        if (rc) return rc;
        rc = dpgo_team_synchronize(t);
        if (rc) return rc;
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - a0).count();
        if (log_iteration_row(t, sel, sec)) return DPGO_ERR;
      }
    }
    done += chunk;
    if (!reaches_leader) break;
    const int st = dpgo_team_should_terminate(t);
    if (st < 0) return st;
    if (st) { term = 1; if (!t->ilog.f.empty()) log_string_all(t, "TERMINATE"); break; }
    if (dpgo_agent_should_update_weights(t, 0) == 1) {
      if (!t->ilog.f.empty()) log_string_all(t, "UPDATE_WEIGHT");
      const int wr = dpgo_team_update_weights(t);
      if (wr < 0) return wr;
      ++rounds;
    }
  }
  if (terminated) *terminated = term;
  if (weight_rounds) *weight_rounds = rounds;
  return done;
}

// ---- peer access (one process per GPU, asynchronous mode): a robot's X / Y arrays are exported as a HIP IPC handle
// and imported by the processes that hold its neighbours, which then read its public poses in place -- one-sided, with
// no message and no rendezvous, over xGMI when the processes sit on different GPUs.  This is what carries the
// asynchronous (ASAPP) configuration, whose robots step from whatever neighbour poses are there
// (src/PGOAgentROS.cpp:119-127); RCCL point-to-point is two-sided and cannot.
int dpgo_agent_export_state(dpgo_team_t *t, int id, unsigned char *handle64, long long *offset_x, long long *offset_y, int *n) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
  hipIpcMemHandle_t h;
  HIPC(hipIpcGetMemHandle(&h, a->d_vec.p));
  std::memcpy(handle64, &h, 64);
  a->exported = true;
  *offset_x = a->dev.buf[B_X] - a->d_vec.p;
  *offset_y = a->dev.buf[B_Y] - a->d_vec.p;
  *n = a->n;
  return DPGO_OK;
}

int dpgo_team_import_peer(dpgo_team_t *t, int robot_id, const unsigned char *handle64, long long offset_x, long long offset_y, int n) {
  if (t->id2local.count(robot_id)) { set_err("import_peer: the robot lives in this team"); return DPGO_ERR; }
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle64, 64);
  void *p = nullptr;
  {
    // a robot imported before (its owner re-exported after its arrays moved): drop the old mapping first
    auto old = t->peers.find(robot_id);
    if (old != t->peers.end()) {
      HIPC(hipStreamSynchronize(t->stream));
      if (old->second.base) (void)hipIpcCloseMemHandle(old->second.base);
      t->peers.erase(old);
    }
  }
  HIPC(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  dpgo_team::Peer pr;
  pr.base = (double *)p; pr.off_x = (size_t)offset_x; pr.off_y = (size_t)offset_y; pr.n = n;
  t->peers[robot_id] = pr;
  // its poses are readable from now on: the agents that neighbour it need no message before they may optimize
  for (auto &a : t->ag) {
    rebuild_index(*a);
    for (size_t q = 0; q < a->np.size(); ++q)
      if (a->np[q].first == robot_id) { a->np_has[0][q] = 1; a->np_has[1][q] = 1; }
  }
  t->descs_dirty = true;
  return DPGO_OK;
}

// ---- the synchronous schedule across processes with the UPDATE token on the device (src/PGOAgentROS.cpp:136-149,
// 443-504, 1161-1189 replaced): see k_mail_signal / k_mail_wait in pose_ops.hip
static int ensure_mailbox(dpgo_team_t *t) {
  const size_t words = 2 * (size_t)t->prm.num_robots;
  if (t->d_mail.p) return 0;
  // fine-grained device memory: a peer GPU's system-scope store must become visible to a wait kernel that is ALREADY
  // running here; HIP promises cross-device visibility of ordinary (coarse-grained) allocations at kernel boundaries only
  // (advisor, round 3).  Falls back to an ordinary allocation where the runtime refuses (then the token path is only
  // sound between processes on one device, which is how it has been tested).
  {
    void *pm = nullptr;
    if (hipExtMallocWithFlags(&pm, sizeof(unsigned long long) * std::max<size_t>(words, 1), hipDeviceMallocFinegrained) == hipSuccess && pm) {
      t->d_mail.p = (unsigned long long *)pm;
      t->d_mail.n = std::max<size_t>(words, 1);
      t->mail_finegrained = true;
    } else {
      (void)hipGetLastError();
      if (t->d_mail.alloc(words)) { set_err("mailbox allocation failed"); return DPGO_ERR; }
    }
  }
  HIPC(hipMemsetAsync(t->d_mail.p, 0, sizeof(unsigned long long) * words, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  t->last_fin.assign(t->prm.num_robots, 0ull);
  return 0;
}

int dpgo_team_export_mailbox(dpgo_team_t *t, unsigned char *handle64) {
  if (ensure_mailbox(t)) return DPGO_ERR;
  hipIpcMemHandle_t h;
  HIPC(hipIpcGetMemHandle(&h, t->d_mail.p));
  std::memcpy(handle64, &h, 64);
  return DPGO_OK;
}

int dpgo_team_import_mailbox(dpgo_team_t *t, const unsigned char *handle64, const int *robot_ids, int count) {
  if (ensure_mailbox(t)) return DPGO_ERR;
  hipIpcMemHandle_t h;
  std::memcpy(&h, handle64, 64);
  void *p = nullptr;
  HIPC(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  t->mail_handles.push_back(p);
  for (int k = 0; k < count; ++k) {
    if (robot_ids[k] < 0 || robot_ids[k] >= t->prm.num_robots || t->id2local.count(robot_ids[k])) {
      set_err("import_mailbox: bad robot id");
      return DPGO_ERR;
    }
    t->peer_mail[robot_ids[k]] = (unsigned long long *)p;
  }
  return DPGO_OK;
}

// `iters` global iterations in which robot sel_ids[q] holds the token, enqueued without any host synchronisation:
// every process calls this with the same list; neighbours in other processes are read in place (dpgo_team_import_peer)
// and ordered by the mailboxes.  Per iteration k (t->iter), with acceleration:
//   wait   fin[s] >= k         for s = the token holder of k - 1, if it neighbours a local robot from another process
//                              (it has finished reading the Y this process is about to move)
//   P1     iterate(false) part of every local robot (dpgo_team_step_begin)
//   signal ready[a] = k + 1    into the mailbox of the token holder's team, for its local neighbours a
//   wait   ready[b] >= k + 1   for the remote neighbours b of a local token holder
//   P2     the block update (dpgo_team_step_end)
//   signal fin[sel] = k + 1    into the mailboxes of the token holder's remote neighbours
// Without acceleration only block updates move poses: the token holder waits for fin[b] of every remote neighbour's last
// block update (what it reads is final, and nobody still reads what it overwrites).
int dpgo_team_run_peer(dpgo_team_t *t, const int *sel_ids, int iters) {
  if (check_exchange_error(t)) return DPGO_ERR;  // (a time-out of an earlier run that nobody has looked at yet)
  if (sync_descs(t)) return DPGO_ERR;
  if (ensure_mailbox(t)) return DPGO_ERR;
  const dpgo_params_t &p = t->prm;
  const int NR = p.num_robots;
  for (auto &a : t->ag) if (!a->has_X) { set_err("run_peer before set_initial"); return DPGO_NOT_READY; }
  // remote neighbours of every local robot must be readable in place and reachable by mail
  for (auto &a : t->ag)
    for (int b : a->neighbors)
      if (!t->id2local.count(b) && (!t->peers.count(b) || !t->peer_mail.count(b))) {
        set_err("run_peer: neighbour " + std::to_string(b) + " of robot " + std::to_string(a->id) + " was not imported (state + mailbox)");
        return DPGO_ERR;
      }
  auto is_nbr = [](const Agent &a, int b) { return std::binary_search(a.neighbors.begin(), a.neighbors.end(), b); };
  auto flush_waits = [&](MailWaits &w) { launch_mail_wait(t->stream, t->d_mail.p, w, t->h_bar_err); w.count = 0; };
  auto add_wait = [&](MailWaits &w, int index, unsigned long long value) {
    for (int q = 0; q < w.count; ++q) if (w.index[q] == index) { w.value[q] = std::max(w.value[q], value); return; }
    if (w.count == MAIL_MAX) flush_waits(w);
    w.index[w.count] = index; w.value[w.count] = value; ++w.count;
  };
  auto flush_sigs = [&](MailSignals &s) { launch_mail_signal(t->stream, s); s.count = 0; };
  auto add_sig = [&](MailSignals &s, unsigned long long *word, unsigned long long value) {
    for (int q = 0; q < s.count; ++q) if (s.word[q] == word) { s.value[q] = value; return; }
    if (s.count == MAIL_MAX) flush_sigs(s);
    s.word[s.count] = word; s.value[s.count] = value; ++s.count;
  };
  int prev_sel = -1;
  {
    // (the token holder of the iteration in front of this call, if any)
    unsigned long long best = 0;
    for (int b = 0; b < NR; ++b) if (t->last_fin[b] > best) { best = t->last_fin[b]; prev_sel = b; }
    if (best != (unsigned long long)t->iter) prev_sel = -1;
  }
  for (int q = 0; q < iters; ++q) {
    const int sel_id = sel_ids[q];
    if (sel_id < 0 || sel_id >= NR) { set_err("run_peer: bad robot id in the schedule"); return DPGO_ERR; }
    const unsigned long long k = (unsigned long long)t->iter;
    auto it = t->id2local.find(sel_id);
    const int sel = (it == t->id2local.end()) ? -2 : it->second;
    const bool restart = p.acceleration && ((t->iter + 2) % p.restart_interval) == 0;
    MailWaits w{};
    MailSignals s{};
    if (p.acceleration && prev_sel >= 0 && !t->id2local.count(prev_sel))
      for (auto &a : t->ag) if (is_nbr(*a, prev_sel)) { add_wait(w, NR + prev_sel, k); break; }
    flush_waits(w);
    int rc = enqueue_team_iteration(t, false, restart, sel, 1);
    if (rc) return rc;
    if (p.acceleration && sel == -2)
      for (auto &a : t->ag) if (is_nbr(*a, sel_id)) add_sig(s, t->peer_mail[sel_id] + a->id, k + 1);
    flush_sigs(s);
    if (sel >= 0)
      for (int b : t->ag[sel]->neighbors)
        if (!t->id2local.count(b)) {
          if (p.acceleration) add_wait(w, b, k + 1);
          if (t->last_fin[b] > 0) add_wait(w, NR + b, t->last_fin[b]);
        }
    flush_waits(w);
    rc = enqueue_team_iteration(t, false, restart, sel, 2);
    if (rc) return rc;
    const bool fused = p.method == DPGO_METHOD_RGD && p.rgd_use_preconditioner && !restart && sel >= 0;
    account_iteration(t, sel, fused || t->last_iteration_folded);
    if (sel >= 0)
      for (int b : t->ag[sel]->neighbors)
        if (!t->id2local.count(b)) add_sig(s, t->peer_mail[b] + NR + sel_id, k + 1);
    flush_sigs(s);
    t->last_fin[sel_id] = k + 1;
    prev_sel = sel_id;
  }
  return DPGO_OK;
}

// diagnostic: the hand-off words of an agent's one-launch RTR solve (phase timestamps in -DDPGO_RTR_TRACE builds)
int dpgo_agent_read_rtr_handoff(dpgo_team_t *t, int id, unsigned long long *out, int n) {
  Agent *a = find_agent(t, id);
  if (!a || !a->d_rtr_bar.p) return DPGO_ERR;
  HIPC(hipStreamSynchronize(t->stream));
  HIPC(hipMemcpy(out, a->d_rtr_bar.p, sizeof(unsigned long long) * std::min(n, RTR_BAR_WORDS), hipMemcpyDeviceToHost));
  return DPGO_OK;
}

// diagnostic: `n` doubles of an agent's partial-sum scratch starting at `offset` (phase timestamps of trace builds)
int dpgo_agent_read_partials(dpgo_team_t *t, int id, int offset, double *out, int n) {
  Agent *a = find_agent(t, id);
  if (!a || offset < 0 || n < 0 || offset + n > PART_TOTAL) return DPGO_ERR;
  HIPC(hipStreamSynchronize(t->stream));
  HIPC(hipMemcpy(out, a->dev.part + offset, sizeof(double) * n, hipMemcpyDeviceToHost));
  return n;
}

int dpgo_team_get_counters(dpgo_team_t *t, double *out, int n) {
  for (auto &a : t->ag) if (refresh_rtr_result(t, *a)) return DPGO_ERR;
  for (int k = 0; k < n && k < 11; ++k) out[k] = t->counters[k];
  return 0;
}

int dpgo_agent_pull_local(dpgo_team_t *t, int id) {
  Agent *a = find_agent(t, id);
  if (!a) return DPGO_ERR;
  if (sync_descs(t)) return DPGO_ERR;
  launch_pull(t->ctx(), a->local, (int)a->shared.size());
  for (size_t q = 0; q < a->np.size(); ++q)
    if (t->id2local.count(a->np[q].first) && !t->isolated) { a->np_has[0][q] = 1; a->np_has[1][q] = 1; }
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
    else if (which == 9)  // the fused step kernel exactly as the pipelined accelerated-RGD loop launches it (state is consumed)
      launch_precond(c, a->local, n, PM_RGD_, B_X, B_GF, B_Z, 0, 0, t->prm.rgd_stepsize, 1, t->prm.num_robots, 2,
                     t->prm.restart_interval, 3);
    else launch_copy(c, -3, -1, (int)t->ag.size(), t->max_n, B_X, B_XPREV, 0);
  };
  if (which == 0) *algorithmic_bytes = precond_operator_bytes(*a) + 3.0 * vec;  // the operator once, v + X in, z out
  else if (which == 9) {
    // a mid-run iteration: M once; gradient, X, V, Y in and Y, X, V out for this agent; X, V in and Y, X out for the
    // look-ahead Nesterov step of every other agent (XPrev, the X2 snapshot and the status partials only move in the
    // last two iterations of a run)
    double others = 0;
    for (auto &b : t->ag) if (b.get() != a) others += 8.0 * r * 4 * b->n;
    *algorithmic_bytes = precond_operator_bytes(*a) + 7.0 * vec + 4.0 * others;
  }
  else *algorithmic_bytes = 8.0 * (16.0 * a->col.size() + 3.0 * r * 4 * n) + 4.0 * (a->col.size() + n + 1);
  if (which == 12 || which == 13) {
    // planning aid for DESIGN 5b: what a fork / join per iteration costs inside a hipGraph.  64 x { [plain preconditioner
    // apply of this agent  ||  cost / gradient evaluation of the NEXT local agent]  ->  one tiny kernel } captured on
    // two streams (which == 12), against the same three kernels in one chain (which == 13).  Stand-ins with the
    // durations of the step kernel's stream, E1 and E2; results are not used.
    const int other = (a->local + 1) % (int)t->ag.size();
    const int on = t->ag[other]->n;
    hipStream_t s2;
    hipEvent_t ef, ej;
    HIPC(hipStreamCreate(&s2));
    HIPC(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    HIPC(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    hipGraph_t g = nullptr;
    HIPC(hipStreamBeginCapture(t->stream, hipStreamCaptureModeThreadLocal));
    LaunchCtx c2 = c;
    c2.stream = s2;
    for (int k = 0; k < 64; ++k) {
      if (which == 12) {
        HIPC(hipEventRecord(ef, t->stream));
        HIPC(hipStreamWaitEvent(s2, ef, 0));
        launch_eval(c2, other, on, B_X, B_T1, B_T2, PART_C, eval_opts(t, 0, 0, 0));
        HIPC(hipEventRecord(ej, s2));
      } else {
        launch_eval(c, other, on, B_X, B_T1, B_T2, PART_C, eval_opts(t, 0, 0, 0));
      }
      launch_precond(c, a->local, n, PM_PLAIN_, B_X, B_GF, B_T2, 0, 0, 0.0, 0, t->prm.num_robots);
      if (which == 12) HIPC(hipStreamWaitEvent(t->stream, ej, 0));
      launch_noop(c, 1, 64);
    }
    HIPC(hipStreamEndCapture(t->stream, &g));
    hipGraphExec_t ge = nullptr;
    HIPC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    (void)hipGraphDestroy(g);
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
    HIPC(hipGraphLaunch(ge, t->stream));
    HIPC(hipEventRecord(e0, t->stream));
    for (int k = 0; k < reps; ++k) HIPC(hipGraphLaunch(ge, t->stream));
    HIPC(hipEventRecord(e1, t->stream));
    HIPC(hipEventSynchronize(e1));
    float ms = 0;
    HIPC(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(ef); (void)hipEventDestroy(ej);
    (void)hipGraphExecDestroy(ge);
    (void)hipStreamDestroy(s2);
    *avg_ms = (double)ms / (64.0 * reps);
    *algorithmic_bytes = 0;
    return 0;
  }
  if (which == 14) {
    // the one-launch iteration (step_fused.hip) launched eagerly x reps between ONE pair of events: its dispatch-to-
    // dispatch time.  The iterations are real ones (restarts included); the closing launch copies the Nesterov state back.
    // Like which == 10 it consumes the state: its last iteration has looked ahead, which a run's last iteration does not.
    const dpgo_params_t &p = t->prm;
    const int na = (int)t->ag.size(), mn = t->max_n, P = (int)t->sched.size();
    bool pipelined = true;
    {
      int total = 0;
      for (auto &b : t->ag) total += b->n;
      for (auto &b : t->ag) if ((total - b->n + precond_nblk(*b) - 1) / precond_nblk(*b) > 64) pipelined = false;
    }
    if (!pipelined || !fused_eval_eligible(t)) { set_err("one-launch iteration not available for this team"); return DPGO_ERR; }
    double others = 0;
    for (auto &b : t->ag) if (b.get() != a) others += 8.0 * r * 4 * b->n;
    *algorithmic_bytes = precond_operator_bytes(*a) + 7.0 * vec + 4.0 * others + spmm_bytes_of(t, *a);
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
    LaunchCtx cc = t->ctx();
    cc.bake_desc = true;
    auto sel_at = [&](int rep) { return t->sched[(size_t)((t->iter + rep) % P)]; };
    NestState *nest_own = t->d_nest_all.p, *nest_fe[2] = {t->d_nest_all.p + na, t->d_nest_all.p + 2 * na};
    launch_nest_pre(cc, -1, -1, na, mn, p.num_robots, p.restart_interval, 1);
    const int total_reps = ((reps + 1) & ~1) + 8;  // (even: the poses end in the primary arrays)
    reps = total_reps - 8;
    const int fd_m0 = fe_deep_m0(t);
    if (fd_m0 > 0) {
      // the deep-carried form (step_deep.hip), as a run's graph enqueues it
      int nblk_all = 0;
      for (auto &b : t->ag) nblk_all = std::max(nblk_all, (4 * b->n + 7) / 8);
      double *pacc[2] = {t->d_fd_pacc.p, t->d_fd_pacc.p + (size_t)nblk_all * p.r * 256};
      const int s0 = sel_at(0), s1 = sel_at(1), s2 = sel_at(2);
      launch_fd_prime(cc, s0, s1, s2, mn, p.num_robots, p.restart_interval, nest_own);
      launch_fd_open(cc, fd_m0, s0, s1, pacc[0]);
      for (int k = 0; k < total_reps; ++k) {
        if (k == 8) HIPC(hipEventRecord(e0, t->stream));
        const int flags = FD_IN | (k + 1 < total_reps ? FD_P : 0) | (k + 2 < total_reps ? FD_W : 0) | (k + 3 < total_reps ? FD_Y : 0);
        launch_step_fd(cc, fd_m0, sel_at(k), sel_at(k + 1), sel_at(k + 2), sel_at(k + 3), p.rgd_stepsize, p.num_robots,
                       p.restart_interval, k == 0 ? nest_own : nest_fe[k & 1], nest_fe[(k + 1) & 1], k & 1, flags,
                       pacc[k & 1], pacc[(k + 1) & 1]);
      }
    } else
    for (int k = 0; k < total_reps; ++k) {
      if (k == 8) HIPC(hipEventRecord(e0, t->stream));
      launch_step_fe(cc, sel_at(k), sel_at(k + 1), p.rgd_stepsize, p.num_robots, p.restart_interval,
                     k == 0 ? nest_own : nest_fe[k & 1], nest_fe[(k + 1) & 1], k & 1, sel_at(k + 2),
                     fe_carry_flags(t, k, total_reps, sel_at));
    }
    HIPC(hipEventRecord(e1, t->stream));
    launch_eval_stats(cc, mn, 0, 0, 1, p.num_robots, p.restart_interval, -1, sel_at(total_reps - 1), nest_fe[total_reps & 1]);
    HIPC(hipEventSynchronize(e1));
    HIPC(hipStreamSynchronize(t->stream));
    t->iter += total_reps;
    float ms = 0;
    HIPC(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_ms = (double)ms / reps;
    return 0;
  }
  if (which == 10 || which == 11) {
    // the fused step kernel inside the running iteration.  which == 10: the pipelined accelerated-RGD sequence launched
    // eagerly, (k_eval_stats, k_precond<PM_RGD>) x reps, between ONE pair of events -> average time per iteration;
    // which == 11: (k_eval_stats) x reps alone -> average time per evaluation launch.  bench.py reports the difference
    // as the step kernel's in-loop launch duration: the dispatch-to-dispatch time rocprofv3's kernel trace shows for it
    // (profiles/r02_timeline.txt).  (An event pair around every launch adds ~5 us of marker packets to each.)
    // The state is consumed; restarts are not honoured.
    const dpgo_params_t &p = t->prm;
    const int na = (int)t->ag.size(), mn = t->max_n;
    double others = 0;
    for (auto &b : t->ag) if (b.get() != a) others += 8.0 * r * 4 * b->n;
    *algorithmic_bytes = precond_operator_bytes(*a) + 7.0 * vec + 4.0 * others;  // as for which == 9
    hipEvent_t e0, e1;
    HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
    LaunchCtx cc = t->ctx();
    // the launches of the timed loop's graphs: the agent of every iteration named in its launches and its descriptor
    // passed by value where the team bakes them (schedule period <= 8), device-selected otherwise
    const int P = (int)t->sched.size();
    const bool bake = t->bake_sel && P >= 1 && P <= 8;
    cc.bake_desc = bake && t->bake_desc;
    auto sel_at = [&](int rep) { return bake ? t->sched[(size_t)((t->iter + rep) % P)] : -1; };
    launch_nest_pre(cc, -1, -1, na, mn, p.num_robots, p.restart_interval);
    for (int k = -8; k < reps; ++k) {
      if (k == 0) HIPC(hipEventRecord(e0, t->stream));
      launch_eval_stats(cc, mn, k == -8, 1, k > -8, p.num_robots, p.restart_interval, sel_at(k + 8), k > -8 ? sel_at(k + 7) : -1);
      if (which == 10) launch_precond(cc, sel_at(k + 8), mn, PM_RGD_, B_X, B_GF, B_Z, 0, 0, p.rgd_stepsize, 1, p.num_robots, 2, p.restart_interval, 3);
    }
    HIPC(hipEventRecord(e1, t->stream));
    HIPC(hipEventSynchronize(e1));
    float ms = 0;
    HIPC(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_ms = (double)ms / reps;
    return 0;
  }
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
