// rank_exchange.cpp -- the public-pose exchange between processes, one process per GPU, carried by RCCL point-to-point
// operations that THIS LIBRARY enqueues on the team stream (no host language in the loop).
//
// Replaces, for one-process-per-GPU runs, the ROS transport of the reference for this path:
//   * `PublicPoses` messages -- sent by publishPublicPoses (src/PGOAgentROS.cpp:662-690), received by
//     publicPosesCallback (:1255-1284) -- become packed r x 4 fp64 slabs (the layout of dpgo_agent_pack_public_poses_device)
//     moved with ncclSend / ncclRecv, one message per pair of ranks and direction;
//   * the staleness gate of :136-149 (maxDelayedIterations, include/dpgo_ros/PGOAgentROS.h:83) decides which slabs are
//     SENT: on this lossless transport a neighbour's copy is re-sent only when it is more than max_delayed_iterations
//     behind the neighbour's latest change;
//   * the UPDATE token (:443-504, 1161-1189) needs no message: every rank is handed the same list of token holders.
// dpgo_team_run_ranks enqueues K whole iterations per host call: [iterate(false) part of every local robot] -> [pack,
// grouped ncclSend / ncclRecv, unpack] -> [block update of the token holder], all on the team stream.
//
// RCCL is bound at run time (dlopen of librccl.so.1): a process that has already loaded RCCL -- torch.distributed's copy
// carries the same soname -- shares that copy, single-GPU users of libdpgo_hip.so never map it, and the library has no
// link-time dependency on it.  Declarations come from <rccl/rccl.h>; no function of it is linked.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <functional>
#include <mutex>
#include <tuple>

#include "team_internal.h"

using namespace dpgo_host;

namespace {

struct RcclApi {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  std::string path;
};

// (one attempt per process, whichever thread comes first; a failed attempt keeps its message for every later caller)
static bool rccl_load(RcclApi &api, std::string &why);

RcclApi *rccl() {
  static RcclApi api;
  static std::string why;
  static std::once_flag once;
  std::call_once(once, [] { if (!rccl_load(api, why)) api.handle = nullptr; });
  if (!api.handle) { set_err(why.empty() ? std::string("RCCL is not loadable") : why); return nullptr; }
  return &api;
}

static bool rccl_load(RcclApi &api, std::string &why) {
  const char *names[] = {std::getenv("DPGO_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *nm : names) {
    if (!nm || !*nm) continue;
    api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (api.handle) { api.path = nm; break; }
  }
  if (!api.handle) {
    const char *e = dlerror();  // (ONE call: dlerror() clears its state, a second call returns null)
    why = std::string("RCCL is not loadable: ") + (e ? e : "librccl.so.1 not found");
    return false;
  }
  bool ok = true;
  auto sym = [&](const char *nm) { void *p = dlsym(api.handle, nm); if (!p) ok = false; return p; };
  api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
  api.Send = (decltype(api.Send))sym("ncclSend");
  api.Recv = (decltype(api.Recv))sym("ncclRecv");
  api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
  api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
  api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
  api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  api.GetVersion = (decltype(api.GetVersion))sym("ncclGetVersion");
  if (!ok) { why = "RCCL library lacks an expected entry point"; dlclose(api.handle); api.handle = nullptr; return false; }
  return true;
}

}  // namespace

struct dpgo_comm {
  ncclComm_t comm = nullptr;
  int device = 0, rank = 0, world = 1;
  double *d_red = nullptr;  // scratch of the small reductions
};

#define NCCLC(expr)                                                                                          \
  do {                                                                                                       \
    ncclResult_t r_ = (expr);                                                                                \
    if (r_ != ncclSuccess) {                                                                                 \
      set_err(std::string(#expr) + ": " + api->GetErrorString(r_) + " @" + std::to_string(__LINE__));        \
      return DPGO_ERR;                                                                                       \
    }                                                                                                        \
  } while (0)

extern "C" {

int dpgo_comm_unique_id(unsigned char *id128) {
  RcclApi *api = rccl();
  if (!api) return DPGO_ERR;
  static_assert(sizeof(ncclUniqueId) == DPGO_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  NCCLC(api->GetUniqueId(&id));
  std::memcpy(id128, &id, sizeof id);
  return DPGO_OK;
}

dpgo_comm_t *dpgo_comm_create(int device, const unsigned char *id128, int rank, int world) {
  RcclApi *api = rccl();
  if (!api) return nullptr;
  if (world < 1 || rank < 0 || rank >= world) { set_err("comm_create: bad rank / world size"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { set_err("comm_create: no HIP device " + std::to_string(device)); return nullptr; }
  auto *c = new dpgo_comm();
  c->device = device; c->rank = rank; c->world = world;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof id);
  const ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { set_err(std::string("ncclCommInitRank: ") + api->GetErrorString(r)); delete c; return nullptr; }
  if (hipMalloc((void **)&c->d_red, sizeof(double) * 64) != hipSuccess) { set_err("comm_create: allocation failed"); (void)api->CommDestroy(c->comm); delete c; return nullptr; }
  return c;
}

void dpgo_comm_destroy(dpgo_comm_t *c) {
  if (!c) return;
  RcclApi *api = rccl();
  if (c->d_red) (void)hipFree(c->d_red);
  if (api && c->comm) (void)api->CommDestroy(c->comm);
  delete c;
}

int dpgo_comm_rank(const dpgo_comm_t *c) { return c ? c->rank : -1; }
int dpgo_comm_world(const dpgo_comm_t *c) { return c ? c->world : -1; }

int dpgo_comm_library(char *out, int cap) {
  RcclApi *api = rccl();
  if (!api) return DPGO_ERR;
  int v = 0;
  (void)api->GetVersion(&v);
  std::snprintf(out, cap, "%s (version code %d)", api->path.c_str(), v);
  return v;
}

int dpgo_comm_allreduce_sum(dpgo_comm_t *c, void *stream, double *inout, int n) {
  RcclApi *api = rccl();
  if (!api || !c) return DPGO_ERR;
  if (n < 0 || n > 64) { set_err("allreduce_sum: at most 64 doubles"); return DPGO_ERR; }
  hipStream_t s = (hipStream_t)stream;
  HIPC(hipMemcpyAsync(c->d_red, inout, sizeof(double) * n, hipMemcpyHostToDevice, s));
  NCCLC(api->AllReduce(c->d_red, c->d_red, (size_t)n, ncclDouble, ncclSum, c->comm, s));
  HIPC(hipMemcpyAsync(inout, c->d_red, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  HIPC(hipStreamSynchronize(s));
  return DPGO_OK;
}

int dpgo_comm_allreduce_max(dpgo_comm_t *c, void *stream, double *inout, int n) {
  RcclApi *api = rccl();
  if (!api || !c) return DPGO_ERR;
  if (n < 0 || n > 64) { set_err("allreduce_max: at most 64 doubles"); return DPGO_ERR; }
  hipStream_t s = (hipStream_t)stream;
  HIPC(hipMemcpyAsync(c->d_red, inout, sizeof(double) * n, hipMemcpyHostToDevice, s));
  NCCLC(api->AllReduce(c->d_red, c->d_red, (size_t)n, ncclDouble, ncclMax, c->comm, s));
  HIPC(hipMemcpyAsync(inout, c->d_red, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  HIPC(hipStreamSynchronize(s));
  return DPGO_OK;
}

int dpgo_team_attach_comm(dpgo_team_t *t, dpgo_comm_t *c, const int *owner_rank_of_robot, int max_delayed_iterations, int loopback) {
  if (!t || !c || !owner_rank_of_robot) { set_err("attach_comm: bad arguments"); return DPGO_ERR; }
  if (c->device != t->device) { set_err("attach_comm: the communicator lives on another device"); return DPGO_ERR; }
  auto &x = t->rx;
  const int NR = t->prm.num_robots;
  x.owner.assign(owner_rank_of_robot, owner_rank_of_robot + NR);
  for (int a = 0; a < NR; ++a)
    if (x.owner[a] < 0 || x.owner[a] >= c->world) { set_err("attach_comm: owner rank out of range"); return DPGO_ERR; }
  for (auto &a : t->ag)
    if (x.owner[a->id] != c->rank) { set_err("attach_comm: robot " + std::to_string(a->id) + " lives in this team but is owned by rank " + std::to_string(x.owner[a->id])); return DPGO_ERR; }
  if (loopback && c->world != 1) { set_err("attach_comm: loopback needs world size 1"); return DPGO_ERR; }
  x.comm = c;
  x.max_delay = std::max(0, max_delayed_iterations);
  x.loopback = loopback != 0;
  x.version.assign(NR, 0);
  x.sent.clear();
  x.iter_seen = t->iter;
  // loopback: every neighbour is treated as living in another process -- no pose is read in place, each one crosses
  // RCCL as a self-send -- so that the message path is exercised end to end on a one-GPU box
  if (t->isolated != x.loopback) { t->isolated = x.loopback; t->descs_dirty = true; t->graph_valid = false; }
  if (x.loopback)
    for (auto &a : t->ag) { std::fill(a->np_has[0].begin(), a->np_has[0].end(), 0); std::fill(a->np_has[1].begin(), a->np_has[1].end(), 0); }
  return DPGO_OK;
}

int dpgo_team_detach_comm(dpgo_team_t *t) {
  if (!t) return DPGO_ERR;
  if (t->stream) (void)hipStreamSynchronize(t->stream);
  if (t->isolated) { t->isolated = false; t->descs_dirty = true; t->graph_valid = false; }
  // (the staging buffers go back to their pool; everything else starts over)
  auto &x = t->rx;
  x.d_send.release(); x.d_recv.release();
  x.comm = nullptr; x.owner.clear(); x.max_delay = 0; x.loopback = false; x.version.clear(); x.sent.clear(); x.iter_seen = -1;
  for (double &c : x.counters) c = 0;
  return DPGO_OK;
}

}  // extern "C"

namespace {

struct Pair { int b, a; unsigned seqs; };  // public poses of robot b (sequences: bit 0 X, bit 1 Y) -> robot a

inline bool crosses(const dpgo_team_t *t, int b, int a) { return t->rx.loopback || t->rx.owner[b] != t->rx.owner[a]; }

// ---- the planning layer: WHO sends WHAT to WHOM in WHICH order -- host arithmetic on what a rank knows of the topology,
// shared by the real path below and by dpgo_rank_plan_simulate (the CPU test replays every rank of a world and checks that
// each send meets a receive of the same length and content order on the other side: no box here holds two GPUs).
struct Topo {
  int me = 0;
  bool loopback = false;
  const std::vector<int> *owner = nullptr;
  std::vector<int> local;                                        // robots that live on this rank, ascending
  std::function<const std::vector<int> &(int)> nbrs;            // neighbours (ascending) of a LOCAL robot
  std::function<int(int, int)> count;                           // public poses of b that a needs (b or a local); <0: no such edge
  bool is_local(int id) const { return std::binary_search(local.begin(), local.end(), id); }
  bool crosses(int b, int a) const { return loopback || (*owner)[b] != (*owner)[a]; }
};
struct Book {  // what every receiver holds of every sender (dpgo_team::RankExchange carries the same fields)
  int max_delay = 0;
  std::vector<long long> *version = nullptr;
  std::map<std::tuple<int, int, int>, long long> *sent = nullptr;
};
struct PlanSeg { int b, a, q, count; };
struct PlanMsg { size_t off = 0, len = 0; std::vector<PlanSeg> segs; };
struct Plan { std::map<int, PlanMsg> out, in; size_t tot_out = 0, tot_in = 0; };

// One batch: one message per peer rank and direction; inside a message the slabs are ordered by (receiving robot,
// sending robot, sequence), which both ends derive alike.
Plan make_plan(const Topo &T, std::vector<Pair> pairs, size_t B) {
  std::sort(pairs.begin(), pairs.end(), [](const Pair &p, const Pair &q) { return std::tie(p.a, p.b) < std::tie(q.a, q.b); });
  Plan P;
  for (const Pair &p : pairs)
    for (int q = 0; q < 2; ++q) {
      if (!((p.seqs >> q) & 1)) continue;
      if (T.is_local(p.b)) {
        const int c = T.count(p.b, p.a);
        if (c >= 0) { PlanMsg &m = P.out[(*T.owner)[p.a]]; m.segs.push_back({p.b, p.a, q, c}); m.len += (size_t)c * B; }
      }
      if (T.is_local(p.a)) {
        const int c = T.count(p.b, p.a);
        if (c >= 0) { PlanMsg &m = P.in[(*T.owner)[p.b]]; m.segs.push_back({p.b, p.a, q, c}); m.len += (size_t)c * B; }
      }
    }
  for (auto &kv : P.out) { kv.second.off = P.tot_out; P.tot_out += kv.second.len; }
  for (auto &kv : P.in) { kv.second.off = P.tot_in; P.tot_in += kv.second.len; }
  return P;
}

// every ordered pair (b -> a) of neighbouring robots that touches this rank and crosses ranks
std::vector<Pair> all_pairs_of(const Topo &T, unsigned seqs) {
  std::vector<Pair> v;
  for (int a : T.local)
    for (int nb : T.nbrs(a)) {
      if (!T.crosses(nb, a)) continue;
      v.push_back({nb, a, seqs});                          // this rank receives
      if (!T.is_local(nb)) v.push_back({a, nb, seqs});     // ... and sends (a local nb lists the pair itself)
    }
  return v;
}

// the neighbours of the token holder that live elsewhere publish to it -- those whose copy there is too old (the staleness
// gate, per sequence, evaluated alike on both ends: a deterministic function of the schedule).  Updates the book.
std::vector<Pair> token_pairs(const Topo &T, Book &K, int sel_id, unsigned seqs) {
  std::vector<Pair> v;
  auto consider = [&](int b) {
    if (!T.crosses(b, sel_id)) return;
    for (const Pair &e : v) if (e.b == b) return;
    long long behind = 0;
    bool missing = false;
    for (int s = 0; s < 2; ++s) {
      if (!((seqs >> s) & 1)) continue;
      auto f = K.sent->find({b, sel_id, s});
      if (f == K.sent->end()) missing = true;
      else behind = std::max(behind, (*K.version)[b] - f->second);
    }
    if (!missing && (behind == 0 || behind <= K.max_delay)) return;  // current, or fresh enough for the staleness gate
    for (int s = 0; s < 2; ++s) if ((seqs >> s) & 1) (*K.sent)[{b, sel_id, s}] = (*K.version)[b];
    v.push_back({b, sel_id, seqs});
  };
  if (T.is_local(sel_id)) for (int b : T.nbrs(sel_id)) consider(b);
  for (int a : T.local) {
    if (a == sel_id) continue;
    const std::vector<int> &nb = T.nbrs(a);
    if (std::binary_search(nb.begin(), nb.end(), sel_id)) consider(a);
  }
  return v;
}

Topo topo_of(dpgo_team_t *t) {
  Topo T;
  T.me = t->rx.comm ? t->rx.comm->rank : 0;
  T.loopback = t->rx.loopback;
  T.owner = &t->rx.owner;
  for (auto &a : t->ag) T.local.push_back(a->id);
  std::sort(T.local.begin(), T.local.end());
  T.nbrs = [t](int id) -> const std::vector<int> & { return t->ag[t->id2local[id]]->neighbors; };
  T.count = [t](int b, int a) -> int {
    auto ib = t->id2local.find(b);
    if (ib != t->id2local.end()) { auto &m = t->ag[ib->second]->n_pubframes; auto f = m.find(a); return f == m.end() ? -1 : f->second; }
    auto ia = t->id2local.find(a);
    if (ia != t->id2local.end()) { auto &m = t->ag[ia->second]->n_nbrslots; auto f = m.find(b); return f == m.end() ? -1 : f->second; }
    return -1;
  };
  return T;
}
Book book_of(dpgo_team_t *t) { Book K; K.max_delay = t->rx.max_delay; K.version = &t->rx.version; K.sent = &t->rx.sent; return K; }

// executes one planned batch on the team stream: ONE pack launch per 16 slabs, one grouped ncclSend / ncclRecv per peer rank
// and direction, ONE unpack launch per 16 slabs
int exchange_pairs(dpgo_team_t *t, std::vector<Pair> pairs) {
  RcclApi *api = rccl();
  if (!api) return DPGO_ERR;
  auto &x = t->rx;
  if (pairs.empty()) return 0;
  const size_t B = (size_t)4 * t->prm.r;
  const Plan P = make_plan(topo_of(t), std::move(pairs), B);
  // (everything that can fail on this rank alone is checked before anything is packed or posted: a rank that leaves between
  // its pack and its sends would let its peers wait in ncclRecv for ever)
  for (auto &kv : P.out)
    for (const PlanSeg &g : kv.second.segs) {
      const Agent &sb = *t->ag[t->id2local[g.b]];
      if (!sb.has_X) { set_err("exchange: robot " + std::to_string(sb.id) + " has no iterate yet"); return DPGO_NOT_READY; }
    }
  // (the staging buffers may have to grow before any kernel is given a pointer into them)
  if (P.tot_out > x.d_send.n || P.tot_in > x.d_recv.n) {
    HIPC(hipStreamSynchronize(t->stream));  // (operations of an earlier batch may still read / write the old buffers)
    if (x.d_send.alloc(P.tot_out + P.tot_out / 2) || x.d_recv.alloc(P.tot_in + P.tot_in / 2)) { set_err("exchange: allocation failed"); return DPGO_ERR; }
  }
  XferSegs sg{};
  auto flush_pack = [&]() { if (sg.n) launch_pack_multi(t->ctx(), sg); sg.n = 0; };
  for (auto &kv : P.out) {
    size_t at = kv.second.off;
    for (const PlanSeg &g : kv.second.segs) {
      Agent &sb = *t->ag[t->id2local[g.b]];
      if (sg.n == XFER_MAX_SEGS) flush_pack();
      sg.src[sg.n] = sb.dev.buf[g.q ? B_Y : B_X]; sg.idx[sg.n] = sb.d_pubframes[g.a]->p; sg.count[sg.n] = g.count; sg.buf[sg.n] = x.d_send.p + at;
      ++sg.n;
      at += (size_t)g.count * B;
    }
  }
  flush_pack();
  NCCLC(api->GroupStart());
  {
    // (a failing call inside the group still closes it: an open group swallows every later RCCL call of the process)
    ncclResult_t bad = ncclSuccess;
    for (auto &kv : P.out)
      if (kv.second.len && bad == ncclSuccess) {
        bad = api->Send(x.d_send.p + kv.second.off, kv.second.len, ncclDouble, kv.first, x.comm->comm, t->stream);
        x.counters[0] += 1; x.counters[2] += 8.0 * kv.second.len;
      }
    for (auto &kv : P.in)
      if (kv.second.len && bad == ncclSuccess) {
        bad = api->Recv(x.d_recv.p + kv.second.off, kv.second.len, ncclDouble, kv.first, x.comm->comm, t->stream);
        x.counters[1] += 1; x.counters[3] += 8.0 * kv.second.len;
      }
    const ncclResult_t end = api->GroupEnd();
    NCCLC(bad);
    NCCLC(end);
  }
  auto flush_unpack = [&]() { if (sg.n) launch_unpack_multi(t->ctx(), sg); sg.n = 0; };
  for (auto &kv : P.in) {
    size_t at = kv.second.off;
    for (const PlanSeg &g : kv.second.segs) {
      Agent &ra = *t->ag[t->id2local[g.a]];
      if (sg.n == XFER_MAX_SEGS) flush_unpack();
      sg.src[sg.n] = ra.dev.nbr[g.q]; sg.idx[sg.n] = ra.d_nbrslots[g.b]->p; sg.count[sg.n] = g.count; sg.buf[sg.n] = x.d_recv.p + at;
      ++sg.n;
      at += (size_t)g.count * B;
      for (size_t sl = 0; sl < ra.np.size(); ++sl) if (ra.np[sl].first == g.b) ra.np_has[g.q][sl] = 1;
    }
  }
  flush_unpack();
  return 0;
}

std::vector<Pair> all_pairs(dpgo_team_t *t, unsigned seqs) { return all_pairs_of(topo_of(t), seqs); }

}  // namespace

extern "C" {

int dpgo_team_exchange_all_ranks(dpgo_team_t *t) {
  if (!t->rx.comm) { set_err("exchange_all_ranks: no communicator attached"); return DPGO_ERR; }
  if (sync_descs(t)) return DPGO_ERR;
  auto &x = t->rx;
  // co-resident neighbours are read in place
  if (!t->isolated) {
    LaunchCtx c = t->ctx();
    for (auto &a : t->ag) {
      launch_pull(c, a->local, (int)a->shared.size());
      for (size_t q = 0; q < a->np.size(); ++q)
        if (t->id2local.count(a->np[q].first)) { a->np_has[0][q] = 1; a->np_has[1][q] = 1; }
    }
  }
  std::vector<Pair> v = all_pairs(t, 3u);
  if (x.iter_seen != t->iter) x.sent.clear();  // (the team moved outside dpgo_team_run_ranks: only what is sent now is known)
  x.iter_seen = t->iter;
  for (const Pair &p : v) { x.sent[{p.b, p.a, 0}] = x.version[p.b]; x.sent[{p.b, p.a, 1}] = x.version[p.b]; }
  return exchange_pairs(t, v);
}

// `iters` global iterations in which robot sel_ids[q] holds the UPDATE token.  Every rank that owns a robot calls this with
// the same list.  Same iterates as dpgo_team_step_begin / [messages] / dpgo_team_step_end driven from the host per step.
int dpgo_team_run_ranks(dpgo_team_t *t, const int *sel_ids, int iters) {
  auto &x = t->rx;
  if (!x.comm) { set_err("run_ranks: no communicator attached"); return DPGO_ERR; }
  const dpgo_params_t &p = t->prm;
  const int NR = p.num_robots;
  for (int q = 0; q < iters; ++q) if (sel_ids[q] < 0 || sel_ids[q] >= NR) { set_err("run_ranks: bad robot id in the schedule"); return DPGO_ERR; }
  if (sync_descs(t)) return DPGO_ERR;
  for (auto &a : t->ag) if (!a->has_X) { set_err("run_ranks before set_initial"); return DPGO_NOT_READY; }
  // nothing crosses a rank: the device-resident schedule (hipGraphs, one launch per iteration) serves, provided the list
  // is the team's own schedule from where it stands
  bool any_cross = false;
  for (auto &a : t->ag) for (int nb : a->neighbors) any_cross = any_cross || crosses(t, nb, a->id);
  if (!any_cross && (int)t->ag.size() == NR && !t->sched.empty()) {
    bool same = true;
    const int P = (int)t->sched.size();
    for (int q = 0; q < iters && same; ++q) same = t->ag[t->sched[(t->iter + q) % P]]->id == sel_ids[q];
    if (same) {
      const int rc = dpgo_team_run(t, iters);
      if (rc == 0) { for (int a = 0; a < NR; ++a) x.version[a] = t->iter; x.iter_seen = t->iter; }
      return rc;
    }
  }
  // iterations driven by anything else since the bookkeeping was last valid (host-driven steps, dpgo_team_run): what the
  // receivers hold is unknown -- every slab is sent once
  if (x.iter_seen != t->iter) x.sent.clear();
  const unsigned seqs = p.acceleration ? 3u : 1u;
  for (int q = 0; q < iters; ++q) {
    const int sel_id = sel_ids[q];
    auto it = t->id2local.find(sel_id);
    const int sel = (it == t->id2local.end()) ? -2 : it->second;
    const long long k = t->iter;
    const bool restart = p.acceleration && ((t->iter + 2) % p.restart_interval) == 0;
    // iterate(false) moves X and Y of everyone but the token holder BEFORE the token holder's neighbours publish them
    if (p.acceleration) for (int a = 0; a < NR; ++a) if (a != sel_id) x.version[a] = k + 1;
    int rc = enqueue_team_iteration(t, false, restart, sel, 1);
    if (rc) return rc;
    // the neighbours of the token holder that live elsewhere publish to it -- those whose copy there is too old
    Book K = book_of(t);
    std::vector<Pair> v = token_pairs(topo_of(t), K, sel_id, seqs);
    rc = exchange_pairs(t, v);
    if (rc) return rc;
    if (sel >= 0 && !neighbor_poses_ready(*t->ag[sel], p.acceleration ? 1 : 0)) { set_err("run_ranks: neighbour poses missing (call dpgo_team_exchange_all_ranks once after set_initial)"); return DPGO_NOT_READY; }
    rc = enqueue_team_iteration(t, false, restart, sel, 2, /* mid_run: no statistics but for the last of the call */ q + 1 < iters);
    if (rc) return rc;
    const bool fused = p.method == DPGO_METHOD_RGD && p.rgd_use_preconditioner && !restart && sel >= 0 && !p.rgd_line_search;
    account_iteration(t, sel, fused || t->last_iteration_folded);
    x.version[sel_id] = k + 1;
    x.iter_seen = t->iter;
  }
  return DPGO_OK;
}

// BASELINE configs[4] across ranks, the lockstep instance of the asynchronous (ASAPP) mode (src/PGOAgentROS.cpp:119-127):
// per tick every boundary slab (X only) crosses the ranks in ONE batch of point-to-point operations, then every rank steps
// all its robots in the same launches (dpgo_team_run_simultaneous).  Every rank that owns a robot calls it with the same count.
int dpgo_team_run_simultaneous_ranks(dpgo_team_t *t, int ticks) {
  auto &x = t->rx;
  if (!x.comm) { set_err("run_simultaneous_ranks: no communicator attached"); return DPGO_ERR; }
  if (sync_descs(t)) return DPGO_ERR;
  for (int k = 0; k < ticks; ++k) {
    int rc = exchange_pairs(t, all_pairs(t, 1u));
    if (rc) return rc;
    rc = dpgo_team_run_simultaneous(t, 1);
    if (rc) return rc;
  }
  for (auto &v : x.version) v = t->iter;
  x.sent.clear();  // (X crossed, Y did not: a later accelerated schedule sends both)
  x.iter_seen = t->iter;
  return DPGO_OK;
}

// one colour class of a colour-parallel sweep across ranks: every member -- on whatever rank it lives --
// receives its neighbours' public poses (those that moved since it last got them), then all members take their block
// update at once (dpgo_team_run_group).  Needs dpgo_team_set_groups with the GLOBAL classes; `count` = global size of the
// class.  Every rank that owns a robot calls it with the same arguments.
int dpgo_team_run_group_ranks(dpgo_team_t *t, int g, int count) {
  auto &x = t->rx;
  if (!x.comm) { set_err("run_group_ranks: no communicator attached"); return DPGO_ERR; }
  if (g < 0 || g >= (int)t->group_ids.size()) { set_err("run_group_ranks: bad group (dpgo_team_set_groups first)"); return DPGO_ERR; }
  if (sync_descs(t)) return DPGO_ERR;
  if (x.iter_seen != t->iter) x.sent.clear();
  const std::vector<int> &members = t->group_ids[g];
  auto is_member = [&](int id) { return std::find(members.begin(), members.end(), id) != members.end(); };
  std::vector<Pair> v;
  auto consider = [&](int b, int a) {  // b's public poses -> member a
    if (!crosses(t, b, a)) return;
    for (const Pair &e : v) if (e.b == b && e.a == a) return;
    auto f = x.sent.find({b, a, 0});
    if (f != x.sent.end() && x.version[b] - f->second <= (long long)x.max_delay) return;
    x.sent[{b, a, 0}] = x.version[b];
    v.push_back({b, a, 1u});
  };
  for (auto &ag : t->ag)
    for (int nb : ag->neighbors) {
      if (is_member(ag->id)) consider(nb, ag->id);     // a local member receives
      if (is_member(nb)) consider(ag->id, nb);          // a local robot sends to a member (wherever it lives)
    }
  int rc = exchange_pairs(t, v);
  if (rc) return rc;
  rc = dpgo_team_run_group(t, g, count);
  if (rc) return rc;
  for (int id : members) x.version[id] = t->iter;
  x.iter_seen = t->iter;
  return DPGO_OK;
}

// The planning layer replayed WITHOUT a device for one rank of a world (host arithmetic only): the full exchange that
// follows set_initial, then `iters` iterations of the token schedule with the staleness gate.  npub[b * N + a] = public poses
// of robot b that robot a needs (0: not neighbours).  out[(1 + iters) x world x 4]: per batch and peer rank {doubles sent,
// doubles received, hash of the sent slabs' (b, a, sequence, count) in order, hash of the received ones'}.  A test calls it for
// every rank and checks that what r sends to p is what p receives from r, batch by batch.
int dpgo_rank_plan_simulate(int num_robots, int world, int rank, const int *owner, const int *npub, int acceleration,
                            int max_delayed_iterations, int r, const int *sel_ids, int iters, long long *out) {
  if (num_robots <= 0 || world <= 0 || rank < 0 || rank >= world || !owner || !npub || !out) { set_err("plan_simulate: bad arguments"); return DPGO_ERR; }
  const int N = num_robots;
  std::vector<int> own(owner, owner + N);
  std::vector<std::vector<int>> nb(N);
  for (int a = 0; a < N; ++a) for (int b = 0; b < N; ++b) if (a != b && (npub[a * N + b] > 0 || npub[b * N + a] > 0)) nb[a].push_back(b);
  Topo T;
  T.me = rank; T.loopback = false; T.owner = &own;
  for (int a = 0; a < N; ++a) if (own[a] == rank) T.local.push_back(a);
  T.nbrs = [&nb](int id) -> const std::vector<int> & { return nb[id]; };
  T.count = [&](int b, int a) -> int { return npub[b * N + a] > 0 ? npub[b * N + a] : -1; };
  std::vector<long long> version(N, 0);
  std::map<std::tuple<int, int, int>, long long> sent;
  Book K; K.max_delay = std::max(0, max_delayed_iterations); K.version = &version; K.sent = &sent;
  const size_t B = (size_t)4 * r;
  auto record = [&](int row, const Plan &P) {
    long long *o = out + (size_t)row * world * 4;
    for (int p = 0; p < world * 4; ++p) o[p] = 0;
    auto hash = [](const PlanMsg &m) {
      unsigned long long h = 1469598103934665603ull;
      for (const PlanSeg &g : m.segs) for (int v : {g.b, g.a, g.q, g.count}) { h ^= (unsigned long long)(unsigned)v; h *= 1099511628211ull; }
      return (long long)(h >> 1);
    };
    for (auto &kv : P.out) { o[kv.first * 4 + 0] = (long long)kv.second.len; o[kv.first * 4 + 2] = hash(kv.second); }
    for (auto &kv : P.in) { o[kv.first * 4 + 1] = (long long)kv.second.len; o[kv.first * 4 + 3] = hash(kv.second); }
  };
  {
    std::vector<Pair> v = all_pairs_of(T, 3u);
    for (const Pair &p : v) { sent[{p.b, p.a, 0}] = version[p.b]; sent[{p.b, p.a, 1}] = version[p.b]; }
    record(0, make_plan(T, v, B));
  }
  const unsigned seqs = acceleration ? 3u : 1u;
  for (int q = 0; q < iters; ++q) {
    const int sel_id = sel_ids[q];
    if (sel_id < 0 || sel_id >= N) { set_err("plan_simulate: bad robot id in the schedule"); return DPGO_ERR; }
    const long long k = q;
    if (acceleration) for (int a = 0; a < N; ++a) if (a != sel_id) version[a] = k + 1;
    record(1 + q, make_plan(T, token_pairs(T, K, sel_id, seqs), B));
    version[sel_id] = k + 1;
  }
  return DPGO_OK;
}

/* global cost across ranks: owned-edge partial sums of this team (NULL: a rank without robots contributes 0) + one
 * 1-double all-reduce.  Collective over the communicator. */
int dpgo_comm_global_cost(dpgo_comm_t *c, dpgo_team_t *t, void *stream, double *f) {
  // (a rank whose own part fails still takes part in the reduction -- its peers are already waiting in it -- and contributes
  // NaN: every rank then sees that the sum is not one)
  double part = 0;
  bool failed = false;
  if (t) {
    if (t->rx.comm && dpgo_team_exchange_all_ranks(t)) failed = true;
    if (!failed && dpgo_team_cost(t, &part)) failed = true;
  }
  const std::string why = failed ? g_err : std::string();
  if (failed) part = std::nan("");
  if (dpgo_comm_allreduce_sum(c, t ? (void *)t->stream : stream, &part, 1)) return DPGO_ERR;
  *f = part;
  if (failed) { set_err(why); return DPGO_ERR; }
  if (part != part) { set_err("global cost: another rank failed to evaluate its part"); return DPGO_ERR; }
  return DPGO_OK;
}

int dpgo_team_comm_counters(dpgo_team_t *t, double *out4) {
  for (int k = 0; k < 4; ++k) out4[k] = t->rx.counters[k];
  return DPGO_OK;
}

}  // extern "C"
