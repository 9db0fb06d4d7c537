// team_internal.h -- host-side state of libdpgo_hip.so shared by assembly.hip (structure + data matrices),
// solve.hip (launch sequencing of the local solves and the team schedule) and capi.hip (the C-ABI).
// Mirrors the DPGO::PGOAgent call surface consumed by src/PGOAgentROS.cpp.  All state
// (X, XPrev, Y, V, Q, G, dense preconditioner, neighbour slabs, solver scalars) lives in HBM; the host only
// sequences launches.  There is no CPU fallback: every entry point that computes fails with DPGO_ERR when
// no HIP device is usable.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/dpgo_hip.h"
#include "dpgo_dev.h"
#include "kernels.h"
#include "twolevel.h"

namespace dpgo_host {
using namespace dpgo;

inline thread_local std::string g_err;
inline void set_err(const std::string &s) { g_err = s; }

#define HIPC(expr)                                                                         \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      set_err(std::string(#expr) + ": " + hipGetErrorString(e_) + " @" + std::to_string(__LINE__)); \
      return DPGO_ERR;                                                                     \
    }                                                                                      \
  } while (0)

// Device buffers that a finished owner gives back are kept (per device, by rounded size, up to 4 GB in all) and handed to
// the next owner instead of going through hipFree / hipMalloc: unmapping and mapping the 150 MB of slabs and scratch of a
// 2500-pose two-level set-up was 7 of the 17 ms of a chordal initialisation, which builds and drops a whole team per call.
// Only buffers whose owner is DONE come here (a team is destroyed behind a drained stream; temporaries end behind a
// synchronisation): a buffer that is regrown mid-life still goes through hipFree, whose implicit device synchronisation
// the launch sequences rely on.  assembly.hip holds the pool.
size_t pool_round(size_t bytes);
void *pool_take(size_t rounded_bytes, int device);  // nullptr: nothing of that size is kept for that device
void pool_give(void *p, size_t rounded_bytes);
size_t pool_flush(int device);                      // hipFree everything kept for the device; bytes given back
size_t pool_held(int device);                       // bytes kept idle for the device (hipMemGetInfo counts them as used)

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  size_t cap_bytes = 0;  // rounded size the allocation was made with (0: not poolable, e.g. adopted memory)
  DevBuf() = default;
  // an owner: a copy would hand the same pointer to the pool twice
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), cap_bytes(o.cap_bytes) { o.p = nullptr; o.n = 0; o.cap_bytes = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; cap_bytes = o.cap_bytes; o.p = nullptr; o.n = 0; o.cap_bytes = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  // the owner is DONE with the buffer (its stream has drained): back to the pool
  void release() {
    if (p) { if (cap_bytes) pool_give(p, cap_bytes); else (void)hipFree(p); }
    p = nullptr; n = 0; cap_bytes = 0;
  }
  int alloc(size_t count) {
    if (count <= n && p) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; n = 0; cap_bytes = 0;
    const size_t want = pool_round(sizeof(T) * std::max<size_t>(count, 1));
    int dev = 0;
    (void)hipGetDevice(&dev);
    void *q = pool_take(want, dev);
    if (!q && hipMalloc(&q, want) != hipSuccess) {
      // the idle buffers of this device count as used memory: give them back and try once more
      (void)hipGetLastError();
      if (pool_flush(dev) == 0 || hipMalloc(&q, want) != hipSuccess) return -1;
    }
    p = (T *)q;
    cap_bytes = want;
    n = std::max<size_t>(count, 1);
    return 0;
  }
  int upload(const std::vector<T> &v, hipStream_t s) {
    if (alloc(v.size())) return -1;
    if (v.empty()) return 0;
    return hipMemcpyAsync(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, s) == hipSuccess ? 0 : -1;
  }
};

// grow-only pinned host buffer: copies to / from it are truly asynchronous (a pageable source or destination makes
// hipMemcpyAsync stage the transfer and block the host for every call)
// the same for pinned host memory (hipHostMalloc / hipHostFree are a fraction of a millisecond each; a team holds several)
void *pinned_take(size_t rounded_bytes, bool coherent);  // from the pool or fresh; nullptr on failure
void pinned_give(void *p, size_t rounded_bytes, bool coherent);

template <class T>
struct PinnedBuf {
  T *p = nullptr;
  size_t n = 0;
  size_t cap_bytes = 0;
  bool coh = false;
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf &) = delete;
  PinnedBuf &operator=(const PinnedBuf &) = delete;
  ~PinnedBuf() { if (p) pinned_give(p, cap_bytes, coh); }
  // coherent: fine-grained memory -- what a kernel stores there (behind a system-scope fence) becomes visible to a host
  // that polls it while the kernel's stream is still busy
  int alloc(size_t count, bool coherent = false) {
    if (count <= n && p) return 0;
    // (a buffer regrown mid-life may still be read by a queued kernel: hipHostFree waits for the device; only a finished
    // owner's buffer goes to the pool)
    if (p) (void)hipHostFree(p);
    p = nullptr; n = 0;
    const size_t want = std::max<size_t>(count + count / 2, 64);
    const size_t bytes = pool_round(sizeof(T) * want);
    p = (T *)pinned_take(bytes, coherent);
    if (!p) return -1;
    cap_bytes = bytes; coh = coherent;
    std::memset(p, 0, sizeof(T) * want);
    n = want;
    return 0;
  }
};

struct Agent {
  int id = 0, local = 0;
  std::vector<dpgo_measurement_t> odom, priv, shared;
  bool index_dirty = true, data_dirty = true;
  bool struct_uploaded = false;  // the index arrays of the current measurement structure are on the device
  int n = 0;
  int max_pose_edges = 0, max_tile_edges = 0;  // shared edges: most at one pose, most in one 64 / r-pose evaluation tile
  // neighbour pose dictionary (sorted (robot, frame)) and per-neighbour public ids
  std::vector<std::pair<int, int>> np;
  std::vector<char> np_has[2];
  std::vector<int> neighbors;
  int state = DPGO_WAIT_FOR_DATA;
  int iter = 0, instance = 0;
  bool publish_requested = false;
  bool has_X = false;
  bool exported = false;  // its X / Y arrays were handed to other processes (IPC): they must not move any more
  double mu = 0;
  int weight_update_count = 0, robust_inner_iter = 0;
  dpgo_opt_result_t opt{};
  bool opt_pending_rgd = false, last_success = true;
  // host copies of the sparse structure
  std::vector<int> rowptr, col;
  std::vector<double> qval;
  int npub = 0;
  // device storage
  DevBuf<int> d_rowptr, d_col, d_pub_pose, d_pub_ptr, d_idx, d_ell_col, d_trowptr, d_tcol, d_pub_index, d_soa_col;
  int precond = DPGO_PRECOND_DENSE;  // what this agent runs (decided when its data matrices are built)
  DevBuf<double> d_dinv;
  // two-level form of the preconditioner (twolevel.h): the dissection (kept while the sparsity pattern stands; a weight
  // update only refills the slabs), its device tables and slabs
  TLPlan tl_plan;
  std::vector<int> tl_rowptr, tl_col;  // the pattern the plan was made for
  int tl_plan_serial = 0, tl_tables_serial = -1;  // plan generation / generation the device tables were uploaded for
  std::shared_ptr<void> tl_layout_cache;  // twolevel.hip: the host layout of tl_plan (TLHostLayout), valid for ...
  int tl_layout_serial = -1;              // ... this plan generation (a weight update keeps the plan: 0.2 ms of layout per round)
  DevBuf<int> d_tl_blk, d_tl_lidx, d_tl_subptr, d_tl_subposes, d_tl_adjptr, d_tl_adjlist, d_tl_rowpose;
  DevBuf<long long> d_tl_doff, d_tl_eoff;
  DevBuf<dpgo::TLWg> d_tl_wg;
  DevBuf<double> d_tl_slabs, d_tl_u;
  DevBuf<unsigned long long> d_tl_flag;
  DevBuf<double> d_qval, d_M, d_vec, d_nbr, d_part, d_scal, d_resid, d_ell_val, d_tval, d_soa_val;
  bool has_soa = false;  // every row fits the ELL part: the blocks are also stored [slot][chunk][pose] (step_fused.hip)
  std::map<int, std::unique_ptr<DevBuf<int>>> d_pubframes, d_nbrslots;  // per neighbour, cached on the device
  std::map<int, int> n_pubframes, n_nbrslots;
  DevBuf<int> d_pub_all;  // the public frames of every neighbour, concatenated in neighbour order (one pack launch)
  int n_pub_all = 0;
  DevBuf<double> d_xfer;
  int tcg_hint = 4, outer_hint = -1;  // launch-pattern sizing from the previous solve of this agent
  int tcg_hint_o[4] = {0, 0, 0, 0};   // per outer iteration: launch pairs its tCG took last time (0 = unknown)
  // one-launch RTR solve (rtr_fused.hip): hand-off counters and partial-sum scratch of this agent's grid
  DevBuf<unsigned long long> d_rtr_bar;
  DevBuf<double> d_rtr_ws;
  int rtr_bar_n = -1;  // pose count the counters were zeroed for
  long long rtr_key[4] = {-1, -1, -1, -1};  // {grid, poses, two-level producers, plan generation} the scratch was sized / zeroed for
  DevBuf<unsigned long long> d_rtr_cum;      // running totals {solves, Hess-vecs, preconditioner applies, outer iterations}
  PinnedBuf<unsigned long long> h_rtr_cum;   // host copy of the totals + of the last solve's record (asynchronous read-back)
  PinnedBuf<dpgo::RtrState> h_rtr;
  unsigned long long rtr_seen[4] = {0, 0, 0, 0};
  bool opt_pending_rtr = false;  // a.opt / the team counters lag behind the device: refresh_rtr_result() catches up
  int rel_src = 0;  // where the last |X - XPrev|^2 partials live: 0 PART_D (per 64-pose tile), 1 PART_B[2] (fused RGD),
                    // 2 none (X untouched), 4 PART_D one double per pose (look-ahead Nesterov step)
  // status of the last iterate(true) (a9; refreshed only when the agent optimizes unless status_every_iterate):
  // where its |X - XPrev|^2 partials live (-1 never optimized, 1 PART_B[2], 2 X untouched, 5 PART_E tiles), whether
  // the solve ran, the share of converged GNC weights at that moment, and the host copy once it has been read
  // host-boundary batching (the per-agent API a ROS wrapper drives): public poses of all neighbours and both sequences
  // are fetched with ONE copy and served from this cache until the team launches anything again; neighbour poses
  // handed over by updateNeighborPoses are staged here and uploaded with one copy at the next use
  unsigned long long pub_epoch = 0;
  bool pub_pinned = false, pub_one_seq = false;  // the current public poses sit in the pinned report image (h_down), not in pub_cache
  std::map<int, std::vector<double>> pub_cache[2];
  std::vector<int> stage_slots[2];
  std::vector<int> stage_pos[2];  // [slot] position of a staged pose in stage_slots / stage_data, -1: not staged
  std::vector<double> stage_data[2];
  PinnedBuf<int> h_up_idx;       // pinned images of the staged upload and of the report that closes an iterate
  PinnedBuf<double> h_up, h_down;
  bool up_pending = false;       // an upload kernel that reads the pinned image may still be queued (cleared by the next
                                 // report the host has seen: the report kernel runs behind it on the same stream)
  DevBuf<unsigned long long> d_report_seq;  // [0] sequence number of the agent's reports (device side), [1] tile ticket of k_iterate_false
  DevBuf<int> d_pubpos_ptr, d_pubpos;       // per public pose: its places in the packed report (one per neighbour sharing it)
  unsigned long long report_seq = 0;        // ... and what the host expects next
  // a report that was enqueued and not yet read (iterate(false) does not wait for its own): what to look for and what it carries
  struct PendingReport {
    bool pending = false, one_seq = false, want_status = false, want_opt = false;
    unsigned long long expect = 0, epoch = 0;
    std::chrono::steady_clock::time_point t_launch{};
  } rep;
  int opt_rel_src = -1;
  bool opt_success = false, opt_cached = false;
  double opt_ratio = 1.0, opt_rel_change = 0.0;
  DevBuf<SharedEdgeDev> d_se;
  DevBuf<double> d_fe_coef;  // the coefficients of the shared edges once more, packed [edge][16] (step_deep.hip reads them with whole-line loads)
  std::vector<SharedEdgeDev> se_host;
  std::vector<int> se_order;          // se_host[e] describes shared[se_order[e]] (sorted by local pose, stable)
  std::vector<dpgo::EdgeDev> edges_host;  // the edge records of the residual / cost kernels (odometry, private, shared)
  DevBuf<int> d_pose_eptr;
  DevBuf<EdgeDev> d_edges;
  DevBuf<RtrState> d_st;
  AgentDev dev{};
  int nedges = 0;
};

}  // namespace dpgo_host

struct dpgo_comm;
struct dpgo_team {
  int device = 0;
  dpgo_params_t prm{};
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::vector<std::unique_ptr<dpgo_host::Agent>> ag;
  std::map<int, int> id2local;
  dpgo_host::DevBuf<dpgo::AgentDev> d_agents;
  dpgo_host::DevBuf<dpgo::TeamDev> d_team;
  dpgo_host::DevBuf<int> d_sched, d_group_ptr, d_group_members;
  std::vector<std::vector<int>> groups;  // colour classes (local agent indices), greedy colouring
  std::vector<std::vector<int>> group_ids;  // the same classes as global robot ids, members that live elsewhere included (dpgo_team_set_groups)
  std::vector<int> color_of;
  int all_group = 0;                     // index (behind the colour classes) of the class holding every local agent
  bool user_groups = false;              // groups supplied by dpgo_team_set_groups (global colouring)
  char *h_block = nullptr;                     // the pinned block h_states / h_state / h_scal / h_bar_err live in
  size_t h_block_bytes = 0;
  dpgo::RtrState *h_states = nullptr;          // pinned, one per local agent
  dpgo_host::DevBuf<double> d_tmp;  // scratch for raw manifold ops / dense factorisation
  std::vector<int> sched;
  int iter = 0;
  bool descs_dirty = true;
  int max_n = 0, max_npub = 0;
  dpgo::RtrState *h_state = nullptr;  // pinned
  double *h_scal = nullptr;     // pinned [16 x local agents] (fetch_scal uses the first 16)
  static constexpr int MAX_GRAPH_ITERS = 64;       // iterations captured in one graph (one graph per distinct count)
  static constexpr int MAX_PIPELINED_GRAPH_ITERS = 256;  // ... of the uniform pipelined accelerated-RGD sequence
  std::map<int, hipGraphExec_t> graphs;            // key: see dpgo_team_run
  std::map<int, int> graph_flip;
  bool graph_valid = false;
  double counters[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  // X / Y arrays of robots that live in other processes, imported through HIP IPC (dpgo_team_import_peer): their
  // public poses are read in place over peer access instead of travelling as messages
  struct Peer { double *base = nullptr; size_t off_x = 0, off_y = 0; int n = 0; };
  std::map<int, Peer> peers;
  // the device-side UPDATE token (dpgo_team_run_peer): this team's mailbox -- [robot] "moved its public poses for
  // iteration k" (k + 1), [num_robots + robot] "finished its block update of iteration k" (k + 1) -- written by the
  // teams that hold those robots; the mailboxes of the other teams, by robot they serve; every robot's last block update
  dpgo_host::DevBuf<unsigned long long> d_mail;
  bool mail_finegrained = false;  // d_mail is fine-grained memory (visible to a running kernel of another device)
  std::map<int, unsigned long long *> peer_mail;      // robot id -> mailbox of the team that holds it
  std::vector<void *> mail_handles;                   // imported mappings (closed with the team)
  std::vector<unsigned long long> last_fin;           // [robot] k + 1 of its last block update in a run_peer schedule
  // time-out flag of the in-kernel exchanges (pinned), CU count
  dpgo_host::DevBuf<dpgo::NestState> d_nest_all;  // NestState of local agent k at [k]: one array, so that a kernel finds any agent's
                                                   // state from the agent index alone (no descriptor round trip)
  // one-launch iterations (step_fused.hip): d_nest_all holds three copies of the NestStates -- the team's own at
  // [0, num_local), the two alternating buffers of the fused launches behind it
  int use_fused_eval = 1;  // DPGO_FUSED_EVAL=0: every pipelined iteration takes the two-launch sequence
  int use_fe_carry = 1;    // DPGO_FE_CARRY=0: every one-launch iteration forms the row products of its agent itself (no carried rows)
  int use_fe_deep = 1;     // DPGO_FE_DEEP=0: no deep-carried one-launch iterations (step_deep.hip): k_step_fe serves every run
  int prefetch_reports = 1;  // DPGO_REPORT_PREFETCH=0: the host does not prefetch an arrived report's image
  int use_report_tail = 1;  // DPGO_REPORT_TAIL=0: the report of an RGD iterate(true) stays a launch of its own (k_report)
  int use_fe_persist = 0;  // DPGO_FE_PERSIST=1: a run of deep-carried iterations is ONE persistent launch (step_persist.hip)
  dpgo_host::DevBuf<unsigned long long> d_pd_bar;  // its hand-off counters (zeroed in front of every launch)
  dpgo_host::DevBuf<double> d_fd_pacc;  // their partial sums: two alternating buffers of [workgroup][r][256] doubles
  int fe_min_n = 0;        // smallest agent the one-launch iteration serves (DPGO_FE_MIN_N; any n >= 32 works, bitwise).  0 = by
                           // measurement: 32 where every iteration finds carried rows (round 5), else 449 -- without carried rows:
                           // measured per iteration, two-launch | one-launch (profiles/experiments/fe_small.py, sphere2500, r = 5):
                           // 312 poses 0.0180 | 0.0192, 357: 0.0191 | 0.0197, 416: 0.0199 | 0.0202, 500: 0.0214 | 0.0207 ms
  // staged neighbour poses of the agent whose iterate(true) is being enqueued: its first launch (k_nest_pre) scatters them
  // dpgo_team_update_weights: (owner's copy, receiver's copy, receiver's local index) of every shared edge whose two end points
  // live in this team, valid while the agents' edge lists are the ones of shared_links_key
  struct SharedLink { dpgo_measurement_t *from, *to; int to_local; };
  std::vector<SharedLink> shared_links;
  std::vector<std::pair<const void *, size_t>> shared_links_key;
  dpgo_host::PinnedBuf<double> h_resid;  // dpgo_team_update_weights: every agent's residuals land here (pinned: the copies do not stage)
  struct PendingUpload { const int *slots = nullptr; const double *in = nullptr; int n0 = 0, n1 = 0; } pend_up;
  // the report of the iterate(true) that is being enqueued, offered to its LAST launch (dpgo_agent_iterate fills in where it
  // goes; enqueue_optimize takes it when that launch is the closing statistics evaluation of a fused RGD step)
  struct ReportOffer { bool valid = false, taken = false; dpgo::ReportTail rt; } rep_offer;
  int *h_bar_err = nullptr;
  int num_cus = 0;
  int max_lds = 160 * 1024;  // hipDeviceAttributeMaxSharedMemoryPerBlock
  int dense_max_n = 0;  // largest agent with a dense inverse (sizes the LDS chunk of the preconditioner kernel)
  std::vector<int> precond_of;  // [local agent] the form each agent runs (selects the preconditioner kernel's variant)
  std::vector<dpgo::AgentDev> h_descs;  // host copies of the uploaded descriptors (baked graphs pass them by value)
  int tl_max_wg = 0;            // most workgroups a two-level apply of this team runs
  int stage_cap = 0;            // k_eval_staged: most shared edges a 64 / r-pose tile of any agent carries (0: the plain k_eval)
  int bake_sel = 1;  // DPGO_BAKE_SEL=0: the graphs of the pipelined iteration select their agent on the device only
  int bake_desc = 1; // DPGO_BAKE_DESC=0: ... and find its descriptor in the device array instead of in their arguments
  bool last_iteration_folded = false;  // ... and enqueue_team_iteration skipped k_nest_post / k_status / k_advance for it
  bool last_rtr_folded = false;  // the last enqueue_optimize ran the one-launch solve WITH the iteration's tail
  bool rtr_validated = false;  // a one-launch solve has completed on this device (its grid is resident at once)
  int use_fused_rtr = 1;  // DPGO_FUSED_RTR=0 keeps the launch-per-step RTR sequence (solve.hip) for every agent
  // The one-launch solve is a persistent kernel whose workgroups wait for each other: two such grids on one device
  // (two teams, two processes) can each hold half the CUs and starve.  A team launches it only while it holds the
  // device's lock -- one holder per device across processes (flock on a file under /dev/shm) and across the teams of
  // this process --, taken at the launch and given back wherever the team's stream is known to have drained; a solve
  // that does not get the lock runs the launch-per-step sequence.  1: held.
  int rtr_lock_state = 0;
  int rtr_lock_fd = -1;
  // one process per GPU with the exchange carried by RCCL from inside the library (rank_exchange.cpp): the communicator,
  // who owns which robot, the staleness gate, and what every receiver holds of every sender
  struct RankExchange {
    dpgo_comm *comm = nullptr;
    std::vector<int> owner;                                   // [robot] rank that holds it
    int max_delay = 0;                                        // maxDelayedIterations (src/PGOAgentROS.cpp:136-149)
    bool loopback = false;                                    // world size 1: every neighbour pair crosses RCCL as a self-send
    std::vector<long long> version;                           // [robot] iteration at which its public poses last changed
    std::map<std::tuple<int, int, int>, long long> sent;      // (b, a, sequence) -> version of b's poses that a's rank holds
    dpgo_host::DevBuf<double> d_send, d_recv;                 // staging of one batch of messages
    long long iter_seen = -1;                                 // team iteration this bookkeeping is valid for (anything else that
                                                              // advanced the team in between invalidates `sent`)
    double counters[4] = {0, 0, 0, 0};                        // messages sent / received, bytes sent / received (this rank)
  } rx;
  // per-iteration log: one CSV per local robot in the reference's column order (src/PGOAgentROS.cpp:863-864,
  // 883-891) + global_cost; written by dpgo_team_run_schedule, which then runs one iteration per host round trip
  struct IterLog {
    std::vector<FILE *> f;                 // [local agent], empty: logging off
    std::vector<double> bytes_received;    // [local agent] public-pose payload its block updates have consumed so far
    std::chrono::steady_clock::time_point t0{};
  } ilog;
  int tl_max_sub = 0;     // largest subdomain of the two-level dissection (0: search for the size that streams the fewest bytes per
                          // apply -- tens of milliseconds on a 2500-pose graph; the chordal relaxation, which applies its
                          // operator twice, asks for a fixed size instead)
  bool isolated = false;  // no neighbour is read in place, co-resident ones included (rx.loopback)
  unsigned long long epoch = 1;  // bumped by everything that enqueues device work (every launch goes through ctx())
  dpgo::LaunchCtx ctx() {
    ++epoch;
    dpgo::LaunchCtx c{prm.r, stream, d_agents.p, d_team.p};
    c.nest_all = d_nest_all.p;
    c.dense_max_n = dense_max_n;
    c.host_precond = precond_of.empty() ? nullptr : precond_of.data();
    c.tl_max_wg = tl_max_wg;
    c.stage_cap = stage_cap;
    c.max_lds = max_lds;
    c.host_agents = h_descs.empty() ? nullptr : h_descs.data();
    c.num_agents = (int)h_descs.size();
    for (int k : precond_of) if (k == DPGO_PRECOND_TWO_LEVEL) c.any_two_level = true;
    return c;
  }
};

namespace dpgo_host {

// ---- twolevel.hip
bool tl_worthwhile(const TLPlan &pl);
int tl_build(dpgo_team *t, const std::vector<Agent *> &agents);
// workgroups of a preconditioner-type launch that leave partials in PART_B (device twin: precond_nblk)
// bytes of the operator one preconditioner apply streams: the dense inverse, the two-level slabs, or the 4 x 4 blocks
inline double precond_operator_bytes(const Agent &a) {
  if (a.precond == DPGO_PRECOND_TWO_LEVEL) return a.tl_plan.bytes;
  if (a.precond == DPGO_PRECOND_BLOCK_JACOBI) return 128.0 * a.n;
  return 8.0 * 16.0 * (double)a.n * (double)a.n;
}
inline int precond_nblk(const Agent &a) { return a.precond == DPGO_PRECOND_TWO_LEVEL ? a.tl_plan.nwg - a.tl_plan.nA : (4 * a.n + 7) / 8; }

// ---- assembly.hip
Agent *find_agent(dpgo_team *t, int id);
void rebuild_index(Agent &a);
int find_np(const Agent &a, int robot, int frame);
std::vector<int> public_ids(const Agent &a, int nbr);
std::vector<int> neighbor_ids(const Agent &a, int nbr);
int finalize_agent(dpgo_team *t, Agent &a, double *scratch);
int sync_descs(dpgo_team *t);          // structure / data matrices / descriptors up to date, staged neighbour poses uploaded
int sync_descs_noflush(dpgo_team *t);  // the same without the upload (used while poses are being staged)
int flush_stage(dpgo_team *t);
int stage_to_pinned(dpgo_team *t, Agent &a, int *n0, int *n1);

// ---- solve.hip
bool acquire_fused_rtr_lock(dpgo_team *t);
void release_fused_rtr_lock(dpgo_team *t);
double converged_ratio(const Agent &a);
void mark_optimized(dpgo_team *t, Agent &a, int rel_src, bool success);
struct OptFlags {
  int aux = 0, pull = 0;
  bool capture = false, fused = false, last_advances = false;
  int rtr_tail = 0;  // one-launch RTR solve: fold the rest of the iteration into it (bit 0: Nesterov V update; status + advance)
  int ls_tail = 0;   // RGD line search: fold the rest of the iteration into k_ls_apply (1: status + advance, 3: + Nesterov V)
  bool skip_stats = false;  // ... and leave out the closing statistics evaluation (mid-run iterations of a graph: nobody reads them)
  bool report_tail = false;  // the closing statistics evaluation is the call's last launch: it may carry the team's rep_offer
};
bool neighbor_poses_ready(const Agent &a, int aux);
EvalOpts eval_opts(const dpgo_team *t, int gmode, int aux, int advance);
double spmm_bytes_of(const dpgo_team *t, const Agent &a);
int enqueue_optimize(dpgo_team *t, int sel, const OptFlags &fl);
int enqueue_iterate(dpgo_team *t, int li, int do_opt, bool defer_advance = false);
int enqueue_team_iteration(dpgo_team *t, bool capture, bool restart, int sel, int phase, bool mid_run = false);
void account_iteration(dpgo_team *t, int sel, bool fused);
int enqueue_optimize_group(dpgo_team *t, int g);
int fetch_scal(dpgo_team *t, Agent &a);
int refresh_rgd_result(dpgo_team *t, Agent &a);
int read_ls_record(dpgo_team *t, Agent &a);
int ls_trials(const dpgo_params_t &p);
int refresh_rtr_result(dpgo_team *t, Agent &a, bool drained = false);
double robust_weight(const dpgo_params_t &p, double mu, double residual);
int compute_residuals(dpgo_team *t, Agent &a, std::vector<double> &res);

}  // namespace dpgo_host
