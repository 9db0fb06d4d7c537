// kernels.h -- host-callable launch wrappers of spmm.hip / precond.hip / pose_ops.hip / dense_inverse.hip
#pragma once
#include <hip/hip_runtime.h>
#include "dpgo_dev.h"

namespace dpgo {

struct LaunchCtx {
  int r;
  hipStream_t stream;
  const AgentDev *agents;  // device array
  TeamDev *team;           // device
  int ny = 1;              // grid.y of the per-agent kernels: members of the colour class being updated
  int dense_max_n = 1 << 30;  // largest agent of the team that streams a dense inverse (0: none): sizes k_precond's LDS chunk
  bool any_two_level = false;      // the team has an agent with the two-level preconditioner
  int tl_max_wg = 0;               // ... and the most workgroups one of their applies runs
  const int *host_precond = nullptr;  // [local agent] DPGO_PRECOND_* it runs (host memory; selects the kernel variant)
  const AgentDev *host_agents = nullptr;  // [local agent] host copies of the descriptors (what d_agents holds)
  int num_agents = 0;                     // ... and how many
  bool bake_desc = false;          // pass the agent's descriptor BY VALUE where the launch names its agent (baked graphs)
  // neighbour poses staged on the host (pinned) that the FIRST launch of an iterate(true) scatters into the agent's slabs
  // itself (launch_nest_pre; one launch less than k_upload2 in front of it): slots / values, counts per sequence
  const int *up_slots = nullptr;
  const double *up_in = nullptr;
  int up_n0 = 0, up_n1 = 0;
  int max_lds = 160 * 1024;        // hipDeviceAttributeMaxSharedMemoryPerBlock of the team's device
  int stage_cap = 0;               // > 0: evaluations that assemble G stage the shared edges' operands through LDS
                                   // (k_eval_staged); the most shared edges any tile of any agent of the team carries
  const NestState *nest_all = nullptr;  // the team's NestStates, [local agent]: lets a kernel read an agent's Nesterov
                                        // state from the agent index alone, next to (not behind) its descriptor
};

// preconditioner kernel modes (see precond.hip)
constexpr int PM_PLAIN_ = 0, PM_TCG_INIT_ = 1, PM_TCG_STEP_ = 2, PM_RGD_ = 3;

struct EvalOpts {
  int gmode = 0;    // 0 G from buffer, 1 assemble from slab, 2 assemble pulling from co-resident agents
  int aux = 0;      // neighbour sequence for G assembly: 0 X, 1 auxiliary Y
  int advance = 0;  // fold the end-of-iteration bookkeeping of the whole team into this launch
  int accel = 0, num_robots = 1, restart_interval = 1;
};

// The report of the per-agent API (k_report, pose_ops.hip) as the TAIL of an evaluation launch (k_eval_report): every
// workgroup copies its share of the agent's public poses into the pinned image in front of its evaluation, the workgroup that
// draws the last ticket sums the partials, advances the agent and writes the sequence word the host polls.
struct ReportTail {
  double *out = nullptr;        // pinned host image: [8 scalars][X of the public frames][Y likewise]
  const int *frames = nullptr;  // the public frames, neighbour after neighbour
  int count = 0;
  int stat_off = 0, stat_cnt = 0, stat_stride = 0, opt_nb = 0;
  unsigned long long *seq = nullptr, *ticket = nullptr;
  unsigned long long expect = 0;  // the sequence number this report carries (the device's word + 1)
  int ai = 0, advance = 0, accel = 0, num_robots = 1, restart_interval = 1;
};

// `sel`: local agent index, or -1 = the agent the device-side schedule selects this iteration
void launch_buildG(const LaunchCtx &c, int sel, int max_npub, int aux, int pull);
void launch_pull(const LaunchCtx &c, int dst, int nshared);
void launch_eval(const LaunchCtx &c, int sel, int max_n, int xb, int egb, int gfb, int poff, const EvalOpts &o);
void launch_eval_report(const LaunchCtx &c, int sel, int max_n, int xb, int egb, int gfb, int poff, const EvalOpts &o,
                        const ReportTail &rt);
size_t eval_staged_lds_bytes(int r, int cap);
constexpr int EVS_STATIC_LDS = 8 * 1024;  // room left for k_eval_staged's static arrays under the device's LDS limit
void launch_hess(const LaunchCtx &c, int sel, int max_n, int xb, int egb, int vb, int ob, int poff);
void launch_precond(const LaunchCtx &c, int sel, int max_n, int mode, int xb, int vb, int zb, int sp, int max_inner,
                    double step, int accel, int num_robots, int advance = 0, int restart_interval = 1, int ahead = 0);
// pipelined accelerated RGD iterations: evaluation of iteration k || statistics of k-1 || bookkeeping of k-1
// eval_sel / stats_sel >= 0: the local agent of the evaluation / statistics half, known to the host (graphs that bake
// the schedule in); -1: read from the device-side schedule state
// nest_copy: this launch leaves a run of one-launch iterations (step_fused.hip): its bookkeeping workgroup copies the
// NestStates those launches advanced next to the team's own (iter / cur_sel have moved with them already)
void launch_eval_stats(const LaunchCtx &c, int max_n, int first, int has_eval, int has_stats, int num_robots,
                       int restart_interval, int eval_sel = -1, int stats_sel = -1, const NestState *nest_copy = nullptr);
// one launch per pipelined accelerated-RGD iteration (step_fused.hip): evaluation + preconditioned step + look-ahead.
// sel / next_sel: the agents of this iteration and the next (baked); nest_src / nest_dst: the NestStates [local agent]
// this launch reads / leaves advanced; parity: which copy of the poses it reads (0: B_X / B_Y, 1: their twins) -- it
// writes the other one, so consecutive launches alternate and a run of them has an even length
bool step_fe_supported(int r);
int step_fe_max_edges();
void launch_step_fe(const LaunchCtx &c, int sel, int next_sel, double step, int num_robots, int restart_interval,
                    const NestState *nest_src, NestState *nest_dst, int parity, int next2_sel = -1, int carry = 0);
int step_fe_carry_max_poses();
// step_deep.hip: the one-launch iteration with the private part of the product formed one launch early (flags FD_*:
// dpgo_dev.h).  m0: leading chunks of every agent's order that are private (step_fd_pick_m0 of the team's minimum; 0: the
// team cannot run it).  pacc_in / pacc_out: the partial sums this launch continues / leaves, [workgroup][r][256]
int step_fd_pick_m0(int min_private_chunks);
// step_persist.hip: K deep-carried iterations (and the two producing ones in front of them) in ONE persistent launch.
// d_sched / sched_len / it0: the team's schedule on the device and where the run starts in it; B, L: the graph's length and
// the schedule period (which iterations leave their statistics); bar: >= 18 * 16 zeroed 64-bit words; err: the team's
// pinned error word (5: a hand-off of this kernel timed out)
constexpr int PD_BAR_WORDS = 18 * 16;
void launch_step_pd(const LaunchCtx &c, int m0, const int *d_sched, int sched_len, int it0, int K, int B, int L, double step, int num_robots,
                    int restart_interval, const NestState *nest_src, NestState *nest_dst, unsigned long long *bar, int *err);
// behind k_fd_prime: the row products of the run's first two agents and the private partial sums of the first
void launch_fd_open(const LaunchCtx &c, int m0, int s0, int s1, double *pacc_out);
void launch_fd_prime(const LaunchCtx &c, int s0, int s1, int s2, int max_n, int num_robots, int restart_interval, const NestState *nest_src);
void launch_step_fd(const LaunchCtx &c, int m0, int sel, int next_sel, int next2_sel, int next3_sel, double step, int num_robots,
                    int restart_interval, const NestState *nest_src, NestState *nest_dst, int parity, int flags,
                    const double *pacc_in, double *pacc_out);
constexpr int LS_MAX_TRIALS = 8;
void launch_ls_trials(const LaunchCtx &c, int sel, int max_n, int dirb, double step0, double shrink, int ntrials);
void launch_ls_cost(const LaunchCtx &c, int sel, int max_n, int dirb, int ntrials);
// tail: fold the rest of a non-restart team iteration into the launch (bit 1: Nesterov V update; status tiles + advance)
void launch_ls_apply(const LaunchCtx &c, int sel, int max_n, double step0, double shrink, double sigma, int ntrials, int tail = 0,
                     int num_robots = 1, int restart_interval = 1);
void launch_tcg_hv(const LaunchCtx &c, int sel, int max_n, int sp, int max_inner);
void launch_retract(const LaunchCtx &c, int sel, int max_n, int xb, int eb, double scale, int ob, int guard_state);
void launch_project_raw(const LaunchCtx &c, const double *X, double *out, int n);
void launch_tangent_raw(const LaunchCtx &c, const double *X, const double *V, double *out, int n);
void launch_retract_raw(const LaunchCtx &c, const double *X, const double *E, double *out, int n);
void launch_nest_pre(const LaunchCtx &c, int sel, int only_agent, int num_agents, int max_n, int num_robots,
                     int restart_interval, int fused_restart = 0);
void launch_stats_nest(const LaunchCtx &c, int num_agents, int max_n, int num_robots, int restart_interval);
void launch_nest_post(const LaunchCtx &c, int sel, int max_n, int num_robots, int restart_interval);
void launch_nest_reset(const LaunchCtx &c, int sel, int max_n);
void launch_advance(const LaunchCtx &c, int only_agent, int num_agents, int accel, int num_robots, int restart_interval,
                    int bump_team, int inc = 1, int team_inc = -1 /* = inc */);
// opt != 0: the agent(s) just took a block update; the tiles are also left in PART_E (status of the last iterate(true))
void launch_status(const LaunchCtx &c, int sel, int only_agent, int num_agents, int max_n, int opt = 0);
void launch_rtr_begin(const LaunchCtx &c, int sel, double Delta0, double tol, int max_outer);
void launch_rtr_eval2(const LaunchCtx &c, int sel, int max_n, int sp);
void launch_rtr_accept(const LaunchCtx &c, int sel, int max_n, int sp, double tol, int max_outer, double max_radius);
void launch_pack(const LaunchCtx &c, const double *X, const int *frames, int count, double *out);
void launch_pack2(const LaunchCtx &c, const double *X, const double *Y, const int *frames, int count, double *out);
void launch_unpack(const LaunchCtx &c, double *slab, const int *slots, int count, const double *in);
// several pack / unpack jobs in ONE launch (grid.y = job): the slabs of one batch of rank-to-rank messages
// (rank_exchange.cpp).  pack: buf[j] <- src[j] poses idx[j][0 .. count[j]); unpack: src[j] slots idx[j][...] <- buf[j]
constexpr int XFER_MAX_SEGS = 16;
struct XferSegs {
  double *src[XFER_MAX_SEGS];
  const int *idx[XFER_MAX_SEGS];
  double *buf[XFER_MAX_SEGS];
  int count[XFER_MAX_SEGS];
  int n;
};
void launch_pack_multi(const LaunchCtx &c, const XferSegs &sg);
void launch_unpack_multi(const LaunchCtx &c, const XferSegs &sg);
// the host boundary of the per-agent API: staged neighbour poses read straight from pinned host memory; public poses,
// status / result sums and a sequence word written straight into it (see pose_ops.hip)
void launch_upload2(const LaunchCtx &c, double *slab0, double *slab1, const int *host_slots, const double *host_in, int n0, int n1);
void launch_report(const LaunchCtx &c, int ai, const int *frames, int count, double *host_out, int stat_off, int stat_cnt,
                   int stat_stride, int opt_nb, unsigned long long *seq, int advance, int accel, int num_robots,
                   int restart_interval, const int *up_slots = nullptr, const double *up_in = nullptr, int up_n0 = 0,
                   int up_n1 = 0, int one_seq = 0);
// the whole accelerated iterate(false) of one agent -- Nesterov step, staged neighbour poses in, bookkeeping, report -- in ONE launch
void launch_iterate_false(const LaunchCtx &c, int ai, int n, int num_robots, int restart_interval, const int *pubpos_ptr,
                          const int *pubpos, double *host_out, unsigned long long *seq, unsigned long long *ticket,
                          const int *up_slots, const double *up_in, int up_n0, int up_n1);
void launch_residuals(const LaunchCtx &c, int ai, int nedges);
void launch_cost(const LaunchCtx &c, int ai);
// the device-side UPDATE token (pose_ops.hip k_mail_signal / k_mail_wait): up to MAIL_MAX words per launch
constexpr int MAIL_MAX = 16;
struct MailSignals { int count; unsigned long long *word[MAIL_MAX]; unsigned long long value[MAIL_MAX]; };
struct MailWaits { int count; int index[MAIL_MAX]; unsigned long long value[MAIL_MAX]; };
void launch_mail_signal(hipStream_t s, const MailSignals &sig);
void launch_mail_wait(hipStream_t s, const unsigned long long *mail, const MailWaits &w, int *err);
void launch_noop(const LaunchCtx &c, int grid, int block);
void launch_copy(const LaunchCtx &c, int sel, int only_agent, int num_agents, int max_n, int from, int to, int publish);
void launch_bsr_to_dense(hipStream_t s, const int *rowptr, const int *col, const double *qval, int n, double shift,
                         double *A);
void launch_q_layouts(hipStream_t s, const int *rowptr, const double *qval, int n, int EW, double *ell_val, const int *trowptr,
                      double *tval, int SW, int tiles, double *soa_val);

// rtr_fused.hip: one launch per local RTR solve, the agent's preconditioner resident in LDS over the whole solve.
// bar: RTR_BAR_WORDS zero-initialised 64-bit words owned by the AGENT (the arrival counts depend on its grid);
// ws: RTR_WS_DOUBLES doubles of partial-sum scratch + RTR_RING x (r x 4n) doubles of H delta ring; err: pinned host word raised on a spin time-out.
constexpr int RTR_BAR_WORDS = 37 * 16;  // (hand-off counters, trace stamps, the two-level exchange counters: rtr_fused.hip)
constexpr int RTR_WS_DOUBLES = 7 * 512;
__host__ __device__ inline size_t rtr_ring_pitch(size_t doubles) { return (doubles + 31) / 32 * 32; }  // whole 256-byte blocks
constexpr int RTR_RING = 32;  // H delta buffers (r x 4n doubles each) behind the partial sums: one per tCG iteration, reused
                              // after RTR_RING iterations (rtr_fused.hip)
bool rtr_fused_eligible(int r, int n, int num_cus);
int rtr_fused_np(int r, int n, int num_cus);  // poses per workgroup of the dense one-launch solve (2, 3) or 0
size_t rtr_fused_lds_bytes(int r, int n);  // LDS the solve of an n-pose agent needs (checked against the device's limit)
// cum: 4 zero-initialised 64-bit words per agent: running totals {solves, Hessian-vector products, preconditioner
// applies, outer iterations} the kernel adds to; host_rec / host_cum: pinned host copies of the solve's record and of
// the totals, written by the kernel itself
int launch_rtr_solve(const LaunchCtx &c, int ai, int n, unsigned long long *bar, double *ws, unsigned long long *cum, RtrState *host_rec,
                     unsigned long long *host_cum, int *err, double Delta0,
                     double tol, int max_outer, int max_inner, double max_radius, int tail = 0, int num_robots = 1,
                     int restart_interval = 1, int tl_nwg = 0, size_t tl_dyn = 0, int np = 2);
// ... for an agent with the two-level preconditioner (tl_nwg workgroups, tl_dyn bytes of slab per workgroup)
size_t rtr_fused_tl_lds_bytes(int r, int max_pre_poses, int ns);
bool rtr_fused_tl_eligible(int r, int nwg, int max_pre_poses, int ns, int num_cus, int max_lds);
int rtr_fused_tl_fit_pairs(int r);

// dense_inverse.hip: M = (A)^-1 for a symmetric positive definite N x N column-major matrix.
// A is destroyed; work must hold N*N doubles.  Returns 0, or the (1-based) failing pivot block.
int dense_spd_inverse(hipStream_t stream, double *A, double *work, double *M, int N);
// the same for `count` matrices at once (one per agent): every step's kernels serve the whole batch (blockIdx.z).
// Returns 0, or failing pivot + (batch index << 24).
int dense_spd_inverse_batched(hipStream_t stream, int count, double *const *A, double *const *work, double *const *M,
                              const int *N, bool work_is_zero = false);

// A X^T = B^T for RR right-hand sides from the Cholesky factor only (A destroyed; the solution overwrites B [N][RR];
// Y [N][RR] is scratch).  Returns 0, or the (1-based) failing pivot.
template <int RR>
// augmented: A is (N x N) with the RR right-hand sides appended as its last RR rows (N counts them), see k_aug_extract
int dense_spd_solve(hipStream_t stream, double *A, int N, double *B, double *Y, bool augmented = false);

}  // namespace dpgo
