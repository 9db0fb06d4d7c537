// solve.hip -- launch sequencing: one agent's local solve (RTR with tCG / fused RGD), PGOAgent::iterate,
// the synchronous team iteration and the colour-parallel group update.  The host enqueues blind launch
// patterns; every decision (accept/reject, tCG termination, schedule) is taken on the device.
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include <mutex>

#include "team_internal.h"

using namespace dpgo;

namespace dpgo_host {

// ---- who may launch the persistent RTR solve on a device (see dpgo_team::rtr_lock_state)
static std::mutex g_rtr_mu;
static std::map<int, dpgo_team *> g_rtr_owner;  // device -> the team of THIS process that holds the lock

bool acquire_fused_rtr_lock(dpgo_team *t) {
  if (t->rtr_lock_state == 1) return true;  // (held since an earlier solve that nobody has waited for yet)
  std::lock_guard<std::mutex> g(g_rtr_mu);
  if (g_rtr_owner.count(t->device)) return false;  // another team of this process has a solve in flight
  // across processes: an advisory lock keyed by the device's PCI bus id (the same GPU whatever the visible-device order)
  if (t->rtr_lock_fd < 0) {
    char bus[64] = "unknown";
    (void)hipDeviceGetPCIBusId(bus, sizeof bus, t->device);
    for (char *c = bus; *c; ++c) if (*c == ':' || *c == '.' || *c == '/') *c = '_';
    const std::string path = std::string("/dev/shm/dpgo_hip_rtr_") + bus + ".lock";
    t->rtr_lock_fd = ::open(path.c_str(), O_CREAT | O_RDWR | O_NOFOLLOW | O_CLOEXEC, 0666);
  }
  // (a lock file that cannot be opened -- another user's, a restrictive umask -- means exclusion across processes
  // cannot be had: the launch-per-step sequence serves instead of an unprotected persistent grid)
  if (t->rtr_lock_fd < 0) return false;
  if (::flock(t->rtr_lock_fd, LOCK_EX | LOCK_NB) != 0) return false;  // another process runs one on this GPU
  g_rtr_owner[t->device] = t;
  t->rtr_lock_state = 1;
  return true;
}

// called wherever the team's stream is known to have drained (nothing of this team is in flight any more)
void release_fused_rtr_lock(dpgo_team *t) {
  if (t->rtr_lock_state != 1) return;
  std::lock_guard<std::mutex> g(g_rtr_mu);
  auto it = g_rtr_owner.find(t->device);
  if (it != g_rtr_owner.end() && it->second == t) g_rtr_owner.erase(it);
  if (t->rtr_lock_fd >= 0) (void)::flock(t->rtr_lock_fd, LOCK_UN);
  t->rtr_lock_state = 0;
}

// share of loop closures whose GNC weight has converged to 0 or 1 (robustOptMinConvergenceRatio,
// src/PGOAgentROSNode.cpp:214) [UPSTREAM-RECALL]
double converged_ratio(const Agent &a) {
  size_t total = a.priv.size() + a.shared.size(), conv = 0;
  for (const auto &m : a.priv) if (m.weight == 1.0 || m.weight == 0.0) ++conv;
  for (const auto &m : a.shared) if (m.weight == 1.0 || m.weight == 0.0) ++conv;
  return total ? (double)conv / (double)total : 1.0;
}

// the agent just ran iterate(true): remember where the status of this block update can be read
void mark_optimized(dpgo_team *t, Agent &a, int rel_src, bool success) {
  a.opt_rel_src = rel_src;
  a.opt_success = success;
  a.opt_cached = false;
  a.opt_ratio = (t->prm.robust_cost_type != DPGO_COST_L2) ? converged_ratio(a) : 1.0;
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

EvalOpts eval_opts(const dpgo_team *t, int gmode, int aux, int advance) {
  EvalOpts o;
  o.gmode = gmode; o.aux = aux; o.advance = advance;
  o.accel = t->prm.acceleration; o.num_robots = t->prm.num_robots; o.restart_interval = t->prm.restart_interval;
  return o;
}

int ls_trials(const dpgo_params_t &p) { return std::max(1, std::min(p.rgd_ls_max_backoffs + 1, LS_MAX_TRIALS)); }

double spmm_bytes_of(const dpgo_team *t, const Agent &a) {
  return 8.0 * (16.0 * a.col.size() + 3.0 * t->prm.r * 4 * a.n) + 4.0 * (a.col.size() + a.n + 1);
}

int enqueue_optimize(dpgo_team *t, int sel, const OptFlags &fl) {
  LaunchCtx c = t->ctx();
  // (an eager launch that names its agent: the descriptor by value -- the host copy is current right now -- spares the step
  // kernel one dependent cold round trip; captured sequences keep the device array, their descriptors may move under them)
  c.bake_desc = sel >= 0 && !fl.capture && t->bake_desc;
  const dpgo_params_t &p = t->prm;
  const int mn = (sel >= 0) ? t->ag[sel]->n : t->max_n;
  const int N4 = 4 * mn;
  const int gmode = fl.pull ? 2 : 1;
  if (p.method == DPGO_METHOD_RGD) {
    launch_eval(c, sel, mn, B_X, B_EGRAD, B_GF, PART_C, eval_opts(t, gmode, fl.aux, 0));
    if (p.rgd_line_search) {
      // backtracking line search (linesearch.hip): direction, every trial point, every trial cost in one pass over Q,
      // decision + move, statistics at the new point -- blind launches, the decision stays on the device
      const int ntr = ls_trials(p);
      int dirb = B_GF;
      if (p.rgd_use_preconditioner) {
        launch_precond(c, sel, mn, PM_PLAIN_, B_X, B_GF, B_Z, 0, 0, 0.0, 0, p.num_robots);
        dirb = B_Z;
      }
      launch_ls_trials(c, sel, mn, dirb, p.rgd_stepsize, p.rgd_ls_shrink, ntr);
      launch_ls_cost(c, sel, mn, dirb, ntr);
      launch_ls_apply(c, sel, mn, p.rgd_stepsize, p.rgd_ls_shrink, p.rgd_ls_sigma, ntr, fl.ls_tail, p.num_robots, p.restart_interval);
      if (!fl.skip_stats) launch_eval(c, sel, mn, B_X, B_EGRAD, B_GF, PART_A, eval_opts(t, 0, 0, 0));
      if (sel >= 0 && !fl.capture) {
        Agent &a = *t->ag[sel];
        if (p.rgd_use_preconditioner) { t->counters[0] += 1; t->counters[1] += precond_operator_bytes(a); }
        const int passes = (fl.skip_stats ? 1 : 2) + (ntr + 3) / 4;  // passes over the sparse operator
        t->counters[2] += passes; t->counters[3] += passes * spmm_bytes_of(t, a);
        // (without the closing evaluation PART_A holds the trial costs of k_ls_cost, not f_opt / |grad|^2: nothing may
        // read them as the solve's result)
        a.opt_pending_rgd = !fl.skip_stats;
      }
      return 0;
    }
    if (fl.fused && p.rgd_use_preconditioner) {
      // K3: preconditioner + RGD step + Nesterov V + |dX|^2 (+ the team's end-of-iteration bookkeeping);
      // K5: f_opt / gradnorm_opt on the snapshot B_X2 that K3 leaves behind
      launch_precond(c, sel, mn, PM_RGD_, B_X, B_GF, B_Z, 0, 0, p.rgd_stepsize, p.acceleration, p.num_robots,
                     fl.last_advances ? 1 : 0, p.restart_interval);
      // (mid-run iterations of a run of many -- dpgo_team_run_ranks -- leave the statistics out: nobody reads them)
      if (!fl.skip_stats) {
        if (fl.report_tail && t->rep_offer.valid && sel >= 0 && !fl.capture) {
          // (per-agent API: the report rides on this launch -- one launch and 4-5 us less in front of the host's wait)
          Agent &a = *t->ag[sel];
          ReportTail rt = t->rep_offer.rt;
          const int ppb = 64 / p.r;
          rt.ai = sel;
          rt.stat_off = PART_B + 2; rt.stat_cnt = precond_nblk(a); rt.stat_stride = PART_STRIDE;
          rt.opt_nb = (a.n + ppb - 1) / ppb;
          rt.advance = 1; rt.accel = p.acceleration; rt.num_robots = p.num_robots; rt.restart_interval = p.restart_interval;
          launch_eval_report(c, sel, mn, B_X2, B_EGRAD2, B_GF2, PART_A, eval_opts(t, 0, 0, 0), rt);
          t->rep_offer.taken = true;
        } else {
          launch_eval(c, sel, mn, B_X2, B_EGRAD2, B_GF2, PART_A, eval_opts(t, 0, 0, 0));
        }
      }
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
      if (p.rgd_use_preconditioner) { t->counters[0] += 1; t->counters[1] += precond_operator_bytes(a); }
      const bool stats = !(fl.skip_stats && fl.fused && p.rgd_use_preconditioner);
      t->counters[2] += stats ? 2 : 1; t->counters[3] += (stats ? 2 : 1) * spmm_bytes_of(t, a);
      a.opt_pending_rgd = stats;  // (without them PART_A still holds an older solve's sums: nothing may read it as this one's)
    }
    return 0;
  }
  if (fl.capture) { set_err("RTR cannot be captured"); return DPGO_ERR; }
  t->last_rtr_folded = false;
  // ---- RTR: trust-region Newton with truncated CG; scalars stay on the device
  Agent &a = *t->ag[sel];
  launch_eval(c, sel, mn, B_X, B_EGRAD, B_GF, PART_A, eval_opts(t, gmode, fl.aux, 0));
  int sp = 0;
  RtrState *hs = t->h_state;
  auto read_state = [&]() -> int {
    HIPC(hipMemcpyAsync(hs, a.dev.st + sp, sizeof(RtrState), hipMemcpyDeviceToHost, t->stream));
    HIPC(hipStreamSynchronize(t->stream));
    return 0;
  };
  auto account = [&]() {
    a.opt.success = 1;
    a.opt.f_init = hs->f_init; a.opt.gradnorm_init = hs->gn_init;
    a.opt.f_opt = hs->f1; a.opt.gradnorm_opt = hs->ngf;
    a.opt.rtr_outer_iters = hs->outer_count; a.opt.tcg_iters_total = hs->tcg_total;
    a.opt.hessvec_count = hs->hv_count; a.opt.precond_count = hs->pc_count; a.opt.accepted = hs->accepted;
    a.opt_pending_rgd = false;
    t->counters[2] += hs->hv_count + 1 + hs->outer_count;
    t->counters[3] += (hs->hv_count + 1 + hs->outer_count) * spmm_bytes_of(t, a);
  };
  const bool tl_fused = a.precond == DPGO_PRECOND_TWO_LEVEL && a.tl_plan.prod_post &&
                        rtr_fused_tl_eligible(p.r, a.tl_plan.nwg - a.tl_plan.nS2, tl_max_pre_poses(a.tl_plan), a.tl_plan.ns, t->num_cus, t->max_lds);
  const bool dense_fused = a.dev.M && rtr_fused_eligible(p.r, a.n, t->num_cus) && rtr_fused_lds_bytes(p.r, a.n) <= (size_t)t->max_lds;
  if (t->use_fused_rtr && (dense_fused || tl_fused) && acquire_fused_rtr_lock(t)) {
    // one launch for the whole solve, the preconditioner resident in LDS (rtr_fused.hip): M leaves HBM once per solve.
    // The host does not wait for it: the kernel leaves the solve's record and the agent's running totals in pinned
    // host memory, read by refresh_rtr_result() whenever somebody asks (opt result, counters) -- except for the very first
    // solve on this device, which is checked at once so that a grid that is not resident at once (another process
    // running a persistent kernel on this GPU) is met with the launch-per-step sequence instead of an error.
    // The hand-off counters depend on the grid (and, two-level, on the number of producers: the exchange counters run
    // to nA x epoch); the H-delta ring behind the partial sums is sized by 4 n r.  Zero / regrow whenever any of them
    // moved -- a re-finalised agent with more poses or a rebuilt plan on the same grid included (advisor, round 4).
    const int grid_key = tl_fused ? 100000000 + a.tl_plan.nwg - a.tl_plan.nS2 : a.n;
    const long long key[4] = {grid_key, a.n, tl_fused ? a.tl_plan.nA : 0, tl_fused ? a.tl_plan_serial : -1};
    if (a.rtr_bar_n != grid_key || std::memcmp(key, a.rtr_key, sizeof key) != 0) {
      std::memcpy(a.rtr_key, key, sizeof key);
      if (a.d_rtr_bar.alloc(RTR_BAR_WORDS) || a.d_rtr_ws.alloc(RTR_WS_DOUBLES + (size_t)RTR_RING * rtr_ring_pitch((size_t)4 * a.n * p.r)) || a.h_rtr.alloc(1) || a.h_rtr_cum.alloc(4)) {
        set_err("RTR scratch allocation failed"); return DPGO_ERR;
      }
      if (!a.d_rtr_cum.p) {
        if (a.d_rtr_cum.alloc(4)) { set_err("RTR scratch allocation failed"); return DPGO_ERR; }
        HIPC(hipMemsetAsync(a.d_rtr_cum.p, 0, sizeof(unsigned long long) * 4, t->stream));
      }
      HIPC(hipMemsetAsync(a.d_rtr_bar.p, 0, sizeof(unsigned long long) * RTR_BAR_WORDS, t->stream));
      a.rtr_bar_n = grid_key;
    }
    if (launch_rtr_solve(c, sel, a.n, a.d_rtr_bar.p, a.d_rtr_ws.p, a.d_rtr_cum.p, a.h_rtr.p, a.h_rtr_cum.p, t->h_bar_err, p.rtr_initial_radius,
                         p.gradnorm_tol, p.rtr_iterations, p.rtr_tcg_iterations, p.rtr_max_radius, fl.rtr_tail, p.num_robots,
                         p.restart_interval, tl_fused ? a.tl_plan.nwg - a.tl_plan.nS2 : 0,
                         tl_fused ? (size_t)128 * (2 * tl_max_pre_poses(a.tl_plan) + 2 * a.tl_plan.ns) : 0,
                         tl_fused ? 2 : rtr_fused_np(p.r, a.n, t->num_cus))) {
      // (LDS attribute or launch refused on this device / partition mode: the launch-per-step sequence below serves)
      t->use_fused_rtr = 0;
      goto per_step;
    }
    t->last_rtr_folded = fl.rtr_tail != 0;
    a.opt_pending_rtr = true;
    a.opt_pending_rgd = false;
    if (t->rtr_validated) return 0;
    HIPC(hipStreamSynchronize(t->stream));
    release_fused_rtr_lock(t);
    if (!*t->h_bar_err) {
      t->rtr_validated = true;
      return refresh_rtr_result(t, a);
    }
    // timed out at the first hand-off: nothing but scratch was written.  Launch-per-step from now on, this solve included
    t->last_rtr_folded = false;
    *t->h_bar_err = 0;
    t->use_fused_rtr = 0;
    a.rtr_bar_n = -1;
    a.opt_pending_rtr = false;
  }
per_step:
  if (refresh_rtr_result(t, a)) return DPGO_ERR;  // (totals of earlier one-launch solves, before a.opt is overwritten)
  launch_rtr_begin(c, sel, p.rtr_initial_radius, p.gradnorm_tol, p.rtr_iterations);
  // One outer iteration = [tCG init, (Hess-vec, step) x J, retract, evaluate, accept].  Every kernel is
  // gated by the device-side phase, so whole patterns are enqueued blindly: the expected number of outer
  // iterations first, then one read-back; more patterns only if the state says the solve is not done.
  // J launch pairs per pattern: what the tCG of the same outer iteration took in this agent's previous solve (38 % of the
  // Hess-vec launches and 21 % of the step launches were phase-gated no-ops with one J for all outer iterations,
  // profiles/experiments/idle_hist.py); a tCG that needs more continues in the next pattern's pairs
  const int Jdef = std::max(1, std::min(a.tcg_hint, p.rtr_tcg_iterations));
  int pat = 0;
  auto pattern = [&]() {
    const int Jo = (pat < 4 && a.tcg_hint_o[pat] > 0) ? a.tcg_hint_o[pat] : Jdef;
    const int J = std::max(1, std::min(Jo, p.rtr_tcg_iterations + 1));
    ++pat;
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
  for (int o = 0; o < 4; ++o) a.tcg_hint_o[o] = (o < hs->outer_count) ? hs->tcg_o[o] : 0;
  account();
  t->counters[0] += hs->pc_count; t->counters[1] += hs->pc_count * precond_operator_bytes(a);
  return 0;
}

// one PGOAgent::iterate for local agent `li` (host-driven variant used by the per-agent API)
// defer_advance: the caller's report kernel (capi.hip) advances the agent's Nesterov scalars / iteration counter, one
// launch less
int enqueue_iterate(dpgo_team *t, int li, int do_opt, bool defer_advance) {
  Agent &a = *t->ag[li];
  LaunchCtx c = t->ctx();
  const dpgo_params_t &p = t->prm;
  const bool restart = p.acceleration && ((a.iter + 2) % p.restart_interval) == 0;
  const bool fused = do_opt && p.method == DPGO_METHOD_RGD && p.rgd_use_preconditioner && !restart && !p.rgd_line_search;
  OptFlags fl;
  fl.fused = fused;
  // (a fused step's statistics evaluation is the last launch of the call when the caller's report takes the bookkeeping)
  fl.report_tail = fused && defer_advance && do_opt == 1;
  // a team that imported peers (dpgo_team_import_peer) has no messages to fill the neighbour slabs from: its per-agent
  // iterate reads every neighbour that is readable in place (imported or co-resident) in place, like the team schedule
  fl.pull = t->peers.empty() ? 0 : 1;
  int rc = 0;
  a.rel_src = 0;
  if (t->pend_up.n0 + t->pend_up.n1 > 0) {  // (dpgo_agent_iterate: the first launch below also scatters the staged poses)
    c.up_slots = t->pend_up.slots; c.up_in = t->pend_up.in; c.up_n0 = t->pend_up.n0; c.up_n1 = t->pend_up.n1;
    t->pend_up = dpgo_team::PendingUpload();
  }
  if (do_opt == 2) {
    // iterate(true) while a neighbour's poses are still missing (the delayed-message case): no local solve, X stays
    // put; under acceleration Y, V and the periodic restart are updated as in any iteration
    if (p.acceleration) {
      launch_nest_pre(c, li, li, 1, a.n, p.num_robots, p.restart_interval, 2);
      c.up_n0 = c.up_n1 = 0;
      launch_nest_post(c, li, a.n, p.num_robots, p.restart_interval);
      if (restart) launch_nest_reset(c, li, a.n);
    } else {
      launch_copy(c, li, li, 1, a.n, B_X, B_XPREV, 0);
    }
    a.rel_src = 2;
    if (!defer_advance) launch_advance(c, li, 1, p.acceleration, p.num_robots, p.restart_interval, 0);
    return 0;
  }
  if (p.acceleration) {
    launch_nest_pre(c, do_opt ? li : -2, li, 1, a.n, p.num_robots, p.restart_interval);
    c.up_n0 = c.up_n1 = 0;
    if (do_opt) {
      fl.aux = 1;
      // RTR, non-restart iteration: the one-launch solve also takes the Nesterov V update and the status partials (what the
      // team schedule folds, enqueue_team_iteration; bit 2: this agent's bookkeeping stays with the caller's report) -- two
      // launches less in front of the host's wait; enqueue_optimize says whether that solve ran
      if (p.method == DPGO_METHOD_RTR && !restart && defer_advance) fl.rtr_tail = 3 | 4;
      rc = enqueue_optimize(t, li, fl);
      if (rc) return rc;
      const bool folded = p.method == DPGO_METHOD_RTR && fl.rtr_tail != 0 && t->last_rtr_folded;
      if (!fused && !folded) launch_nest_post(c, li, a.n, p.num_robots, p.restart_interval);
      if (restart) {
        fl.aux = 0;
        rc = enqueue_optimize(t, li, fl);
        if (rc) return rc;
        launch_nest_reset(c, li, a.n);
      }
      if (fused || folded) a.rel_src = 1; else launch_status(c, li, li, 1, a.n, 1);
    }
  } else {
    launch_copy(c, li, li, 1, a.n, B_X, B_XPREV, 0);
    bool folded = false;
    if (do_opt) {
      if (p.method == DPGO_METHOD_RTR && defer_advance) fl.rtr_tail = 2 | 4;
      rc = enqueue_optimize(t, li, fl);
      if (rc) return rc;
      folded = p.method == DPGO_METHOD_RTR && fl.rtr_tail != 0 && t->last_rtr_folded;
    }
    if (fused || folded) a.rel_src = 1; else launch_status(c, li, li, 1, a.n, do_opt ? 1 : 0);
  }
  if (!defer_advance) launch_advance(c, li, 1, p.acceleration, p.num_robots, p.restart_interval, 0);
  return 0;
}

// the record of the agent's last one-launch RTR solve and the running totals of all of them (copied back
// asynchronously behind every solve) -> a.opt and the team counters
int refresh_rtr_result(dpgo_team *t, Agent &a, bool drained) {
  if (!a.opt_pending_rtr) return 0;
  if (!drained) HIPC(hipStreamSynchronize(t->stream));  // (drained: the caller has seen a report launched behind the solve)
  release_fused_rtr_lock(t);
  a.opt_pending_rtr = false;
  if (*t->h_bar_err) {
    const int code = *t->h_bar_err;  // 2 grid hand-off of the solve, 3 exchange of a two-level apply, 4 mailbox wait
    *t->h_bar_err = 0;
    t->use_fused_rtr = 0;
    a.rtr_bar_n = -1;
    set_err("an in-kernel exchange timed out (code " + std::to_string(code) + ": 2 hand-off of the one-launch RTR solve, 3 two-level "
            "preconditioner, 4 mailbox of the device-side token); the iterates since the last synchronisation are invalid");
    return DPGO_ERR;
  }
  const RtrState *hs = a.h_rtr.p;
  a.opt.success = 1;
  a.opt.f_init = hs->f_init; a.opt.gradnorm_init = hs->gn_init;
  a.opt.f_opt = hs->f1; a.opt.gradnorm_opt = hs->ngf;
  a.opt.rtr_outer_iters = hs->outer_count; a.opt.tcg_iters_total = hs->tcg_total;
  a.opt.hessvec_count = hs->hv_count; a.opt.precond_count = hs->pc_count; a.opt.accepted = hs->accepted;
  unsigned long long d[4];
  for (int k = 0; k < 4; ++k) { d[k] = a.h_rtr_cum.p[k] - a.rtr_seen[k]; a.rtr_seen[k] = a.h_rtr_cum.p[k]; }
  t->counters[0] += (double)d[2];
  t->counters[1] += (double)d[0] * precond_operator_bytes(a);  // the operator leaves HBM once per solve
  t->counters[2] += (double)(d[1] + d[0] + d[3]);
  t->counters[3] += (double)(d[1] + d[0] + d[3]) * spmm_bytes_of(t, a);
  return 0;
}

int fetch_scal(dpgo_team *t, Agent &a) {
  HIPC(hipMemcpyAsync(t->h_scal, a.dev.scal, sizeof(double) * 16, hipMemcpyDeviceToHost, t->stream));
  HIPC(hipStreamSynchronize(t->stream));
  return 0;
}

// the record k_ls_apply left in the agent's scalars (RGD line search)
int read_ls_record(dpgo_team *t, Agent &a) {
  if (!t->prm.rgd_line_search || t->prm.method != DPGO_METHOD_RGD) return 0;
  if (fetch_scal(t, a)) return DPGO_ERR;
  a.opt.ls_backoffs = (int)t->h_scal[8];
  a.opt.accepted = (int)t->h_scal[9];
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
  a.opt.precond_count = t->prm.rgd_use_preconditioner ? 1 : 0; a.opt.accepted = 1; a.opt.ls_backoffs = 0;
  a.opt_pending_rgd = false;
  return read_ls_record(t, a);
}

double robust_weight(const dpgo_params_t &p, double mu, double residual) {
  switch (p.robust_cost_type) {
    case DPGO_COST_L2: return 1.0;
    case DPGO_COST_L1: return 1.0 / residual;
    case DPGO_COST_HUBER: return residual < p.huber_threshold ? 1.0 : p.huber_threshold / residual;
    case DPGO_COST_TLS: return residual < p.tls_threshold ? 1.0 : 0.0;
    case DPGO_COST_GM: { const double a = 1.0 + residual * residual; return 1.0 / (a * a); }
    default: break;  // GNC_TLS
  }
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

// One global RBCD iteration over the agents of this team.
//   sel: local index of the agent that optimizes, -1 = device-selected (graph capture), -2 = the
//        selected agent lives on another rank (every local agent runs iterate(false)).
//   phase: 0 whole iteration; 1 = begin (everything before the neighbour exchange: Nesterov Y/X/V of all
//          local agents); 2 = end (local solve of `sel` + bookkeeping).
int enqueue_team_iteration(dpgo_team *t, bool capture, bool restart, int sel, int phase, bool mid_run) {
  LaunchCtx c = t->ctx();
  const dpgo_params_t &p = t->prm;
  const int na = (int)t->ag.size();
  const int mn = t->max_n;
  const bool fused = p.method == DPGO_METHOD_RGD && p.rgd_use_preconditioner && !restart && sel != -2 && !p.rgd_line_search;
  OptFlags fl;
  fl.pull = 1; fl.capture = capture; fl.fused = fused; fl.last_advances = fused;
  int rc = 0;
  if (phase != 2) {
    if (p.acceleration) launch_nest_pre(c, sel, -1, na, mn, p.num_robots, p.restart_interval);  // K1 (+ publishes cur_sel)
    else launch_copy(c, -3, -1, na, mn, B_X, B_XPREV, capture ? 1 : 0);
  }
  if (phase == 1) return 0;
  bool folded = false;
  if (sel != -2) {
    fl.aux = p.acceleration ? 1 : 0;
    // RTR, non-restart iteration: the one-launch solve also takes the Nesterov V update, the status partials and the
    // end-of-iteration bookkeeping (three launches less); enqueue_optimize reports whether that solve ran
    if (p.method == DPGO_METHOD_RTR && sel >= 0 && !capture && !restart) fl.rtr_tail = p.acceleration ? 3 : 2;
    // RGD with the line search, non-restart iteration: k_ls_apply also takes the Nesterov V update, the status tiles and
    // the end-of-iteration bookkeeping; mid-run iterations of a graph leave out the statistics nobody reads
    const bool ls_folded = p.method == DPGO_METHOD_RGD && p.rgd_line_search && !restart && phase == 0;
    if (ls_folded) { fl.ls_tail = p.acceleration ? 3 : 1; fl.skip_stats = mid_run; }
    if (fused && phase == 2) fl.skip_stats = mid_run;  // (the split iteration of the multi-rank runs)
    rc = enqueue_optimize(t, sel, fl);
    if (rc) return rc;
    folded = (p.method == DPGO_METHOD_RTR && t->last_rtr_folded) || ls_folded;
    if (!fused && !folded) {
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
      launch_status(c, sel, -1, 1, ns, 1);
    }
  }
  if (!fused && !folded) launch_advance(c, -1, na, p.acceleration, p.num_robots, p.restart_interval, 1);
  t->last_iteration_folded = folded && p.method == DPGO_METHOD_RTR;  // (status source: PART_B[2] for the folded RTR solve only)
  return 0;
}

// host-side bookkeeping after one global iteration in which local agent `sel` (or nobody: -2) optimized
void account_iteration(dpgo_team *t, int sel, bool fused) {
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
    mark_optimized(t, *t->ag[sel], fused ? 1 : 5, true);
  }
  t->iter += 1;
  t->counters[4] += 1;
}

// ---- colour-parallel sweeps: the agents of one colour class share no edge, so their block
// updates commute; they run in the same launches (blockIdx.y = member) and the result equals the sequential
// schedule that visits the classes in order.  Non-accelerated RBCD only (the Nesterov scalars advance per
// global iteration and do not commute).
int enqueue_optimize_group(dpgo_team *t, int g) {
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
    if (p.rgd_line_search) {
      const int ntr = ls_trials(p);
      launch_ls_trials(c, sel, mn, dirb, p.rgd_stepsize, p.rgd_ls_shrink, ntr);
      launch_ls_cost(c, sel, mn, dirb, ntr);
      launch_ls_apply(c, sel, mn, p.rgd_stepsize, p.rgd_ls_shrink, p.rgd_ls_sigma, ntr);
    } else
    launch_retract(c, sel, mn, B_X, dirb, -p.rgd_stepsize, B_X, -1);
    launch_eval(c, sel, mn, B_X, B_EGRAD, B_GF, PART_A, eval_opts(t, 0, 0, 0));
    for (int k : mem) {
      Agent &a = *t->ag[k];
      if (p.rgd_use_preconditioner) { t->counters[0] += 1; t->counters[1] += precond_operator_bytes(a); }
      t->counters[2] += 2; t->counters[3] += 2 * spmm_bytes_of(t, a);
      a.opt_pending_rgd = true;
    }
    return 0;
  }
  // The members of a class share no edge: their block updates commute, and on ONE device nothing is gained by running
  // them in the same launches -- the launch-per-step sequence below (grid.y = members) measured 0.37 ms per block update
  // against 0.30 for the one-launch solve of one agent after the other, which needs the whole device.  Where every
  // member can take the one-launch solve, the class is a sequence of those (the parallelism of a class is across GPUs:
  // DistributedRBCD.sweep_colored, one rank per member).
  if (t->use_fused_rtr) {
    bool all = true;
    for (int k : mem) {
      const Agent &a = *t->ag[k];
      const bool tl_fused = a.precond == DPGO_PRECOND_TWO_LEVEL && a.tl_plan.prod_post &&
                            rtr_fused_tl_eligible(p.r, a.tl_plan.nwg - a.tl_plan.nS2, tl_max_pre_poses(a.tl_plan), a.tl_plan.ns, t->num_cus, t->max_lds);
      const bool dense_fused = a.dev.M && rtr_fused_eligible(p.r, a.n, t->num_cus) && rtr_fused_lds_bytes(p.r, a.n) <= (size_t)t->max_lds;
      all = all && (tl_fused || dense_fused);
    }
    if (all) {
      OptFlags fl;
      fl.pull = 1;
      for (int k : mem) {
        const int rc = enqueue_optimize(t, k, fl);
        if (rc) return rc;
      }
      return 0;
    }
  }
  for (int k : mem) if (refresh_rtr_result(t, *t->ag[k])) return DPGO_ERR;
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
    t->counters[0] += hs.pc_count; t->counters[1] += hs.pc_count * precond_operator_bytes(a);
    t->counters[2] += hs.hv_count + 1 + hs.outer_count;
    t->counters[3] += (hs.hv_count + 1 + hs.outer_count) * spmm_bytes_of(t, a);
  }
  return 0;
}

}  // namespace dpgo_host
