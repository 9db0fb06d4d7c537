// iter_fused.hip -- ONE launch per pipelined accelerated-RGD iteration: the cost /
// gradient evaluation of iteration k (k_eval_stats' evaluation half), a grid-wide hand-off, and the preconditioned
// step with its Nesterov tail (k_precond<PM_RGD>, advance = 2) in the same kernel.  What the fusion buys:
//   * the preconditioner M = (Q + shift I)^-1 -- the only HBM stream of the iteration, 32 MB for a 500-pose agent --
//     is requested by three of the four waves of every workgroup at launch and arrives UNDER the evaluation and the
//     hand-off instead of after a kernel boundary;
//   * one kernel boundary per iteration instead of two.
// Hand-off (cdna_hip_programming.md Guideline 16, R1 form): the evaluation tiles store the Riemannian gradient
// write-through (agent-scope relaxed stores = global_store sc1), drain, and their workgroup arrives on a counter that
// is sharded by blockIdx % 8 (the XCD the block is observed to run on -- a speed assumption only: nobody relies on a
// neighbour's cache write-back); the last arriver of a shard arrives on the top counter, the last of those publishes
// the epoch to eight generation words; one lane per workgroup polls its shard's word (relaxed, s_sleep) and the
// gradient is then staged with agent-scope loads (sc1).  The wave that arrives and polls holds no outstanding M loads
// (vmcnt returns in order): it evaluates, arrives, polls, and only then requests its quarter of the slab, which lands
// while the vector is staged.  Counters are monotonic (arrivals * epoch, 64 bit) and every workgroup arrives exactly
// once per launch, so the state heals by itself even if a spin times out (bounded: the kernel raises *err and carries
// on with garbage instead of hanging the GPU; the host checks the flag).
// The run state (iteration counter, selected agent, Nesterov scalars) is read ONCE, before the hand-off, by every
// workgroup and advanced in registers; workgroup 0 writes the advanced state back AFTER the hand-off, when nobody
// reads it any more.  Arithmetic, operands and summation order are those of the two-launch sequence: the iterates are
// bitwise identical to it.
#include <cstdlib>

#include "kernel_common.h"

namespace dpgo {

constexpr int BAR_LINE = 16;            // 64-bit words per 128-byte line
constexpr int BAR_TOP = 8 * BAR_LINE;   // cnt[g] at g * BAR_LINE
constexpr int BAR_GEN = 9 * BAR_LINE;   // gen[g] at BAR_GEN + g * BAR_LINE
constexpr int BAR_EPOCH = 17 * BAR_LINE;
constexpr int BAR_SPIN_LIMIT = 1 << 18;

template <class T>
__device__ __forceinline__ T ld_once(const T *p) {  // one read, never re-materialised behind the hand-off
  return __hip_atomic_load(const_cast<T *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// optional per-phase timestamps (100 MHz wall clock) of two workgroups, build with -DDPGO_ITER_TRACE: hardware block 0
// (owns an evaluation tile) writes words [TRACE .. TRACE+15], hardware block 100 words [TRACE+16 .. TRACE+31]
constexpr int BAR_TRACE = 17 * BAR_LINE + 2;
#ifdef DPGO_ITER_TRACE
#define ITER_STAMP(k) do { if (lane == 0 && wave == 0 && (hw == 0 || hw == 100)) bar[BAR_TRACE + (hw ? 16 : 0) + (k)] = wall_clock64(); } while (0)
#else
#define ITER_STAMP(k) do { } while (0)
#endif

#ifndef DPGO_ITER_STAGE_BATCHES
#define DPGO_ITER_STAGE_BATCHES 1
#endif

struct IterBcast {  // run state read by the polling wave before the hand-off, handed to the other waves through LDS
  double ns_gamma;
  int ns_iter, nxt;
};

template <int R, int KC>
__global__ __launch_bounds__(256) void k_iter_rgd(const AgentDev *__restrict__ agents, TeamDev *team, NestState *nest_all,
                                                  unsigned long long *bar, int *err, int first, int nb_eval,
                                                  double step, int num_robots, int restart_interval, int ahead) {
  const int hw = (int)blockIdx.x, G8 = (int)gridDim.x / 8;
  const int bx = (hw % 8) * G8 + hw / 8;  // XCD-aware block order, as in k_precond
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  ITER_STAMP(0);

  // ---- round trip 1: the team line (iteration counter, selected agents, schedule) and the hand-off epoch
  // (readfirstlane: the values are wave-uniform, and uniform indices keep the descriptor reads on the scalar path)
  const int iter_old = __builtin_amdgcn_readfirstlane(ld_once(&team->iter));
  const int cur = __builtin_amdgcn_readfirstlane(first ? ld_once(&team->cur_sel) : ld_once(&team->next_sel));
  const unsigned long long epoch = ld_once(&bar[BAR_EPOCH]) + 1ull;
  const int iter_new = iter_old + (first ? 0 : 1);
  const int num_agents = team->num_agents;
  // ---- round trip 2 (all independent of one another): descriptor of the agent, its Nesterov state, the agent of
  // iteration k+1, and -- workgroup b < num_agents -- the Nesterov state of agent b for the write-back
  const AgentDev &ag = agents[cur];
  const int N4 = ag.N4;
  const int nblk = precond_blocks(N4);
  if (bx >= nblk) return;  // padding workgroup: not counted by the hand-off
  const double Nr = (double)num_robots;

  constexpr int MREG = KC / 64;
  constexpr int PPB = 64 / R;
  // the polling wave's quarter of the slab is prefetched into LDS by the three other waves when it fits
  constexpr bool SLAB_LDS = false && (size_t)R * KC * 8 + (size_t)MREG * 1024 + 4096 <= 160 * 1024;
  __shared__ __attribute__((aligned(16))) double vs[R * KC];
  __shared__ __attribute__((aligned(16))) double zs[8 * R];
  __shared__ double red[32 * (8 * R + 1)];
  __shared__ double Ysh[2 * 4 * R];
  __shared__ double Esh[3][2 * 4 * R];
  __shared__ double EvY[PPB * 4 * R], EvW[PPB * 4 * R];
  __shared__ double2 mslab[SLAB_LDS ? MREG * 64 : 1];
  __shared__ IterBcast bc;

  const int cg = tid >> 5, kl = tid & 31;
  const int col0 = 8 * bx, col = col0 + cg;
  const bool cact = col < N4;
  const double *Mc = ag.M + (size_t)(cact ? col : 0) * N4;
  const int npose = min(2, ag.n - 2 * bx);
  double2 mreg[MREG];
  // A workgroup that owns an evaluation tile holds its slab prefetch back until the tile is done: the tile's dependent
  // round trips would otherwise queue behind 128 KB of this CU's own stream (9.5 us instead of 4 for the tile, and the
  // hand-off waits for the slowest tile); its slab then arrives under the hand-off and the vector staging.
  const bool eval_wg = hw < nb_eval;
  if (wave != 0) {
    if (eval_wg) __syncthreads();
#pragma unroll
    for (int m = 0; m < MREG; ++m) {
      const int k = 2 * kl + 64 * m;
      mreg[m] = (cact && k < N4) ? ld2_nt(Mc + k) : make_double2(0.0, 0.0);
    }
    if (SLAB_LDS) {
      // what lane l of the polling wave would load (columns col0, col0 + 1): copy m lands at mslab[m][l]; copies
      // outside the matrix stay away (the polling wave substitutes zeros by the same test)
      const int c0 = col0 + (lane >> 5), k0 = 2 * (lane & 31);
#pragma unroll
      for (int m = 0; m < MREG; ++m) {
        if (m % 3 != wave - 1) continue;
        const int k = k0 + 64 * m;
        if (c0 < N4 && k < N4)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ag.M + (size_t)c0 * N4 + k),
                                           (__attribute__((address_space(3))) void *)(mslab + m * 64), 16, 0, 2 /* nt */);
      }
    }
  }

  // epilogue operands of the two poses this workgroup owns (nothing before the hand-off writes them)
  const bool la_status = (ahead & 4) != 0, want_stats = (ahead & 8) != 0;
  double pre_x = 0, pre_v = 0, pre_y = 0, pre_p = 0;
  bool la_act = false, la_opt = false, la_restart = false;
  int la_agent = 0, la_pose = 0;
  double la_x[4 * R], la_y[4 * R], la_r2 = 0;
  double wb_gamma = 0;  // write-back: the Nesterov state of agent bx (workgroups bx < num_agents)
  int wb_iter = 0;
  if (wave == 0) {
    // run state as the previous launch left it, read by THIS wave only (it drains its loads before it arrives, so every
    // read of the old state precedes the write-back behind the hand-off); advanced in registers (what k_eval_stats'
    // bookkeeping workgroup does in the two-launch sequence)
    double ns_gamma = ld_once(&nest_all[cur].gamma);
    int ns_iter = ld_once(&nest_all[cur].iter);
    const int nxt0 = team->sched[(iter_new + 1) % team->sched_len];
    if (bx < num_agents) { wb_gamma = ld_once(&nest_all[bx].gamma); wb_iter = ld_once(&nest_all[bx].iter); }
    if (!first) {  // advance_agent()
      const bool rs = ((ns_iter + 2) % restart_interval) == 0;
      ns_gamma = rs ? 0.0 : (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * ns_gamma * ns_gamma)) / (2.0 * Nr);
      ns_iter += 1;
    }
    if (lane == 0) { bc.ns_gamma = ns_gamma; bc.ns_iter = ns_iter; bc.nxt = nxt0; }
    if (eval_wg) {
      eval_body<R, true>(agents, team, cur, B_X, B_EGRAD, B_GF, PART_C, 2, 1, hw, EvY, EvW);
      __syncthreads();  // releases the slab prefetch of the other three waves
    }
    ITER_STAMP(1);
    if (tid < npose * 4 * R) {
      pre_x = ag.buf[B_X][(size_t)col0 * R + tid];
      pre_v = ag.buf[B_V][(size_t)col0 * R + tid];
      pre_y = ag.buf[B_Y][(size_t)col0 * R + tid];
      if (want_stats) pre_p = ag.buf[B_XPREV][(size_t)col0 * R + tid];
    }
    if (ahead & 2) {
      // Nesterov step of iteration k+1 for this workgroup's share of the OTHER agents' poses (what k_precond's second
      // wave does behind the stream): one lane per pose; operands and results stay in registers across the hand-off
      const bool rs_now = ((ns_iter + 2) % restart_interval) == 0;
      la_restart = ((ns_iter + 3) % restart_interval) == 0;
      const double gam = rs_now ? 0.0 : (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * ns_gamma * ns_gamma)) / (2.0 * Nr);
      const double gam2 = (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * gam * gam)) / (2.0 * Nr);
      const double alpha2 = 1.0 / (gam2 * Nr);
      int pre[LOOKAHEAD_MAX_AGENTS + 1];
      const double *px[LOOKAHEAD_MAX_AGENTS], *pv[LOOKAHEAD_MAX_AGENTS];
#pragma unroll
      for (int k = 0; k <= LOOKAHEAD_MAX_AGENTS; ++k) pre[k] = team->pose_prefix[k];
#pragma unroll
      for (int k = 0; k < LOOKAHEAD_MAX_AGENTS; ++k) {
        px[k] = (k < num_agents) ? agents[k].buf[B_X] : nullptr;
        pv[k] = (k < num_agents) ? agents[k].buf[B_V] : nullptr;
      }
      const int total = pre[LOOKAHEAD_MAX_AGENTS] - ag.n;
      const int per = (total + nblk - 1) / nblk;  // <= 64, checked by the host
      const int q = bx * per + lane;
      if (lane < per && q < total) {
        int self_lo = 0;
#pragma unroll
        for (int k = 0; k < LOOKAHEAD_MAX_AGENTS; ++k) if (k == cur) self_lo = pre[k];
        const int g = q < self_lo ? q : q + ag.n;
        int a = 0, lo = 0;
        const double *xa = px[0], *va = pv[0];
#pragma unroll
        for (int k = 1; k < LOOKAHEAD_MAX_AGENTS; ++k)
          if (k < num_agents && g >= pre[k]) { a = k; lo = pre[k]; xa = px[k]; va = pv[k]; }
        la_act = true; la_agent = a; la_pose = g - lo;
        la_opt = nxt0 == a;
        const size_t o = (size_t)la_pose * 4 * R;
        double la_v[4 * R];
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) { la_x[i] = xa[o + i]; la_v[i] = va[o + i]; }
        if (!la_restart) {
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) la_y[i] = (1.0 - alpha2) * la_x[i] + alpha2 * la_v[i];
          polar_inplace<R>(la_y);
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) { const double d = la_y[i] - la_x[i]; la_r2 += d * d; }
        }
      }
    }
    ITER_STAMP(2);
    // the tile's stores have left this CU before the arrival is published
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ITER_STAMP(3);
    if (lane == 0) {
      const int g = hw & 7;
      const int size_g = min(G8, max(0, nblk - g * G8));
      const int ngroups = min(8, (nblk + G8 - 1) / G8);
      const unsigned long long old =
          __hip_atomic_fetch_add(&bar[g * BAR_LINE], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1ull == (unsigned long long)size_g * epoch) {
        const unsigned long long old2 =
            __hip_atomic_fetch_add(&bar[BAR_TOP], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old2 + 1ull == (unsigned long long)ngroups * epoch) {
#pragma unroll
          for (int q = 0; q < 8; ++q)
            __hip_atomic_store(&bar[BAR_GEN + q * BAR_LINE], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      ITER_STAMP(4);
      int spins = 0;
      while (__hip_atomic_load(&bar[BAR_GEN + g * BAR_LINE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > BAR_SPIN_LIMIT) { *err = 1; break; }
      }
      ITER_STAMP(5);
    }
  }
  __syncthreads();
  ITER_STAMP(6);
  const double ns_gamma = bc.ns_gamma;
  const int ns_iter = bc.ns_iter, nxt = bc.nxt;

  if (wave == 0 && lane == 0) {
    // every workgroup has read the old state and arrived: the advanced state goes back (what k_eval_stats' bookkeeping
    // workgroup and the step kernel's first thread write in the two-launch sequence), one agent per workgroup
    if (!first && bx < num_agents) {
      const bool rs = ((wb_iter + 2) % restart_interval) == 0;
      NestState w;
      w.gamma = rs ? 0.0 : (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * wb_gamma * wb_gamma)) / (2.0 * Nr);
      w.alpha = rs ? 0.0 : 1.0 / (w.gamma * Nr);
      w.iter = wb_iter + 1;
      w.pad = 0;
      nest_all[bx] = w;
    }
    if (bx == 0) {
      if (!first) team->iter = iter_new;
      team->cur_sel = cur;
      team->stats_sel = cur;
      team->next_sel = nxt;
      bar[BAR_EPOCH] = epoch;
    }
  }

  // ---- Nesterov scalars of iterations k and k+1 (k_precond, advance == 2)
  const bool restart_now = ((ns_iter + 2) % restart_interval) == 0;
  const bool restart_next = ((ns_iter + 3) % restart_interval) == 0;
  const double nest_gamma = restart_now ? 0.0 : (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * ns_gamma * ns_gamma)) / (2.0 * Nr);
  const double g2 = (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * nest_gamma * nest_gamma)) / (2.0 * Nr);
  const double ahead_alpha = 1.0 / (g2 * Nr);
  const bool ahead_opt = nxt == cur;

  if (wave == 0 && la_act) {
    // look-ahead of the other agents' poses: computed before the hand-off, stored behind it (the evaluation tiles of
    // this launch read the neighbours' Y)
    const AgentDev &oa = agents[la_agent];
    const size_t o = (size_t)la_pose * 4 * R;
    if (la_restart) {
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) {
        if (la_status) oa.buf[B_XPREV][o + i] = la_x[i];
        if (!la_opt) { oa.buf[B_Y][o + i] = la_x[i]; oa.buf[B_V][o + i] = la_x[i]; }
      }
      if (la_status && !la_opt) oa.part[PART_D + la_pose] = 0.0;
    } else {
      if (la_status && !la_opt) oa.part[PART_D + la_pose] = la_r2;
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) {
        if (la_status) oa.buf[B_XPREV][o + i] = la_x[i];
        oa.buf[B_Y][o + i] = la_y[i];
        oa.buf[B_X][o + i] = la_y[i];
      }
    }
  }

  ITER_STAMP(7);
  // ---- the gradient, written by other workgroups of this launch, into LDS in its native [k][a] layout
  const double *Vstage = ag.buf[B_GF];
  constexpr int NSTG = (KC * R / 2 + 255) / 256;
  if (wave == 0 && !SLAB_LDS) {
    // the polling wave's quarter of the slab: requested as soon as the hand-off is over, so that it lands under the
    // vector staging (32 KB per workgroup in front of the vector's loads)
#pragma unroll
    for (int m = 0; m < MREG; ++m) {
      const int k = 2 * kl + 64 * m;
      mreg[m] = (cact && k < N4) ? ld2_nt(Mc + k) : make_double2(0.0, 0.0);
    }
  }
  {
    // 16-byte non-temporal loads through registers, NBATCH batches (the M slab already occupies 128 registers).
    // nt loads bypass this CU's L1 (MI355X_MICROARCH.md, visibility table: "sc1 / sc0 sc1 / nt loads bypass L1 only"),
    // the producers stored write-through, and the L2s are kept coherent with remote write-through stores by the fabric's
    // snoop filter: the copy read here is the one the evaluation tiles published.  Measured
    // (profiles/experiments/fused_trace.py): registers 3.3 us in two batches; LDS-DMA copies 5.2 us; an agent-scope
    // acquire in front of plain loads costs 2 us on its own.
    constexpr int NBATCH = DPGO_ITER_STAGE_BATCHES;
    constexpr int HB = (NSTG + NBATCH - 1) / NBATCH;
#pragma unroll
    for (int h = 0; h < NBATCH; ++h) {
      double2 v[HB];
#pragma unroll
      for (int u = 0; u < HB; ++u) {
        const int tt = 2 * (tid + 256 * (h * HB + u));
        v[u] = (h * HB + u < NSTG && tt < N4 * R) ? ld2_nt(Vstage + tt) : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < HB; ++u) {
        const int tt = 2 * (tid + 256 * (h * HB + u));
        if (h * HB + u < NSTG && tt < KC * R) *reinterpret_cast<double2 *>(&vs[tt]) = v[u];
      }
    }
  }
  __syncthreads();
  ITER_STAMP(8);
  if (wave == 0 && SLAB_LDS) {
#pragma unroll
    for (int m = 0; m < MREG; ++m) {
      const int k = 2 * kl + 64 * m;
      mreg[m] = (cact && k < N4) ? mslab[m * 64 + lane] : make_double2(0.0, 0.0);
    }
  }
  double acc[R];
#pragma unroll
  for (int a = 0; a < R; ++a) acc[a] = 0;
#pragma unroll
  for (int m = 0; m < MREG; ++m) {
    const int k = 2 * kl + 64 * m;
    double w[2 * R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const double2 t2 = *reinterpret_cast<const double2 *>(&vs[k * R + 2 * j]);
      w[2 * j] = t2.x; w[2 * j + 1] = t2.y;
    }
#pragma unroll
    for (int a = 0; a < R; ++a) acc[a] += w[a] * mreg[m].x + w[R + a] * mreg[m].y;
  }

  // k-lane partial sums through LDS, one lane per (column, a) adds them in k-lane order (as k_precond does)
#pragma unroll
  for (int a = 0; a < R; ++a) red[kl * (8 * R + 1) + cg * R + a] = acc[a];
  if (tid < npose * 4 * R) {
    Ysh[tid] = pre_x;
    Esh[0][tid] = pre_v; Esh[1][tid] = pre_y; Esh[2][tid] = pre_p;
  }
  __syncthreads();
  if (tid < 8 * R) {
    double s = 0;
#pragma unroll
    for (int q = 0; q < 32; ++q) s += red[q * (8 * R + 1) + tid];
    zs[tid] = s;
  }
  ITER_STAMP(9);
  __syncthreads();
  ITER_STAMP(10);

  // ---- one lane per pose finishes the step in registers (k_precond PM_RGD tail, accel, advance == 2)
  double rel = 0;
  if (tid < npose) {
    const int lp = tid;
    const size_t o = (size_t)(2 * bx + lp) * 4 * R;
    double x[4 * R], z[4 * R];
#pragma unroll
    for (int i = 0; i < 4 * R; ++i) { x[i] = Ysh[lp * 4 * R + i]; z[i] = zs[lp * 4 * R + i]; }
    tangent_inplace<R>(x, z);
#pragma unroll
    for (int i = 0; i < 4 * R; ++i) x[i] -= step * z[i];
    qf_inplace<R>(x);
    if (want_stats) {
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) {
        ag.buf[B_X2][o + i] = x[i];
        const double d = x[i] - Esh[2][lp * 4 * R + i];
        rel += d * d;
      }
    }
    const bool reset = restart_now;
    double v[4 * R];
    if (reset) {
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) v[i] = x[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) v[i] = Esh[0][lp * 4 * R + i] + nest_gamma * (x[i] - Esh[1][lp * 4 * R + i]);
      polar_inplace<R>(v);
    }
    if (ahead & 1) {
      if (la_status) {
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) ag.buf[B_XPREV][o + i] = x[i];
      }
      if (restart_next) {
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) {
          ag.buf[B_X][o + i] = x[i];
          if (!ahead_opt) { ag.buf[B_Y][o + i] = x[i]; v[i] = x[i]; } else if (reset) ag.buf[B_Y][o + i] = x[i];
        }
        if (la_status && !ahead_opt) ag.part[PART_D + 2 * bx + lp] = 0.0;
      } else {
        double y[4 * R];
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) y[i] = (1.0 - ahead_alpha) * x[i] + ahead_alpha * v[i];
        polar_inplace<R>(y);
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) { ag.buf[B_Y][o + i] = y[i]; ag.buf[B_X][o + i] = y[i]; }
        if (la_status && !ahead_opt) {
          double rel2 = 0;
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) { const double d = y[i] - x[i]; rel2 += d * d; }
          ag.part[PART_D + 2 * bx + lp] = rel2;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) { ag.buf[B_X][o + i] = x[i]; if (reset) ag.buf[B_Y][o + i] = x[i]; }
    }
#pragma unroll
    for (int i = 0; i < 4 * R; ++i) ag.buf[B_V][o + i] = v[i];
  }
  if (tid < 64 && want_stats) {
    rel = wave_sum(rel);
    if (tid == 0) ag.part[PART_B + (size_t)bx * PART_STRIDE + 2] = rel;
  }
  ITER_STAMP(11);
}

// host side: whether the fused kernel can serve an agent of n poses evaluated in nb_eval tiles.  Every evaluation tile
// needs an active workgroup (hardware block h evaluates tile h and must not be a padding block), the agent must fit one
// chunk, and the whole grid must be resident at once (one workgroup per CU).
bool iter_fused_eligible(int r, int max_n, const int *agent_n, int num_agents, int num_cus) {
  if (4 * max_n > 2048) return false;  // (agents of this size always hold the dense inverse unless block-Jacobi was forced)
  const int grid = ((4 * max_n + 7) / 8 + 7) / 8 * 8, G8 = grid / 8;
  if (grid > num_cus) return false;
  const int ppb = 64 / r;
  for (int k = 0; k < num_agents; ++k) {
    const int n = agent_n[k], nblk = (4 * n + 7) / 8;
    const int tiles = (n + ppb - 1) / ppb;  // evaluation tiles that hold poses of this agent
    for (int h = 0; h < tiles; ++h) {
      if (h >= grid) return false;
      if ((h % 8) * G8 + h / 8 >= nblk) return false;  // hardware block h would be a padding block for this agent
    }
  }
  return true;
}

void launch_iter_rgd(const LaunchCtx &c, int max_n, NestState *nest_all, unsigned long long *bar, int *err, int first,
                     double step, int num_robots, int restart_interval, int ahead) {
  const int grid = (((4 * max_n + 7) / 8) + 7) / 8 * 8;
  const int nb_eval = spmm_grid(c.r, max_n);
  if (4 * max_n > 1024) {
    DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL((k_iter_rgd<R, 2048>), dim3(grid), dim3(256), 0, c.stream, c.agents, c.team, nest_all,
                                            bar, err, first, nb_eval, step, num_robots, restart_interval, ahead));
  } else {
    DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL((k_iter_rgd<R, 1024>), dim3(grid), dim3(256), 0, c.stream, c.agents, c.team, nest_all,
                                            bar, err, first, nb_eval, step, num_robots, restart_interval, ahead));
  }
}

}  // namespace dpgo
