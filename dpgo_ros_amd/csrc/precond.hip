// precond.hip -- the dense preconditioner apply z = P_X(v (Q + shift I)^-1) (PreConditioner) and the
// solver steps fused into it: tCG set-up and step (a4), the RGD step with the Nesterov V update and the look-ahead
// Nesterov step of the pipelined iteration (a4, a6).  The one kernel of the path that streams HBM.
#include "kernel_common.h"
#include "twolevel_dev.h"
#include <algorithm>

namespace dpgo {

// optional per-phase timestamps of hardware block 100, wave 0 / wave 1 (build with -DDPGO_PC_TRACE): written to the
// partial-sum scratch of the agent, region PART_E, words [4000 ..]
#ifndef DPGO_PC_MPRE
#define DPGO_PC_MPRE 4
#endif
#ifndef DPGO_PC_KCMID
#define DPGO_PC_KCMID 1280
#endif
#ifndef DPGO_PC_TRACE_BLOCK
#define DPGO_PC_TRACE_BLOCK 100
#endif
#ifdef DPGO_PC_TRACE
#define PC_TRACE_ON true
#define PC_STAMP(k) do { if (MODE == PM_RGD_ && (threadIdx.x & 63) == 0 && blockIdx.x == DPGO_PC_TRACE_BLOCK) gp(ag.part)[PART_E + 4000 * PART_STRIDE + ((threadIdx.x >> 6) * 16) + (k)] = (double)wall_clock64(); \
    if (MODE == PM_RGD_ && threadIdx.x < 128 && (threadIdx.x & 63) == 0 && ((k) == 0 || (k) == 7)) gp(ag.part)[PART_E + (4100 + 2 * (int)blockIdx.x + (int)(threadIdx.x >> 6)) * PART_STRIDE + ((k) ? 1 : 0)] = (double)wall_clock64(); } while (0)
#else
#define PC_TRACE_ON false
#define PC_STAMP(k) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// Dense preconditioner apply  z = P_X( v (Q + shift I)^-1 )  (a3 PreConditioner).
// Workgroup = 256 threads = 8 scalar columns (2 poses) x 32 k-lanes.  M (the only large operand,
// N4^2 doubles, streamed exactly once, non-temporal so it does not evict the small operands from
// L2) is fetched with 16-byte coalesced loads that are ALL issued before the first use: one memory
// round trip per 2048-row chunk.  The input vector is staged in LDS as SoA [a][k].  Modes:
//   PM_PLAIN    v = buf[vb]                        -> buf[zb]            partials [0]<z,v> [1]<v,v>
//   PM_TCG_INIT v = gf; r0 = gf; eta = 0; d0 = -z  (tCG set-up, RtrState ping-pong)
//   PM_TCG_STEP stages Hd only: r += alpha Hd, eta += alpha d, z += alpha P(Hd M)   (tCG body, part 2)
//   PM_RGD      v = gf; X <- Retr_X(-step z); [V <- proj(V + gamma (X - Y))]; partial [2] |X - XPrev|^2
//               (the whole RGD step + Nesterov V update of the two poses this workgroup owns)
//               advance: 1 = end-of-iteration bookkeeping here (3-launch iteration), 2 = pipelined iteration (see
//               k_eval_stats): publishes stats_sel / next_sel only, derives gamma and the restart flags of iterations
//               k and k+1 from the NestState, and a restart iteration is a plain step with V = Y = X.
//               ahead (pipelined only), bit 0: the first wave also takes the Nesterov step of iteration k+1 of its
//               two poses; bit 1: the second wave takes it for the workgroup's share of the other agents' poses;
//               bit 2: those steps also leave what a status query reads (XPrev, |Y' - X|^2 per pose); bit 3: this
//               step leaves its statistics (X2 snapshot, |X - XPrev|^2).  Mid-run launches carry 3, the last two
//               of a run 7 and 8.  bit 4 (lockstep ticks, advance = 0): this step leaves NO statistics (every tick but
//               the last of a graph).
// KC = rows of M (scalars of the input vector) handled per chunk: KC * R * 8 bytes of LDS and KC / 64
// 16-byte registers per lane.  One 2048-row chunk covers a 500-pose agent in a single round trip with one
// workgroup per CU; larger agents use 1024-row chunks so that 3 workgroups fit a CU and one workgroup's
// arithmetic overlaps the others' streams.

// TLC: the two-level product is compiled in (it costs the dense variants registers, i.e. occupancy: teams without a
// two-level agent run kernels without it)
// BAKED: the agent's descriptor arrives by value with the launch (pick_agent, kernel_common.h)
// LEAN: a mid-run step of the pipelined sequence (accelerated, advance = 2, ahead = 3: nothing a status query reads is
// left behind) -- the flags become compile-time constants and the tail loses its statistics paths
// the m-th 64-row chunk of a column of M a lane takes: agents of <= 512 poses (one 2048-row pass) follow the agent's
// chunk order -- private chunks first (dpgo_dev.h, fe_ord; uniform: scalar loads) --, which keeps this kernel's sums
// bitwise those of the one-launch iteration that forms the private part a launch early (step_fused.hip)
template <int KC>
__device__ __forceinline__ int chunk_of(const AgentDev &ag, int m) {
  if constexpr (KC == 2048) return (int)ag.fe_ord[m];
  else return m;
}

template <int R, int MODE, int KC, bool TLC, bool BAKED, bool LEAN = false>
__global__ __launch_bounds__(256) void k_precond(const AgentDev *__restrict__ agents, TeamDev *team, int sel, int xb, int vb,
                                                 int zb, int sp, int max_inner, double step, int accel,
                                                 int num_robots, int advance, int restart_interval, int ahead,
                                                 const NestState *nest_all, const AgentDev agv) {
  if constexpr (LEAN) { accel = 1; advance = 2; ahead = 3; }
  // XCD-aware block order: hardware workgroup h runs on XCD h % 8 (each with its own L2).  Logical block
  // (h % 8) * (grid / 8) + h / 8 gives every XCD one contiguous range of poses, so that the cache lines shared by
  // neighbouring poses (a pose is 4R doubles, not a multiple of a line) are written inside one L2 instead of
  // being split between two.  The grid is padded to a multiple of 8; padding blocks fall out at the nblk test.
  // Two-level agents (ag.tl, twolevel.h) keep the hardware order: their first workgroups are the producers of the
  // exchange and must be dispatched first.
  const int agent_index = BAKED ? sel : sel_cur(team, sel);
  const AgentDev &ag = pick_agent<BAKED>(agv, agents, agent_index);
  const bool is_tl = TLC && ag.tl.nwg > 0;
  // (two-level agents: the first nA workgroups of the launch are the producers of the exchange -- they own no columns and
  // no logical index; workgroup nA + k is logical block k)
  const int hb = (int)blockIdx.x;
  const bool producer = is_tl && hb < ag.tl.nA;
  const int bx = is_tl ? hb - ag.tl.nA : (hb % 8) * ((int)gridDim.x / 8) + hb / 8;
  PC_STAMP(0);
  if (MODE == PM_RGD_ && advance == 2 && bx == 0 && threadIdx.x == 0) {
    // pipelined iterations: nothing that a workgroup of THIS launch reads is written here (cur_sel, iter and the
    // NestStates move in the next k_eval_stats); the next launch finds its statistics agent and its own agent
    team->stats_sel = team->cur_sel;
    team->next_sel = team->sched[(team->iter + 1) % team->sched_len];
  }
  if (MODE == PM_RGD_ && advance == 1 && bx == 0 && threadIdx.x == 0) {
    // end-of-iteration bookkeeping of the whole team, folded here: no workgroup of this kernel reads
    // team->iter (they use cur_sel) or a NestState (gamma' comes from scal[6]), and the next kernel that
    // does (k_nest_pre of the following iteration) is ordered behind this launch
    for (int k = 0; k < team->num_agents; ++k) advance_agent(agents[k], accel, num_robots, restart_interval);
    team->iter += 1;
    team->stats_sel = team->cur_sel;
  }
  // LDS: the staged input vector of the dense stream (R * KC doubles) and the 64 partial-sum rows of the two-level
  // product (twolevel_dev.h) share one area -- a workgroup runs one or the other
  constexpr int MREG = KC / 64;
  constexpr int SMEM = (!TLC || R * KC > TL_RED_DOUBLES(R)) ? (R * KC > 0 ? R * KC : 1) : TL_RED_DOUBLES(R);
  __shared__ double vs[SMEM];
  __shared__ double zs[8 * R];
  __shared__ double red[32 * (8 * R + 1)];
  __shared__ double Ysh[2 * 4 * R];
  __shared__ double Esh[3][2 * 4 * R];  // PM_RGD: V, Yaux, XPrev of the two poses
  const int tid = threadIdx.x, lane = tid & 63;
  const int N4 = ag.N4;
  const int nblk = precond_nblk(ag);
  if (bx >= nblk) return;  // (producers: bx < 0)
  // the two poses this workgroup owns: consecutive ones, or the pair the two-level layout assigns
  TLWg tlw = {};
  int pj0 = 2 * bx, pj1 = (2 * bx + 1 < ag.n) ? 2 * bx + 1 : -1;
  if constexpr (TLC) {
    if (is_tl) { tlw = ag.tl.wg[hb]; pj0 = tlw.own[0]; pj1 = tlw.own[1]; }
  }

  // ---- scalar prologue (identical in every workgroup)
  double alpha = 0, tau = 0;
  int jpar = 0;
  bool boundary = false;
  RtrState S;
  if (MODE == PM_TCG_INIT_ || MODE == PM_TCG_STEP_) {
    S = ag.st[sp];
    // phase gating: the host enqueues [init, (hv, step) x J, retract, eval2, accept] patterns blindly;
    // a kernel whose phase is not due forwards the state and returns
    const bool idle = S.outer_done || (MODE == PM_TCG_STEP_ && !S.tcg_active) || (MODE == PM_TCG_INIT_ && !S.need_init);
    if (idle) {
      if (bx == 0 && tid == 0) ag.st[sp ^ 1] = S;
      return;
    }
    if (MODE == PM_TCG_STEP_) {
      const double d_Hd = sum_partials(ag.part + PART_A, spmm_blocks<R>(ag.n), PART_STRIDE, lane);
      alpha = S.z_r / d_Hd;
      const double e_Pe_new = S.e_Pe + 2.0 * alpha * S.e_Pd + alpha * alpha * S.d_Pd;
      jpar = S.tcg_j & 1;
      if (d_Hd <= 0 || e_Pe_new >= S.Delta * S.Delta) {
        boundary = true;
        tau = (-S.e_Pd + sqrt(S.e_Pd * S.e_Pd + S.d_Pd * (S.Delta * S.Delta - S.e_Pe))) / S.d_Pd;
        if (bx == 0 && tid == 0) {
          RtrState T = S;
          T.tcg_active = 0;
          T.tcg_status = (d_Hd <= 0) ? 1 : 2;
          ag.st[sp ^ 1] = T;
        }
      } else if (bx == 0 && tid == 0) {
        RtrState T = S;
        T.e_Pe = e_Pe_new;
        T.alpha = alpha;
        T.tcg_j = S.tcg_j + 1;
        T.pc_count = S.pc_count + 1;
        ag.st[sp ^ 1] = T;
      }
    } else if (bx == 0 && tid == 0) {
      RtrState T = S;
      T.tcg_active = 1; T.tcg_j = 0; T.tcg_status = 0; T.need_init = 0;
      T.e_Pd = 0; T.e_Pe = 0; T.alpha = 0;
      T.pc_count = S.pc_count + 1;
      T.outer_count = S.outer_count + 1;
      ag.st[sp ^ 1] = T;
    }
  }

  const double *Vin = (MODE == PM_PLAIN_) ? ag.buf[vb]
                      : ((MODE == PM_TCG_INIT_ || MODE == PM_RGD_) ? ag.buf[B_GF] : ag.buf[jpar ? B_R1 : B_R0]);
  const double *Hd = ag.buf[B_HD];
  // tCG step: the preconditioned residual obeys the same recurrence as the residual,
  //   r+ = r + alpha Hd   =>   z+ = P(r+ M) = z + alpha P(Hd M)      (P and M are linear),
  // so only ONE vector (Hd) has to be pulled through every workgroup's LDS; r and z are updated in place by
  // their owners.  (The oracle recomputes z from r+ directly; the two differ by round-off only.)
  const double *Vstage = (MODE == PM_TCG_STEP_) ? Hd : Vin;
  // two-level agents: the first pass of the product is requested here, in front of everything the epilogue will need
  TLPre<R> tlpre;
  if constexpr (TLC) {
    if (is_tl) tl_issue<R>(ag.tl, tlw, hb, Vstage, tid, tlpre);
  }
  const int npose = (pj1 >= 0) ? 2 : 1;
  // element `tid` (< npose * 4R) of the own poses in an r x 4n array
  const size_t own_off = (size_t)((tid >= 4 * R) ? max(pj1, 0) : pj0) * 4 * R + (size_t)(tid % (4 * R));

  if (MODE == PM_TCG_STEP_) {
    // eta += (alpha | tau) * delta on the two poses owned by this workgroup
    const double *D = ag.buf[jpar ? B_D1 : B_D0];
    double *E = ag.buf[B_ETA];
    const double stepc = boundary ? tau : alpha;
    if (tid < npose * 4 * R && !producer) E[own_off] += stepc * D[own_off];
    if (boundary) return;
  }

  // epilogue operands of the two poses this workgroup owns: requested now, consumed after the M stream
  // ahead bits: 1 look-ahead of this workgroup's poses, 2 look-ahead of the other agents' poses, 4 the look-ahead also
  // leaves what a status query needs (XPrev, |Y' - X|^2 per pose), 8 this step leaves its statistics (X2 snapshot,
  // |X - XPrev|^2).  Only the last two iterations of a run set 4 / 8: nothing reads those values in between.
  const bool la_status = (ahead & 4) != 0, want_stats = (MODE != PM_RGD_) || (advance != 2 && !(ahead & 16)) || (ahead & 8) != 0;
  double pre_x = 0, pre_v = 0, pre_y = 0, pre_p = 0;
  double nest_gamma = 0;
  if (tid < npose * 4 * R) {
    pre_x = ag.buf[xb][own_off];
    if (MODE == PM_RGD_) {
      pre_v = gp(ag.buf[B_V])[own_off];
      pre_y = gp(ag.buf[B_Y])[own_off];
      if (want_stats) pre_p = gp(ag.buf[B_XPREV])[own_off];
    }
  }
  double ahead_alpha = 0;
  bool ahead_opt = false, restart_now = false, restart_next = false;
  if (MODE == PM_RGD_ && accel) {
    if (advance == 2) {
      // the NestState describes iteration k-1 (it is advanced by the next k_eval_stats): gamma of this iteration,
      // and gamma / alpha / selected agent of iteration k+1 for the look-ahead Nesterov step of the epilogue
      const NestState ns = nest_all ? nest_all[agent_index] : *ag.nest;  // (one round trip less than through the descriptor)
      const double Nr = (double)num_robots;
      // restart iterations are part of the uniform sequence: iteration k restarts when (iter + 2) % interval == 0
      // (the test of k_nest_pre / advance_agent); its step is a plain RGD step from X with V = Y = X afterwards
      // and gamma = 0 for what follows.  (The reference also runs -- and discards -- an accelerated solve first; it
      // has no observable effect and is skipped here.)
      restart_now = ((ns.iter + 2) % restart_interval) == 0;
      restart_next = ((ns.iter + 3) % restart_interval) == 0;
      nest_gamma = restart_now ? 0.0 : (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * ns.gamma * ns.gamma)) / (2.0 * Nr);
      const double g2 = (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * nest_gamma * nest_gamma)) / (2.0 * Nr);
      ahead_alpha = 1.0 / (g2 * Nr);
      ahead_opt = team->sched[(team->iter + 1) % team->sched_len] == sel_cur(team, sel);
    } else {
      nest_gamma = ag.scal[6];
    }
  }
  // Look-ahead, other agents (pipelined iterations): while the first wave finishes the step of this workgroup's
  // two poses, the second wave takes the Nesterov step of iteration k+1 (what k_nest_pre would do next) for this
  // workgroup's share of the poses of every OTHER agent -- one lane per pose, straight from / to global memory:
  // XPrev = X; Y = proj((1 - alpha') X + alpha' V); X = Y; and for the agents that do not optimize at k+1:
  // V = proj(V), |Y - X|^2 per pose into PART_D.  Disjoint data: this launch reads nothing else of those agents.
  // The operands are requested next to the vector stage so that they arrive under the stream.  (The NestStates of
  // all agents advance in lockstep, so alpha' is the one computed above.)
  bool la_act = false, la_opt = false;
  int la_agent = 0, la_pose = 0;
  double la_x[4 * R], la_v[4 * R];
  if (is_tl) {
    // two-level operator: the product of twolevel_dev.h (one exchange inside the launch), then the common epilogues
    if constexpr (TLC) {
      if (!tl_apply<R, PC_TRACE_ON>(ag.tl, tlw, hb, Vstage, tlpre, vs, zs, tid,
                                    (MODE == PM_RGD_ && blockIdx.x == DPGO_PC_TRACE_BLOCK) ? ag.part + PART_E + 4000 * PART_STRIDE : nullptr))
        return;  // a producer: its entry of u is published, it owns no columns
    }
    if (tid < npose * 4 * R) {
      Ysh[tid] = pre_x;
      if (MODE == PM_RGD_) { Esh[0][tid] = pre_v; Esh[1][tid] = pre_y; Esh[2][tid] = pre_p; }
    }
    // (zs and these rows are used by wave 0 / the first lanes of wave 1 only; wave 0 wrote Ysh / Esh itself, and wave 1
    // reads none of them)
    if (tid >= 128) return;
  } else {
  const int col0 = 8 * bx;
  const int cg = tid >> 5, kl = tid & 31;
  const int col = col0 + cg;
  const bool cact = col < N4;
  const double *Mc = ag.M + (size_t)(cact ? col : 0) * N4;
  double acc[R];
#pragma unroll
  for (int a = 0; a < R; ++a) acc[a] = 0;

  // Per chunk of KC rows: (1) the input vector is copied into LDS in its native [k][a] layout with one
  // batch of 16-byte loads, (2) barrier, (3) the whole M slab of this workgroup is requested (32 x 16 B per
  // lane, non-temporal), (4) the FMA loop drains the slab in issue order, so arithmetic overlaps the
  // stream.  A lane reads the 2R contiguous doubles v[k][:], v[k+1][:] as R ds_read_b128 (16R-byte lane
  // stride: conflict-free for R = 3, 5).  Measured (profiles/experiments/pc_bench.hip): ingest per CU, not HBM, is
  // the limit -- every workgroup has to pull the full 8*R*N4-byte vector through L2 next to its slab.
  constexpr int NSTG = (KC * R / 2 + 255) / 256;  // 16-byte pairs per lane per chunk
  PC_STAMP(1);
  if (!ag.M) {
    // block-Jacobi agent (the declared fallback where the dense inverse does not fit, include/dpgo_hip.h): column
    // c of pose p is  v_p (Q_pp + shift I)^-1[:, c], a 4 x 4 block per pose -- the first k-lane of each column forms it
    // straight from global memory, the others contribute zeros to the reduction below
    if (kl == 0 && cact) {
      const int p = col >> 2, c = col & 3;
      const double *B = ag.Dinv + (size_t)16 * p + 4 * c;
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) {
        const double b = B[cp];
#pragma unroll
        for (int a = 0; a < R; ++a) acc[a] += Vstage[((size_t)4 * p + cp) * R + a] * b;
      }
    }
  }
  // MPRE of the MREG slab loads of a chunk are requested BEFORE the vector is staged: a handful of requests per lane
  // (they do not hold the wave at the issue queue as the full 32 would) that take the ~2 us first-byte latency of the
  // stream -- row activation, translation -- under the 1.6 us of the staging instead of behind it
  constexpr int MPRE = DPGO_PC_MPRE;
  if constexpr (KC > 0)
  for (int k0 = 0; ag.M && k0 < N4; k0 += KC) {
    const int kn = min(KC, N4 - k0);
    if (k0 > 0) __syncthreads();
    double2 mreg[MREG];
#pragma unroll
    for (int m = 0; m < MPRE; ++m) {
      const int k = 2 * kl + 64 * chunk_of<KC>(ag, m);
      // (an unconditional load from a clamped address: a load under a lane-varying condition is a branch of its own, and
      // behind 32 of them the compiler waits for the whole slab before the first multiply -- round 5, read off the ISA.
      // Rows beyond the chunk meet zeros of the staged vector; columns beyond the last pose feed sums nobody reads.)
      mreg[m] = ld2g_nt(Mc + min(k0 + k, N4 - 2));
    }
    {
      double2 v[NSTG];
#pragma unroll
      for (int u = 0; u < NSTG; ++u) {
        const int tt = 2 * (tid + 256 * u);  // kn * R is even
        const double2 t2 = ld2g(Vstage + (size_t)k0 * R + min(tt, kn * R - 2));
        v[u] = (tt < kn * R) ? t2 : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < NSTG; ++u) {
        const int tt = 2 * (tid + 256 * u);
        if (tt < KC * R) *reinterpret_cast<double2 *>(&vs[tt]) = v[u];
      }
    }
    __syncthreads();
    PC_STAMP(2);
#pragma unroll
    for (int m = MPRE; m < MREG; ++m) {
      const int k = 2 * kl + 64 * chunk_of<KC>(ag, m);
      mreg[m] = ld2g_nt(Mc + min(k0 + k, N4 - 2));
    }
    PC_STAMP(3);
#pragma unroll
    for (int m = 0; m < MREG; ++m) {
      const int k = 2 * kl + 64 * chunk_of<KC>(ag, m);
      double w[2 * R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const double2 t2 = *reinterpret_cast<const double2 *>(&vs[k * R + 2 * j]);
        w[2 * j] = t2.x; w[2 * j + 1] = t2.y;
      }
#pragma unroll
      for (int a = 0; a < R; ++a) acc[a] = __builtin_fma(w[R + a], mreg[m].y, __builtin_fma(w[a], mreg[m].x, acc[a]));  // (two FMAs: the one-expression form is mul, fma, add)
    }
  }
  PC_STAMP(4);
  // the 32 k-lane partial sums of every column go through LDS ([k-lane][column * R + a], rows padded to an odd number
  // of doubles: conflict-free both ways) and ONE lane per (column, a) adds them in k-lane order: 0.5 us where five
  // 5-stage ds_bpermute butterflies took 2.2 (profiles/experiments/step_trace.py)
#pragma unroll
  for (int a = 0; a < R; ++a) red[kl * (8 * R + 1) + cg * R + a] = acc[a];
  if (tid < npose * 4 * R) {
    Ysh[tid] = pre_x;
    if (MODE == PM_RGD_) { Esh[0][tid] = pre_v; Esh[1][tid] = pre_y; Esh[2][tid] = pre_p; }
  }
  __syncthreads();
  // from here on the waves go separate ways and need no workgroup barrier: the first wave adds the partial sums and
  // finishes the step of the two poses (LDS operations of one wave execute in order), the second wave takes the
  // look-ahead step of the other agents' poses (its operand addresses come from a chain of scalar loads that used to sit
  // on every wave's path: 1.5 us), the other two are done
  if (tid >= 128) return;
  if (tid < 8 * R) {
    double s = 0;
#pragma unroll
    for (int q = 0; q < 32; ++q) s += red[q * (8 * R + 1) + tid];
    zs[tid] = s;
  }
  }  // (dense / block-Jacobi)
  PC_STAMP(5);
  if (tid < 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  PC_STAMP(6);
  if (MODE == PM_RGD_ && (ahead & 2) && tid >= 64 && tid < 128) {
    // look-ahead operands of the second wave, requested once its share of the stream is consumed: they arrive while
    // the partial sums are reduced and the first wave starts its tail, and they do not occupy registers during the
    // stream (the tail is register-bound).
    // Every address comes from wave-uniform (scalar) loads -- a per-lane fetch of agents[a].buf would queue
    // behind the vector stream and stall the wave.
    const int self = sel_cur(team, sel);
    int pre[LOOKAHEAD_MAX_AGENTS + 1];
    const double *px[LOOKAHEAD_MAX_AGENTS], *pv[LOOKAHEAD_MAX_AGENTS];
#pragma unroll
    for (int k = 0; k <= LOOKAHEAD_MAX_AGENTS; ++k) pre[k] = team->pose_prefix[k];
    const int na = team->num_agents;
#pragma unroll
    for (int k = 0; k < LOOKAHEAD_MAX_AGENTS; ++k) {
      px[k] = (k < na) ? agents[k].buf[B_X] : nullptr;
      pv[k] = (k < na) ? agents[k].buf[B_V] : nullptr;
    }
    const int total = pre[LOOKAHEAD_MAX_AGENTS] - ag.n;
    const int per = (total + nblk - 1) / nblk;  // <= 64, checked by the host
    const int l1 = tid - 64;
    const int q = bx * per + l1;   // index among the poses of the other agents
    if (l1 >= 0 && l1 < per && q < total) {
      int self_lo = 0;
#pragma unroll
      for (int k = 0; k < LOOKAHEAD_MAX_AGENTS; ++k) if (k == self) self_lo = pre[k];
      const int g = q < self_lo ? q : q + ag.n;  // index among all poses of the team
      int a = 0, lo = 0;
      const double *xa = px[0], *va = pv[0];
#pragma unroll
      for (int k = 1; k < LOOKAHEAD_MAX_AGENTS; ++k)
        if (k < na && g >= pre[k]) { a = k; lo = pre[k]; xa = px[k]; va = pv[k]; }
      la_act = true; la_agent = a; la_pose = g - lo;
      la_opt = team->sched[(team->iter + 1) % team->sched_len] == a;
      const size_t o = (size_t)la_pose * 4 * R;
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) { la_x[i] = gp(xa)[o + i]; la_v[i] = gp(va)[o + i]; }
    }
  }

  if (MODE == PM_RGD_ && (ahead & 2) && tid >= 64 && tid < 128) {
    // look-ahead of the other agents' poses on the second wave (operands prefetched in the prologue)
    if (la_act) {
      const AgentDev &oa = agents[la_agent];
      const size_t o = (size_t)la_pose * 4 * R;
      if (restart_next) {
        // iteration k+1 restarts: XPrev = X, and V = Y = X for the agents that do not optimize (X does not move)
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) {
          if (la_status) gp(oa.buf[B_XPREV])[o + i] = la_x[i];
          if (!la_opt) { gp(oa.buf[B_Y])[o + i] = la_x[i]; gp(oa.buf[B_V])[o + i] = la_x[i]; }
        }
        if (la_status && !la_opt) gp(oa.part)[PART_D + la_pose] = 0.0;
      } else {
        // V of an agent that does not optimize is re-projected by the reference (V = proj(V)); V left its last update
        // as a polar factor, so the projection is the identity up to round-off and V is neither read nor written here
        double y[4 * R];
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) y[i] = (1.0 - ahead_alpha) * la_x[i] + ahead_alpha * la_v[i];
        polar_inplace<R>(y);
        if (la_status && !la_opt) {
          double r2 = 0;
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) { const double d = y[i] - la_x[i]; r2 += d * d; }
          gp(oa.part)[PART_D + la_pose] = r2;
        }
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) {
          if (la_status) gp(oa.buf[B_XPREV])[o + i] = la_x[i];
          gp(oa.buf[B_Y])[o + i] = y[i];
          gp(oa.buf[B_X])[o + i] = y[i];
        }
      }
    }
    PC_STAMP(7);
    return;
  }
  if (MODE == PM_RGD_) {
    // one lane per pose finishes the step in registers: z = P(zs), X = qf(X - step z), V update
    double rel = 0;
    if (tid < npose) {
      const int lp = tid;
      const size_t o = (size_t)(lp ? pj1 : pj0) * 4 * R;
      double x[4 * R], z[4 * R];
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) { x[i] = Ysh[lp * 4 * R + i]; z[i] = zs[lp * 4 * R + i]; }
      tangent_inplace<R>(x, z);
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) x[i] -= step * z[i];
      PC_STAMP(8);
      qf_inplace<R>(x);
      PC_STAMP(9);
      if (want_stats) {
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) {
          gp(ag.buf[B_X2])[o + i] = x[i];  // snapshot for the final-statistics evaluation of this iteration
          const double d = x[i] - Esh[2][lp * 4 * R + i];
          rel += d * d;
        }
      }
      // results are stored as soon as they exist: the per-pose arrays of this tail do not fit the register file
      // together (they spill to AGPRs, which costs more than the stores)
      const bool reset = accel && advance == 2 && restart_now;  // restart iteration: V = Y = X (k_nest_reset)
      double v[4 * R];
      if (accel) {
        if (reset) {
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) v[i] = x[i];
        } else {
          const double gamma = nest_gamma;
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) v[i] = Esh[0][lp * 4 * R + i] + gamma * (x[i] - Esh[1][lp * 4 * R + i]);
          polar_inplace<R>(v);
        }
      }
      PC_STAMP(10);
      if (accel && (ahead & 1)) {
        // Nesterov step of iteration k+1 for this pose (what k_nest_pre would do next): XPrev = X, then
        //   k+1 regular:  Y = proj((1 - alpha') X + alpha' V), X = Y (V = proj(V) is the identity: V was just projected)
        //   k+1 restarts: V = Y = X unless this agent is selected again (X does not move)
        if (la_status) {
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) gp(ag.buf[B_XPREV])[o + i] = x[i];
        }
        if (restart_next) {
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) {
            gp(ag.buf[B_X])[o + i] = x[i];
            if (!ahead_opt) { gp(ag.buf[B_Y])[o + i] = x[i]; v[i] = x[i]; } else if (reset) gp(ag.buf[B_Y])[o + i] = x[i];
          }
          if (la_status && !ahead_opt) gp(ag.part)[PART_D + (lp ? pj1 : pj0)] = 0.0;
        } else {
          double y[4 * R];
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) y[i] = (1.0 - ahead_alpha) * x[i] + ahead_alpha * v[i];
          polar_inplace<R>(y);
          PC_STAMP(11);
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) { gp(ag.buf[B_Y])[o + i] = y[i]; gp(ag.buf[B_X])[o + i] = y[i]; }
          if (la_status && !ahead_opt) {
            double rel2 = 0;
#pragma unroll
            for (int i = 0; i < 4 * R; ++i) { const double d = y[i] - x[i]; rel2 += d * d; }
            gp(ag.part)[PART_D + (lp ? pj1 : pj0)] = rel2;  // look-ahead steps leave |Y' - X|^2 per pose
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) { gp(ag.buf[B_X])[o + i] = x[i]; if (reset) gp(ag.buf[B_Y])[o + i] = x[i]; }
      }
      if (accel) {
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) gp(ag.buf[B_V])[o + i] = v[i];
      }
    }
    if (tid < 64 && want_stats) {  // (mid-run launches leave no statistics: nothing reads them)
      rel = wave_sum(rel);
      if (tid == 0) gp(ag.part)[PART_B + (size_t)bx * PART_STRIDE + 2] = rel;
    }
    PC_STAMP(7);
    return;
  }

  // ---- epilogue: tangent projection of the two poses, dots, mode-specific stores
  double zr = 0, rr = 0;
  if (tid < npose * R) {
    const int lp = tid / R, a = tid - lp * R;
    const size_t o = (size_t)(lp ? pj1 : pj0) * 4 * R;
    double z[4];
    tangent_row<R>(Ysh + lp * 4 * R, zs + lp * 4 * R, a, z);
    z[3] = zs[lp * 4 * R + 3 * R + a];
    double *Z = ag.buf[(MODE == PM_PLAIN_) ? zb : B_Z];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double v = Vin[o + c * R + a];
      if (MODE == PM_TCG_STEP_) {
        v += alpha * Hd[o + c * R + a];
        ag.buf[jpar ? B_R0 : B_R1][o + c * R + a] = v;  // r_new into the other half
        z[c] = Z[o + c * R + a] + alpha * z[c];         // z_new = z_old + alpha P(Hd M)
      }
      if (MODE == PM_TCG_INIT_) {
        gp(ag.buf[B_R0])[o + c * R + a] = v;
        gp(ag.buf[B_ETA])[o + c * R + a] = 0.0;
        gp(ag.buf[B_D0])[o + c * R + a] = -z[c];
      }
      Z[o + c * R + a] = z[c];
      zr += z[c] * v;
      rr += v * v;
    }
  }
  if (tid < 64) {
    zr = wave_sum(zr);
    rr = wave_sum(rr);
    if (tid == 0) {
      double *P = ag.part + PART_B + (size_t)bx * PART_STRIDE;
      P[0] = zr; P[1] = rr;
    }
  }
}

void launch_precond(const LaunchCtx &c, int sel, int max_n, int mode, int xb, int vb, int zb, int sp, int max_inner,
                    double step, int accel, int num_robots, int advance, int restart_interval, int ahead) {
  // multiple of 8 (see the XCD-aware block order in k_precond); two-level agents run a few workgroups more than pose
  // pairs (every part of their ownership order is padded to an even number of slots)
  const int grid = (std::max((4 * max_n + 7) / 8, c.tl_max_wg) + 7) / 8 * 8;
  // What this launch may meet: a host-selected agent is known; a device-selected one (schedule, colour class) may be
  // any agent of the team.  dn: largest agent that streams a DENSE inverse (0: none); tl: a two-level agent is possible.
  int dn = std::min(max_n, c.dense_max_n);
  bool tl = c.any_two_level;
  if (sel >= 0 && c.host_precond) {
    tl = c.host_precond[sel] == 3;
    if (c.host_precond[sel] != 1) dn = 0;
  }
  // graphs with the schedule baked in pass the step kernel's agent descriptor by value (PM_RGD only)
  const bool baked = mode == PM_RGD_ && sel >= 0 && c.host_agents && c.bake_desc && c.ny == 1;
  AgentDev none{};
  const AgentDev &agv = baked ? c.host_agents[sel] : none;
#define PC_LAUNCH(M, KCV, TLV)                                                                                      \
    if (baked && M == PM_RGD_ && accel == 1 && advance == 2 && ahead == 3) {                                        \
      DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL((k_precond<R, PM_RGD_, KCV, TLV, true, true>), dim3(grid, c.ny), dim3(256), 0, c.stream, \
                                              c.agents, c.team, sel, xb, vb, zb, sp, max_inner, step, accel, num_robots,   \
                                              advance, restart_interval, ahead, c.nest_all, agv));                         \
    } else if (baked && M == PM_RGD_) {                                                                             \
      DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL((k_precond<R, PM_RGD_, KCV, TLV, true>), dim3(grid, c.ny), dim3(256), 0, c.stream, \
                                              c.agents, c.team, sel, xb, vb, zb, sp, max_inner, step, accel, num_robots,   \
                                              advance, restart_interval, ahead, c.nest_all, agv));                         \
    } else {                                                                                                        \
      DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL((k_precond<R, M, KCV, TLV, false>), dim3(grid, c.ny), dim3(256), 0, c.stream, \
                                              c.agents, c.team, sel, xb, vb, zb, sp, max_inner, step, accel, num_robots,   \
                                              advance, restart_interval, ahead, c.nest_all, agv));                         \
    }
  // chunk size by agent size: one 2048-row chunk for agents of 257..512 poses (one round trip, one workgroup per CU);
  // DPGO_PC_KCMID-row chunks for 513..(DPGO_PC_KCMID / 2) poses (two round trips instead of three, still three
  // workgroups per CU); 1024-row chunks otherwise; no dense stream at all (KC = 0) where no dense agent can be met
#define PC_CALL2(M, TLV)                                                                                            \
  if (dn == 0) { PC_LAUNCH(M, 0, TLV); }                                                                               \
  else if (4 * dn > 1024 && 4 * dn <= 2048) { PC_LAUNCH(M, 2048, TLV); }                                               \
  else if (DPGO_PC_KCMID > 0 && 4 * dn > 2048 && 4 * dn <= 2 * DPGO_PC_KCMID) { PC_LAUNCH(M, (DPGO_PC_KCMID > 0 ? DPGO_PC_KCMID : 1024), TLV); } \
  else { PC_LAUNCH(M, 1024, TLV); }
#define PC_CALL(M) if (tl) { PC_CALL2(M, true) } else { PC_CALL2(M, false) }
  if (mode == PM_PLAIN_) { PC_CALL(PM_PLAIN_); }
  else if (mode == PM_TCG_INIT_) { PC_CALL(PM_TCG_INIT_); }
  else if (mode == PM_TCG_STEP_) { PC_CALL(PM_TCG_STEP_); }
  else { PC_CALL(PM_RGD_); }
#undef PC_CALL
#undef PC_CALL2
#undef PC_LAUNCH
}

}  // namespace dpgo
