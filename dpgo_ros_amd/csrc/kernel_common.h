// kernel_common.h -- device-side helpers shared by the hand-written HIP kernels (gfx950 / CDNA4, wave64) of the
// RBCD hot path: spmm.hip (block-sparse evaluations), precond.hip (dense preconditioner apply and the solver steps
// fused into it), pose_ops.hip (per-pose manifold operations, scalar state machines, exchange and cost kernels).
//
// SURVEY 8a rows served here (call sites in /root/reference, bodies external):
//   a2  G assembly from neighbour public poses           (src/PGOAgentROS.cpp:1276,1278 feed it)
//   a3  QuadraticProblem f / EucGrad / RieGrad / Hess-vec / PreConditioner  (:169-172)
//   a4  RTR (Steihaug tCG) state machine + RGD step      (src/PGOAgentROSNode.cpp:85,90,96-100)
//   a5  tangent projection / QF retraction / polar projection (:1420-1422)
//   a6  Nesterov gamma/alpha/Y/V sequences + restart     (src/PGOAgentROSNode.cpp:126-130)
//   a7  pack/unpack of public-pose slabs                 (:662-690, :1255-1284)
//   a8  per-edge residuals                               (:1049)
//
// Layout: X is r x 4n column-major (pose = 4r contiguous doubles).  Q is stored twice: block-CSR
// (row j lists (i, Q_ij), (XQ)_j = sum_i X_i Q_ij) for assembly/read-back, and slot-major ELL
// (+ CSR tail for long rows) for the SpMM kernels, so that the column indices and the 4x4 blocks of
// a row are fetched with loads that do not depend on each other (the operands are L2/MALL resident;
// what bounds these kernels is the number of dependent round trips, not bytes).  One lane owns one
// (pose, row a) pair: a 64-wide wave covers floor(64/R) poses.
// Scalars of the inner solve (dots, alpha, beta, rho, radius) never visit the host: every
// workgroup re-derives them from the same per-block partial sums in the same order, and
// workgroup 0 publishes the next state into the other half of a ping-pong pair.
#pragma once
#include "device_math.h"
#include "dpgo_dev.h"
#include "kernels.h"

namespace dpgo {

// agent selection.  First kernel of an iteration: from the device-side schedule (and it publishes
// team->cur_sel); every later kernel: team->cur_sel, so that the last kernel may advance team->iter.
__device__ __forceinline__ int sel_sched(const TeamDev *team, int sel) {
  return sel >= 0 ? sel : team->sched[team->iter % team->sched_len];
}

__device__ __forceinline__ int sel_cur(const TeamDev *team, int sel) {
  if (sel >= 0) return sel;
  if (sel == -5) return team->stats_sel;
  if (sel == -6) return team->next_sel;
  if (sel > SEL_GROUP0) return team->cur_sel;
  if (sel == SEL_ALL) return (int)blockIdx.y;
  return team->group_members[team->group_ptr[SEL_GROUP0 - sel] + blockIdx.y];  // colour-parallel update
}

// Graphs that bake the agent of every iteration into its launches (dpgo_team_run, schedule period <= 8) also pass that
// agent's DESCRIPTOR by value: its fields then come from the kernel-argument segment with the launch instead of from
// agents[i] behind the argument pointer -- one dependent (scalar) round trip less at the head of both kernels of an
// iteration.  BAKED is a template flag: the two sources live in different address spaces.
template <bool BAKED>
__device__ __forceinline__ const AgentDev &pick_agent(const AgentDev &by_value, const AgentDev *__restrict__ agents, int index) {
  if constexpr (BAKED) return by_value;
  else return agents[index];
}

template <int R>
__device__ __forceinline__ int spmm_blocks(int n) { return (n + (64 / R) - 1) / (64 / R); }

__device__ __forceinline__ int precond_blocks(int N4) { return (N4 + 7) / 8; }
// workgroups of a preconditioner-type launch that own poses of this agent (and leave partials in PART_B)
__device__ __forceinline__ int precond_nblk(const AgentDev &ag) { return ag.tl.nwg > 0 ? ag.tl.nwg - ag.tl.nA : (ag.N4 + 7) / 8; }

// workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope release / acquire over ALL address
// spaces, and gfx9 counts loads and stores in one counter (vmcnt): every global load still in flight is waited for in
// front of the barrier -- a kernel that keeps a stream of HBM requests outstanding across its barriers must not use it.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Loads from GLOBAL memory, said so.  A pointer that was itself read from memory (a field of a descriptor in the agents
// array) is a generic pointer to the compiler and its loads are flat loads: they may return out of order, so a wait for
// one of them is `vmcnt(0) lgkmcnt(0)` -- a wait for every load and every LDS operation in flight.  The row products,
// the linear term and the statistics kernels carried 130 .. 300 of them each until round 5.  gp(p)[i] is p[i] as a
// global load; ld2 / ld2_nt are 16-byte global loads.  (Never for LDS.)
__device__ __forceinline__ double2 ld2(const double *p) {
  const v2d_t v = *(const __attribute__((address_space(1))) v2d_t *)p;
  return make_double2(v.x, v.y);
}

#ifndef DPGO_M_NT
#define DPGO_M_NT 1
#endif

__device__ __forceinline__ double2 ld2_nt(const double *p) {
#if DPGO_M_NT
  const v2d_t v = __builtin_nontemporal_load((const __attribute__((address_space(1))) v2d_t *)p);
  return make_double2(v.x, v.y);
#else
  return ld2(p);
#endif
}

__device__ __forceinline__ double2 ld2g(const double *p) { return ld2(p); }
__device__ __forceinline__ double2 ld2g_nt(const double *p) { return ld2_nt(p); }

// acc + a0 b0 + a1 b1 + a2 b2 + a3 b3 as four fused multiply-adds on the accumulator, in this order -- EVERY block
// product of the library (k_eval, the Hessian kernels, the one-launch solve and the one-launch iteration) goes through
// here, so they stay bitwise equal to each other.  (`acc += a0 * b0 + a1 * b1 + ...` compiles to mul, fma, fma, fma, add
// through one temporary: five dependent instructions where four independent accumulators' chains interleave.)
__device__ __forceinline__ double fma4(double a0, double b0, double a1, double b1, double a2, double b2, double a3, double b3,
                                       double acc) {
  return __builtin_fma(a3, b3, __builtin_fma(a2, b2, __builtin_fma(a1, b1, __builtin_fma(a0, b0, acc))));
}

// one group of up to 4 ELL slots: every index/block load is issued before the first use
template <int R, int NV, class Src>
__device__ __forceinline__ void ell_group(const AgentDev &ag, int j, int slot0, Src src, double (*acc)[4]) {
  const int W = ag.ell_w, n = ag.n;
  int idx[4];
  double2 B[4][8];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const bool valid = slot0 + u < W;
    idx[u] = valid ? gp(ag.ell_col)[(size_t)(slot0 + u) * n + j] : j;
    const double *bp = ag.ell_val + ((size_t)(valid ? slot0 + u : 0) * n + j) * 16;
#pragma unroll
    for (int q = 0; q < 8; ++q) B[u][q] = valid ? ld2(bp + 2 * q) : make_double2(0.0, 0.0);
  }
  double x[4][NV][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) src(idx[u], x[u]);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc[v][c] = fma4(x[u][v][0], B[u][2 * c].x, x[u][v][1], B[u][2 * c].y, x[u][v][2], B[u][2 * c + 1].x, x[u][v][3],
                         B[u][2 * c + 1].y, acc[v][c]);
}

// acc[v][c] += sum_i sum_cp src_v(i, cp) * Q_ij[cp, c]   for output pose j, row a; NV vectors at once
template <int R, int NV, class Src>
__device__ __forceinline__ void spmm_row(const AgentDev &ag, int j, Src src, double (*acc)[4]) {
  ell_group<R, NV>(ag, j, 0, src, acc);
  if (ag.ell_w > 4) ell_group<R, NV>(ag, j, 4, src, acc);
  const int p0 = gp(ag.trowptr)[j], p1 = gp(ag.trowptr)[j + 1];
  for (int p = p0; p < p1; ++p) {
    const int i = gp(ag.tcol)[p];
    const double *bp = ag.tval + (size_t)16 * p;
    double x[NV][4];
    src(i, x);
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double2 b01 = ld2(bp + 4 * c), b23 = ld2(bp + 4 * c + 2);
        acc[v][c] = fma4(x[v][0], b01.x, x[v][1], b01.y, x[v][2], b23.x, x[v][3], b23.y, acc[v][c]);
      }
  }
}

// G_j row a from the shared edges of public pose index q (a2).  The edges of a pose are taken four at a time: their
// descriptors (one round trip), then their neighbour poses and coefficients (one more) -- a pose of the tunnels data has
// up to 20 shared edges, and one edge after the other was 2 dependent round trips EACH (k_eval of a lockstep tick:
// 25 us).  The sums run edge after edge, as before.
template <int R>
__device__ __forceinline__ void g_row_range(const AgentDev &ag, int e0, int e1, int a, int aux, int pull, double g[4]) {
  g[0] = g[1] = g[2] = g[3] = 0.0;
  for (int eb = e0; eb < e1; eb += 4) {
    const double *xp[4];
    double *slab[4];
    bool copy[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const auto *se = gp(ag.se) + min(eb + u, e1 - 1);
      slab[u] = ag.nbr[aux] + (size_t)se->slot * 4 * R;
      const double *src = se->src[aux];
      copy[u] = pull && src;
      xp[u] = copy[u] ? src : slab[u];
    }
    double x[4][4], cf[4][16];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const auto *se = gp(ag.se) + min(eb + u, e1 - 1);
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) x[u][cp] = gp(xp[u])[cp * R + a];
#pragma unroll
      for (int i = 0; i < 16; ++i) cf[u][i] = se->coef[i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (eb + u < e1) {
        if (copy[u]) {
#pragma unroll
          for (int cp = 0; cp < 4; ++cp) gp(slab[u])[cp * R + a] = x[u][cp];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int cp = 0; cp < 4; ++cp) g[c] -= x[u][cp] * cf[u][cp + 4 * c];
      }
    }
  }
}

// the same sums from operands that k_eval_staged (spmm.hip) put into LDS: per edge [neighbour pose, 4R doubles][16
// coefficients]; edge after edge, coefficient after coefficient, exactly g_row_range's order (bitwise the same G)
template <int R>
__device__ __forceinline__ void g_row_lds(const double *Eop, int e0, int e1, int a, double g[4]) {
  constexpr int EPE = 4 * R + 16;
  g[0] = g[1] = g[2] = g[3] = 0.0;
  for (int e = e0; e < e1; ++e) {
    const double *E = Eop + (size_t)e * EPE;
    double x[4], cf[16];
#pragma unroll
    for (int cp = 0; cp < 4; ++cp) x[cp] = E[cp * R + a];
#pragma unroll
    for (int i = 0; i < 8; ++i) { const double2 t = *reinterpret_cast<const double2 *>(E + 4 * R + 2 * i); cf[2 * i] = t.x; cf[2 * i + 1] = t.y; }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) g[c] -= x[cp] * cf[cp + 4 * c];
  }
}

template <int R>
__device__ __forceinline__ void g_row(const AgentDev *__restrict__ agents, const AgentDev &ag, int q, int a, int aux, int pull,
                                      double g[4]) {
  g_row_range<R>(ag, ag.pub_ptr[q], ag.pub_ptr[q + 1], a, aux, pull, g);
}

// end of an iteration: advance gamma/alpha/iter of one agent
__device__ __forceinline__ void advance_agent(const AgentDev &ag, int accel, int num_robots, int restart_interval,
                                              int inc = 1) {
  NestState ns = *ag.nest;
  if (accel) {
    const double Nr = (double)num_robots;
    const bool restart = ((ns.iter + 2) % restart_interval) == 0;
    if (restart) { ns.gamma = 0; ns.alpha = 0; }
    else {
      ns.gamma = (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * ns.gamma * ns.gamma)) / (2.0 * Nr);
      ns.alpha = 1.0 / (ns.gamma * Nr);
    }
  }
  ns.iter += inc;
  *ag.nest = ns;
}

// ------------------------------------------------------------------------------------------------
// f, Euclidean gradient, Riemannian gradient (a3).  partials: [0] f, [1] |rgrad|^2
// gmode: 0 G from the buffer, 1 assemble G from the slab, 2 assemble G pulling from co-resident
// agents (both also store G).
// IN_WAVE: the tile runs on ONE wave of a larger workgroup (fused iteration kernel): `sel` is the agent index itself,
// the LDS exchange between the lanes of the tile is ordered by a wave-level fence instead of __syncthreads(), and the
// Riemannian gradient is stored write-through (agent-scope relaxed atomics = global_store sc1) because other
// workgroups of the SAME launch read it behind the grid barrier.
template <int R, bool IN_WAVE = false, bool BAKED = false>
__device__ __forceinline__ void eval_body(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, int xb, int egb, int gfb,
                                          int poff, int gmode, int aux, int bx, double *Ysh, double *Wsh,
                                          const AgentDev &agv, const double *Eop = nullptr, int ebase = 0) {
  const AgentDev &ag = pick_agent<BAKED>(agv, agents, BAKED ? 0 : (IN_WAVE ? sel : sel_cur(team, sel)));
  constexpr int PPB = 64 / R;
  const int lane = IN_WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x, lp = lane / R, a = lane - lp * R;
  const int j = bx * PPB + lp;
  if (bx * PPB >= ag.n) return;
  const bool act = lp < PPB && j < ag.n;
  const double *X = ag.buf[xb];
  double fpart = 0, gpart = 0, eg3 = 0;
  int e0 = 0, e1 = 0;
  double acc[1][4] = {{0, 0, 0, 0}};
  if (act) {
    // shared edges of this pose, requested in front of the SpMM so that the two chains of dependent round trips
    // (index -> X gather, edge range -> edge -> neighbour pose) run side by side
    e0 = gp(ag.pose_eptr)[j]; e1 = gp(ag.pose_eptr)[j + 1];
    spmm_row<R, 1>(ag, j, [&](int i, double(*x)[4]) {
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) x[0][cp] = gp(X)[((size_t)4 * i + cp) * R + a];
    }, acc);
  }
  // (k_eval_staged: the helper waves have put the operands of the tile's shared edges into LDS meanwhile.  The barrier
  // sits in uniform control flow: every wave of the workgroup -- this one and the helpers -- reaches it exactly once)
  if (Eop) __syncthreads();
  if (act) {
    double g[4] = {0, 0, 0, 0};
    double *Gj = ag.buf[B_G] + (size_t)j * 4 * R;
    if (e1 > e0) {
      if (gmode == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) g[c] = gp(Gj)[c * R + a];
      } else {
        // (Eop: the operands of this tile's shared edges are in LDS -- k_eval_staged -- and the pulled poses already
        // went to the slab)
        if (Eop) g_row_lds<R>(Eop, e0 - ebase, e1 - ebase, a, g);
        else g_row_range<R>(ag, e0, e1, a, aux, gmode == 2, g);
#pragma unroll
        for (int c = 0; c < 4; ++c) gp(Gj)[c * R + a] = g[c];
      }
    }
    double *EG = ag.buf[egb] + (size_t)j * 4 * R;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double xr = gp(X)[((size_t)4 * j + c) * R + a];
      fpart += (0.5 * acc[0][c] + g[c]) * xr;
      const double eg = acc[0][c] + g[c];
      gp(EG)[c * R + a] = eg;
      Ysh[lp * 4 * R + c * R + a] = xr;
      Wsh[lp * 4 * R + c * R + a] = eg;
      if (c == 3) eg3 = eg;
    }
  }
  if (IN_WAVE) {
    // LDS operations of one wave execute in order; the fence keeps the compiler from moving the reads up
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  } else {
    __syncthreads();
  }
  if (act) {
    double o[3];
    tangent_row<R>(Ysh + lp * 4 * R, Wsh + lp * 4 * R, a, o);
    double *GF = ag.buf[gfb] + (size_t)j * 4 * R;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (IN_WAVE) __hip_atomic_store(GF + c * R + a, o[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else gp(GF)[c * R + a] = o[c];
      gpart += o[c] * o[c];
    }
    if (IN_WAVE) __hip_atomic_store(GF + 3 * R + a, eg3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else gp(GF)[3 * R + a] = eg3;
    gpart += eg3 * eg3;
  }
  fpart = wave_sum(fpart);
  gpart = wave_sum(gpart);
  if (lane == 0) {
    double *P = ag.part + poff + (size_t)bx * PART_STRIDE;
    gp(P)[0] = fpart; gp(P)[1] = gpart;
  }
}

// shared tail of every Hessian-vector product: curvature correction + tangent projection.
// in : wrow[4] = (V Q)_j row a, vrow[4] = V_j row a, Ysh/Esh = full Y_j / egrad_j staged in LDS
// out: hrow[4] = Hess f[V]_j row a ;  Wsh used as scratch
template <int R>
__device__ __forceinline__ void hess_tail(const double *Ysh, const double *Esh, double *Wsh, int a,
                                          const double wrow[4], const double vrow[4], double hrow[4], bool act) {
  if (act) {
    double S[9];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        double s = 0;
#pragma unroll
        for (int b = 0; b < R; ++b) s += Ysh[p * R + b] * Esh[q * R + b];
        S[3 * p + q] = s;
      }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      double s = wrow[q];
#pragma unroll
      for (int p = 0; p < 3; ++p) s -= vrow[p] * 0.5 * (S[3 * p + q] + S[3 * q + p]);
      Wsh[q * R + a] = s;
    }
  }
  __syncthreads();
  if (act) {
    double o[3];
    tangent_row<R>(Ysh, Wsh, a, o);
    hrow[0] = o[0]; hrow[1] = o[1]; hrow[2] = o[2]; hrow[3] = wrow[3];
  }
}

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// per-pose kernels: one lane per pose with the whole pose in registers.  A 64-pose tile (64 * 4R
// contiguous doubles) moves between HBM and registers through LDS so that every global access is a
// fully coalesced 512-byte wave transaction instead of 64 strided 8-byte ones.
template <int R>
struct Tile {
  static constexpr int P = 4 * R + 1;  // odd pitch: conflict-free row access
  double d[64 * P];
};

// 64 lanes x 4R elements = exactly one tile: fixed trip count, every load issued before the first
// LDS store (a runtime-bounded loop makes the compiler wait for each load in turn)
template <int R>
__device__ __forceinline__ void tile_in(Tile<R> &t, const double *g, int j0, int cnt, int tid) {
  const double *src = g + (size_t)j0 * 4 * R;
  const int total = cnt * 4 * R;
  double tmp[4 * R];
#pragma unroll
  for (int k = 0; k < 4 * R; ++k) {
    const int e = tid + 64 * k;
    const double v = gp(src)[min(e, total - 1)];  // (an unconditional global load from a clamped index: no branch per element)
    tmp[k] = (e < total) ? v : 0.0;
  }
#pragma unroll
  for (int k = 0; k < 4 * R; ++k) {
    const int e = tid + 64 * k;
    t.d[(e / (4 * R)) * Tile<R>::P + e % (4 * R)] = tmp[k];
  }
}

template <int R>
__device__ __forceinline__ void tile_out(const Tile<R> &t, double *g, int j0, int cnt, int tid) {
  double *dst = g + (size_t)j0 * 4 * R;
  const int total = cnt * 4 * R;
#pragma unroll
  for (int k = 0; k < 4 * R; ++k) {
    const int e = tid + 64 * k;
    if (e < total) gp(dst)[e] = t.d[(e / (4 * R)) * Tile<R>::P + e % (4 * R)];
  }
}

template <int R>
__device__ __forceinline__ void tile_get(const Tile<R> &t, int row, double *v) {
#pragma unroll
  for (int i = 0; i < 4 * R; ++i) v[i] = t.d[row * Tile<R>::P + i];
}

template <int R>
__device__ __forceinline__ void tile_put(Tile<R> &t, int row, const double *v) {
#pragma unroll
  for (int i = 0; i < 4 * R; ++i) t.d[row * Tile<R>::P + i] = v[i];
}

// ------------------------------------------------------------------------------------------------
// Nesterov sequences (a6).  blockIdx.y = local agent.  For every agent:
//   XPrev = X;  gamma' = (1 + sqrt(1 + 4 N^2 gamma^2)) / 2N;  alpha = 1 / (gamma' N)
//   Y = proj((1 - alpha) X + alpha V);  X = Y
// and for the agents that do NOT optimize this iteration (everything but `sel`, or all when
// sel == -2):  V = proj(V)  [= proj(V + gamma (X - Y))], then the periodic restart X = XPrev,
// V = Y = X; partial [0] of PART_D = |X_new - XPrev|^2.  First kernel of an accelerated iteration:
// publishes team->cur_sel.
// `hook` runs once behind the loads of the agent's X and V tiles (k_nest_pre places reads of pinned host memory there: memory
// reads return in order, so issued in FRONT of the tiles' loads they would hold the step up for a PCIe round trip, issued
// behind them they land while it runs).
struct NoHook { __device__ __forceinline__ void operator()() const {} };
template <int R, class Hook = NoHook>
__device__ __forceinline__ void nest_pre_body(const AgentDev *__restrict__ agents, TeamDev *team, int sel, int only_agent,
                                              int num_robots, int restart_interval, int bx, int by, Tile<R> &TX,
                                              Tile<R> &TV, int fused_restart = 0, Hook hook = Hook()) {
  // fused_restart bit 0: the pipelined RGD sequence takes a restart iteration as one plain step from X (the accelerated
  // solve that the un-fused path runs first, and discards, is skipped), so the selected agent only saves XPrev.
  // bit 1 (keep X): iterate(true) whose neighbour poses have not all arrived -- the local solve is skipped and X stays
  // where it is (it does NOT move to Y), while Y and, afterwards, V are updated as in any accelerated iteration
  const int ai = only_agent >= 0 ? only_agent : by;
  const AgentDev &ag = agents[ai];
  const int selected = (sel == -2) ? -1 : sel_sched(team, sel);
  if (bx == 0 && by == 0 && threadIdx.x == 0 && sel == -1) team->cur_sel = selected;
  const int j0 = bx * 64, tid = threadIdx.x;
  if (j0 >= ag.n) return;
  const int cnt = min(64, ag.n - j0);
  const bool optimizing = (ai == selected);
  const NestState ns = *ag.nest;
  const double Nr = (double)num_robots;
  const double gamma = (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * ns.gamma * ns.gamma)) / (2.0 * Nr);
  const double alpha = 1.0 / (gamma * Nr);
  const bool restart = ((ns.iter + 2) % restart_interval) == 0;  // iter is pre-increment: (iter+1)+1
  if (bx == 0 && tid == 0) ag.scal[6] = gamma;  // read by the fused RGD tail instead of the (mutable) NestState
  tile_in<R>(TX, ag.buf[B_X], j0, cnt, tid);
  tile_in<R>(TV, ag.buf[B_V], j0, cnt, tid);
  hook();
  __syncthreads();
  tile_out<R>(TX, ag.buf[B_XPREV], j0, cnt, tid);
  const bool keep_x = (fused_restart & 2) != 0;
  if ((fused_restart & 1) && optimizing && restart) return;
  double x[4 * R], v[4 * R], y[4 * R];
  double rel = 0;
  if (tid < cnt) {
    tile_get<R>(TX, tid, x);
    tile_get<R>(TV, tid, v);
#pragma unroll
    for (int i = 0; i < 4 * R; ++i) y[i] = (1.0 - alpha) * x[i] + alpha * v[i];
    polar_inplace<R>(y);
    if (!optimizing && !restart) {
      polar_inplace<R>(v);
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) { const double d = y[i] - x[i]; rel += d * d; }
    }
  }
  __syncthreads();
  if (tid < cnt) {
    if (optimizing || !restart) { tile_put<R>(TX, tid, y); tile_put<R>(TV, tid, v); }
    // restart of a non-optimizing agent: X = XPrev (tile still holds x); V = Y = X
  }
  __syncthreads();
  if (optimizing) {
    tile_out<R>(TX, ag.buf[B_Y], j0, cnt, tid);
    if (!keep_x) tile_out<R>(TX, ag.buf[B_X], j0, cnt, tid);  // the local solve starts from Y, in place on X
  } else if (restart) {
    tile_out<R>(TX, ag.buf[B_Y], j0, cnt, tid);
    tile_out<R>(TX, ag.buf[B_V], j0, cnt, tid);
  } else {
    tile_out<R>(TX, ag.buf[B_Y], j0, cnt, tid);
    tile_out<R>(TX, ag.buf[B_X], j0, cnt, tid);
    tile_out<R>(TV, ag.buf[B_V], j0, cnt, tid);
  }
  if (!optimizing) {
    rel = wave_sum(rel);
    if (tid == 0) ag.part[PART_D + (size_t)bx * PART_STRIDE] = rel;
  }
}

// ---- exchange between the workgroups of ONE launch (rtr_fused.hip, the two-level preconditioner's phase A -> B)
// Publish / read flavours, kept as build switches because they were measured against each other (us per tCG iteration,
// DESIGN.md section 4): write-through stores + sc1 loads (default) 17.0 | write-through stores + plain loads behind an
// agent acquire per hand-off (DPGO_RTR_PLAIN_LD=1) 18.0 | plain stores + agent release per hand-off
// (DPGO_RTR_PLAIN_ST=1) 22.5-25.  DPGO_RTR_LDAUX is the cache policy of the reads (16 = sc1; 2 = nt and 18 = sc1 nt
// are slower; 1 = sc0 is faster and NOT valid: it hits this CU's L1).
#ifndef DPGO_RTR_PLAIN_ST
#define DPGO_RTR_PLAIN_ST 0
#endif
#ifndef DPGO_RTR_PLAIN_LD
#define DPGO_RTR_PLAIN_LD 0
#endif
#ifndef DPGO_RTR_LDAUX
#define DPGO_RTR_LDAUX 16  // buffer-load cache policy of the cross-workgroup reads: 16 = sc1
#endif
__device__ __forceinline__ void st_c(double *p, double v) {
#if DPGO_RTR_PLAIN_ST
  *p = v;
#else
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// How a workgroup reads what the others published with st_c: buffer loads with sc1 (they bypass this CU's L1 and are
// served by the L2 / the fabric), 8 or 16 bytes.  Ordinary loads to the compiler, so a batch of them is issued back
// to back.  The descriptor must be wave-uniform (built from scalar pointers; a lane-dependent choice of buffer turns
// every load into a waterfall loop).
typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
typedef unsigned int v2u_t __attribute__((ext_vector_type(2)));
struct CVec {
  __amdgpu_buffer_rsrc_t rs;
  __device__ __forceinline__ CVec(const double *p, int count)
      : rs(__builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p), 0, count * 8, 0x00020000)) {}
  __device__ __forceinline__ double ld(int i) const {
    const v2u_t r = __builtin_amdgcn_raw_buffer_load_b64(rs, i * 8, 0, DPGO_RTR_PLAIN_LD ? 0 : DPGO_RTR_LDAUX);
    double d;
    __builtin_memcpy(&d, &r, 8);
    return d;
  }
  __device__ __forceinline__ double2 ld2(int i) const {
    const v4u_t r = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 8, 0, DPGO_RTR_PLAIN_LD ? 0 : DPGO_RTR_LDAUX);
    double2 d;
    __builtin_memcpy(&d, &r, 16);
    return d;
  }
};

#define WSYNC()                                                  \
  do {                                                           \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       \
    __builtin_amdgcn_wave_barrier();                             \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       \
  } while (0)

#define DPGO_DISPATCH_R(R_, CALL)            \
  switch (R_) {                              \
    case 3: { constexpr int R = 3; CALL; } break; \
    case 4: { constexpr int R = 4; CALL; } break; \
    case 5: { constexpr int R = 5; CALL; } break; \
    case 6: { constexpr int R = 6; CALL; } break; \
    case 7: { constexpr int R = 7; CALL; } break; \
    case 8: { constexpr int R = 8; CALL; } break; \
    default: break;                          \
  }

static inline int spmm_grid(int r, int n) { const int ppb = 64 / r; return (n + ppb - 1) / ppb; }

}  // namespace dpgo
