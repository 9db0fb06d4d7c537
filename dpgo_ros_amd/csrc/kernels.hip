// kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the RBCD hot path.
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
  return team->group_members[team->group_ptr[SEL_GROUP0 - sel] + blockIdx.y];  // colour-parallel update
}

template <int R>
__device__ __forceinline__ int spmm_blocks(int n) { return (n + (64 / R) - 1) / (64 / R); }
__device__ __forceinline__ int precond_blocks(int N4) { return (N4 + 7) / 8; }

__device__ __forceinline__ double2 ld2(const double *p) { return *reinterpret_cast<const double2 *>(p); }
typedef double v2d_t __attribute__((ext_vector_type(2)));
#ifndef DPGO_M_NT
#define DPGO_M_NT 1
#endif
__device__ __forceinline__ double2 ld2_nt(const double *p) {
#if DPGO_M_NT
  const v2d_t v = __builtin_nontemporal_load(reinterpret_cast<const v2d_t *>(p));
  return make_double2(v.x, v.y);
#else
  return *reinterpret_cast<const double2 *>(p);
#endif
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
    idx[u] = valid ? ag.ell_col[(size_t)(slot0 + u) * n + j] : j;
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
        acc[v][c] += x[u][v][0] * B[u][2 * c].x + x[u][v][1] * B[u][2 * c].y + x[u][v][2] * B[u][2 * c + 1].x +
                     x[u][v][3] * B[u][2 * c + 1].y;
}

// acc[v][c] += sum_i sum_cp src_v(i, cp) * Q_ij[cp, c]   for output pose j, row a; NV vectors at once
template <int R, int NV, class Src>
__device__ __forceinline__ void spmm_row(const AgentDev &ag, int j, Src src, double (*acc)[4]) {
  ell_group<R, NV>(ag, j, 0, src, acc);
  if (ag.ell_w > 4) ell_group<R, NV>(ag, j, 4, src, acc);
  const int p0 = ag.trowptr[j], p1 = ag.trowptr[j + 1];
  for (int p = p0; p < p1; ++p) {
    const int i = ag.tcol[p];
    const double *bp = ag.tval + (size_t)16 * p;
    double x[NV][4];
    src(i, x);
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double2 b01 = ld2(bp + 4 * c), b23 = ld2(bp + 4 * c + 2);
        acc[v][c] += x[v][0] * b01.x + x[v][1] * b01.y + x[v][2] * b23.x + x[v][3] * b23.y;
      }
  }
}

// G_j row a from the shared edges of public pose index q (a2)
template <int R>
__device__ __forceinline__ void g_row(const AgentDev *agents, const AgentDev &ag, int q, int a, int aux, int pull,
                                      double g[4]) {
  g[0] = g[1] = g[2] = g[3] = 0.0;
  for (int e = ag.pub_ptr[q]; e < ag.pub_ptr[q + 1]; ++e) {
    const SharedEdgeDev &se = ag.se[e];
    double *slab = ag.nbr[aux] + (size_t)se.slot * 4 * R;
    double x[4];
    if (pull && se.src_agent_local >= 0) {
      const double *src = agents[se.src_agent_local].buf[aux ? B_Y : B_X] + (size_t)se.src_frame * 4 * R;
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) { x[cp] = src[cp * R + a]; slab[cp * R + a] = x[cp]; }
    } else {
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) x[cp] = slab[cp * R + a];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) g[c] -= x[cp] * se.coef[cp + 4 * c];
  }
}

// ------------------------------------------------------------------------------------------------
// stand-alone G assembly.  One lane per (public pose, row a).  pull != 0: read the neighbour's pose
// straight from the neighbour agent's X / Y array on this GPU (device-to-device exchange that
// replaces the PublicPoses topic) and refresh the slab; else read the slab filled by unpack.
template <int R>
__global__ __launch_bounds__(64) void k_buildG(const AgentDev *agents, const TeamDev *team, int sel, int aux,
                                               int pull) {
  const AgentDev &ag = agents[sel_cur(team, sel)];
  constexpr int PPB = 64 / R;
  const int lane = threadIdx.x, lp = lane / R, a = lane - lp * R;
  const int q = blockIdx.x * PPB + lp;
  if (lp >= PPB || q >= ag.npub) return;
  double g[4];
  g_row<R>(agents, ag, q, a, aux, pull, g);
  double *G = ag.buf[B_G] + (size_t)ag.pub_pose[q] * 4 * R;
#pragma unroll
  for (int c = 0; c < 4; ++c) G[c * R + a] = g[c];
}

// refresh every slab entry of one agent from co-resident neighbours (both sequences)
template <int R>
__global__ void k_pull(const AgentDev *agents, int dst) {
  const AgentDev &ag = agents[dst];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ag.nshared * 4 * R) return;
  const int e = t / (4 * R), k = t - e * 4 * R;
  const SharedEdgeDev &se = ag.se[e];
  if (se.src_agent_local < 0) return;
  const AgentDev &sa = agents[se.src_agent_local];
  ag.nbr[0][(size_t)se.slot * 4 * R + k] = sa.buf[B_X][(size_t)se.src_frame * 4 * R + k];
  ag.nbr[1][(size_t)se.slot * 4 * R + k] = sa.buf[B_Y][(size_t)se.src_frame * 4 * R + k];
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
template <int R>
__device__ __forceinline__ void eval_body(const AgentDev *agents, const TeamDev *team, int sel, int xb, int egb, int gfb,
                                          int poff, int gmode, int aux, int bx, double *Ysh, double *Wsh) {
  const AgentDev &ag = agents[sel_cur(team, sel)];
  constexpr int PPB = 64 / R;
  const int lane = threadIdx.x, lp = lane / R, a = lane - lp * R;
  const int j = bx * PPB + lp;
  if (bx * PPB >= ag.n) return;
  const bool act = lp < PPB && j < ag.n;
  const double *X = ag.buf[xb];
  double fpart = 0, gpart = 0, eg3 = 0;
  if (act) {
    double acc[1][4] = {{0, 0, 0, 0}};
    spmm_row<R, 1>(ag, j, [&](int i, double(*x)[4]) {
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) x[0][cp] = X[((size_t)4 * i + cp) * R + a];
    }, acc);
    double g[4] = {0, 0, 0, 0};
    const int q = ag.pub_index[j];
    double *Gj = ag.buf[B_G] + (size_t)j * 4 * R;
    if (q >= 0) {
      if (gmode == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) g[c] = Gj[c * R + a];
      } else {
        g_row<R>(agents, ag, q, a, aux, gmode == 2, g);
#pragma unroll
        for (int c = 0; c < 4; ++c) Gj[c * R + a] = g[c];
      }
    }
    double *EG = ag.buf[egb] + (size_t)j * 4 * R;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double xr = X[((size_t)4 * j + c) * R + a];
      fpart += (0.5 * acc[0][c] + g[c]) * xr;
      const double eg = acc[0][c] + g[c];
      EG[c * R + a] = eg;
      Ysh[lp * 4 * R + c * R + a] = xr;
      Wsh[lp * 4 * R + c * R + a] = eg;
      if (c == 3) eg3 = eg;
    }
  }
  __syncthreads();
  if (act) {
    double o[3];
    tangent_row<R>(Ysh + lp * 4 * R, Wsh + lp * 4 * R, a, o);
    double *GF = ag.buf[gfb] + (size_t)j * 4 * R;
#pragma unroll
    for (int c = 0; c < 3; ++c) { GF[c * R + a] = o[c]; gpart += o[c] * o[c]; }
    GF[3 * R + a] = eg3;
    gpart += eg3 * eg3;
  }
  fpart = wave_sum(fpart);
  gpart = wave_sum(gpart);
  if (lane == 0) {
    double *P = ag.part + poff + (size_t)bx * PART_STRIDE;
    P[0] = fpart; P[1] = gpart;
  }
}

template <int R>
__global__ __launch_bounds__(64) void k_eval(const AgentDev *agents, const TeamDev *team, int sel, int xb, int egb,
                                             int gfb, int poff, int gmode, int aux) {
  constexpr int PPB = 64 / R;
  __shared__ double Ysh[PPB * 4 * R], Wsh[PPB * 4 * R];
  eval_body<R>(agents, team, sel, xb, egb, gfb, poff, gmode, aux, (int)blockIdx.x, Ysh, Wsh);
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

// generic Riemannian Hessian-vector product at point xb with Euclidean gradient egb:  ob = Hess[vb]
// partials: [0] <v, Hv>
template <int R>
__global__ __launch_bounds__(64) void k_hess(const AgentDev *agents, const TeamDev *team, int sel, int xb, int egb,
                                             int vb, int ob, int poff) {
  const AgentDev &ag = agents[sel_cur(team, sel)];
  constexpr int PPB = 64 / R;
  __shared__ double Ysh[PPB * 4 * R], Esh[PPB * 4 * R], Wsh[PPB * 4 * R];
  const int lane = threadIdx.x, lp = lane / R, a = lane - lp * R;
  const int j = blockIdx.x * PPB + lp;
  if (blockIdx.x * PPB >= ag.n) return;
  const bool act = lp < PPB && j < ag.n;
  const double *V = ag.buf[vb];
  double w[1][4] = {{0, 0, 0, 0}}, vrow[4] = {0, 0, 0, 0}, hrow[4];
  if (act) {
    spmm_row<R, 1>(ag, j, [&](int i, double(*x)[4]) {
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) x[0][cp] = V[((size_t)4 * i + cp) * R + a];
    }, w);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      vrow[c] = V[((size_t)4 * j + c) * R + a];
      Ysh[lp * 4 * R + c * R + a] = ag.buf[xb][((size_t)4 * j + c) * R + a];
      Esh[lp * 4 * R + c * R + a] = ag.buf[egb][((size_t)4 * j + c) * R + a];
    }
  }
  __syncthreads();
  hess_tail<R>(Ysh + lp * 4 * R, Esh + lp * 4 * R, Wsh + lp * 4 * R, a, w[0], vrow, hrow, act);
  double d = 0;
  if (act) {
    double *O = ag.buf[ob] + (size_t)j * 4 * R;
#pragma unroll
    for (int c = 0; c < 4; ++c) { O[c * R + a] = hrow[c]; d += vrow[c] * hrow[c]; }
  }
  d = wave_sum(d);
  if (lane == 0) ag.part[poff + (size_t)blockIdx.x * PART_STRIDE] = d;
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
    tmp[k] = (e < total) ? src[e] : 0.0;
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
    if (e < total) dst[e] = t.d[(e / (4 * R)) * Tile<R>::P + e % (4 * R)];
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
template <int R>
__device__ __forceinline__ void nest_pre_body(const AgentDev *agents, TeamDev *team, int sel, int only_agent,
                                              int num_robots, int restart_interval, int bx, int by, Tile<R> &TX,
                                              Tile<R> &TV) {
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
  __syncthreads();
  tile_out<R>(TX, ag.buf[B_XPREV], j0, cnt, tid);
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
    tile_out<R>(TX, ag.buf[B_X], j0, cnt, tid);  // the local solve starts from Y, in place on X
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
//               ahead (pipelined iterations, see k_eval_stats): bit 0 = the first wave also takes the Nesterov step of
//               iteration k+1 of its two poses, bit 1 = the second wave takes it for the workgroup's share of the
//               other agents' poses; advance: 1 = end-of-iteration bookkeeping here, 2 = pipelined (publishes
//               stats_sel / next_sel only)
// KC = rows of M (scalars of the input vector) handled per chunk: KC * R * 8 bytes of LDS and KC / 64
// 16-byte registers per lane.  One 2048-row chunk covers a 500-pose agent in a single round trip with one
// workgroup per CU; larger agents use 1024-row chunks so that 3 workgroups fit a CU and one workgroup's
// arithmetic overlaps the others' streams.

template <int R, int MODE, int KC>
__global__ __launch_bounds__(256) void k_precond(const AgentDev *agents, TeamDev *team, int sel, int xb, int vb,
                                                 int zb, int sp, int max_inner, double step, int accel,
                                                 int num_robots, int advance, int restart_interval, int ahead) {
  // XCD-aware block order: hardware workgroup h runs on XCD h % 8 (each with its own L2).  Logical block
  // (h % 8) * (grid / 8) + h / 8 gives every XCD one contiguous range of poses, so that the cache lines shared by
  // neighbouring poses (a pose is 4R doubles, not a multiple of a line) are written inside one L2 instead of
  // being split between two.  The grid is padded to a multiple of 8; padding blocks fall out at the nblk test.
  const int bx = ((int)blockIdx.x % 8) * ((int)gridDim.x / 8) + (int)blockIdx.x / 8;
  const AgentDev &ag = agents[sel_cur(team, sel)];
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
  constexpr int MREG = KC / 64;
  __shared__ double vs[R * KC];
  __shared__ double zs[8 * R];
  __shared__ double Ysh[2 * 4 * R];
  __shared__ double Esh[3][2 * 4 * R];  // PM_RGD: V, Yaux, XPrev of the two poses
  const int tid = threadIdx.x, lane = tid & 63;
  const int N4 = ag.N4;
  const int nblk = precond_blocks(N4);
  if (bx >= nblk) return;

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
  const int col0 = 8 * bx;
  const int npose = min(2, ag.n - 2 * bx);

  if (MODE == PM_TCG_STEP_) {
    // eta += (alpha | tau) * delta on the two poses owned by this workgroup
    const double *D = ag.buf[jpar ? B_D1 : B_D0];
    double *E = ag.buf[B_ETA];
    const double stepc = boundary ? tau : alpha;
    if (tid < npose * 4 * R) {
      const size_t o = (size_t)col0 * R + tid;
      E[o] += stepc * D[o];
    }
    if (boundary) return;
  }

  // epilogue operands of the two poses this workgroup owns: requested now, consumed after the M stream
  double pre_x = 0, pre_v = 0, pre_y = 0, pre_p = 0;
  double nest_gamma = 0;
  if (tid < npose * 4 * R) {
    pre_x = ag.buf[xb][(size_t)col0 * R + tid];
    if (MODE == PM_RGD_) {
      pre_v = ag.buf[B_V][(size_t)col0 * R + tid];
      pre_y = ag.buf[B_Y][(size_t)col0 * R + tid];
      pre_p = ag.buf[B_XPREV][(size_t)col0 * R + tid];
    }
  }
  double ahead_alpha = 0;
  bool ahead_opt = false;
  if (MODE == PM_RGD_ && accel) {
    if (advance == 2) {
      // the NestState describes iteration k-1 (it is advanced by the next k_eval_stats): gamma of this iteration,
      // and gamma / alpha / selected agent of iteration k+1 for the look-ahead Nesterov step of the epilogue
      const NestState ns = *ag.nest;
      const double Nr = (double)num_robots;
      nest_gamma = (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * ns.gamma * ns.gamma)) / (2.0 * Nr);
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
  for (int k0 = 0; k0 < N4; k0 += KC) {
    const int kn = min(KC, N4 - k0);
    if (k0 > 0) __syncthreads();
    {
      double2 v[NSTG];
#pragma unroll
      for (int u = 0; u < NSTG; ++u) {
        const int tt = 2 * (tid + 256 * u);  // kn * R is even
        v[u] = (tt < kn * R) ? ld2(Vstage + (size_t)k0 * R + tt) : make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int u = 0; u < NSTG; ++u) {
        const int tt = 2 * (tid + 256 * u);
        if (tt < KC * R) *reinterpret_cast<double2 *>(&vs[tt]) = v[u];
      }
    }
    __syncthreads();
    double2 mreg[MREG];
#pragma unroll
    for (int m = 0; m < MREG; ++m) {
      const int k = 2 * kl + 64 * m;
      mreg[m] = (cact && k < kn) ? ld2_nt(Mc + k0 + k) : make_double2(0.0, 0.0);
    }
    if (k0 == 0 && MODE == PM_RGD_ && (ahead & 2)) {
      // look-ahead operands of the second wave, requested right behind the M slab: they arrive under the stream.
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
        for (int i = 0; i < 4 * R; ++i) { la_x[i] = xa[o + i]; la_v[i] = va[o + i]; }
      }
    }
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
  }
#pragma unroll
  for (int a = 0; a < R; ++a) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc[a] += __shfl_xor(acc[a], off, 64);
  }
  if (kl == 0) {
#pragma unroll
    for (int a = 0; a < R; ++a) zs[cg * R + a] = acc[a];
  }
  if (tid < npose * 4 * R) {
    Ysh[tid] = pre_x;
    if (MODE == PM_RGD_) { Esh[0][tid] = pre_v; Esh[1][tid] = pre_y; Esh[2][tid] = pre_p; }
  }
  __syncthreads();

  if (MODE == PM_RGD_ && (ahead & 2) && tid >= 64 && tid < 128) {
    // look-ahead of the other agents' poses on the second wave (operands prefetched in the prologue)
    if (la_act) {
      const AgentDev &oa = agents[la_agent];
      const size_t o = (size_t)la_pose * 4 * R;
      double y[4 * R];
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) y[i] = (1.0 - ahead_alpha) * la_x[i] + ahead_alpha * la_v[i];
      polar_inplace<R>(y);
      if (!la_opt) {
        polar_inplace<R>(la_v);
        double r2 = 0;
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) { const double d = y[i] - la_x[i]; r2 += d * d; }
        oa.part[PART_D + la_pose] = r2;
      }
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) {
        oa.buf[B_XPREV][o + i] = la_x[i];
        oa.buf[B_Y][o + i] = y[i];
        oa.buf[B_X][o + i] = y[i];
        if (!la_opt) oa.buf[B_V][o + i] = la_v[i];
      }
    }
    return;
  }
  if (MODE == PM_RGD_) {
    // one lane per pose finishes the step in registers: z = P(zs), X = qf(X - step z), V update
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
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) {
        if (!(ahead & 1)) ag.buf[B_X][o + i] = x[i];
        ag.buf[B_X2][o + i] = x[i];  // snapshot for the final-statistics evaluation of this iteration
        const double d = x[i] - Esh[2][lp * 4 * R + i];
        rel += d * d;
      }
      double v[4 * R];
      if (accel) {
        const double gamma = nest_gamma;
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) v[i] = Esh[0][lp * 4 * R + i] + gamma * (x[i] - Esh[1][lp * 4 * R + i]);
        polar_inplace<R>(v);
      }
      if (accel && (ahead & 1)) {
        // Nesterov step of iteration k+1 for this pose (what k_nest_pre would do next): XPrev = X,
        // Y = proj((1 - alpha') X + alpha' V), X = Y, and V = proj(V) unless this agent is selected again
        double y[4 * R];
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) y[i] = (1.0 - ahead_alpha) * x[i] + ahead_alpha * v[i];
        polar_inplace<R>(y);
        if (!ahead_opt) {
          polar_inplace<R>(v);
          double rel2 = 0;
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) { const double d = y[i] - x[i]; rel2 += d * d; }
          ag.part[PART_D + 2 * bx + lp] = rel2;  // look-ahead steps leave |Y' - X|^2 per pose
        }
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) {
          ag.buf[B_XPREV][o + i] = x[i];
          ag.buf[B_Y][o + i] = y[i];
          ag.buf[B_X][o + i] = y[i];
        }
      }
      if (accel) {
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) ag.buf[B_V][o + i] = v[i];
      }
    }
    if (tid < 64) {
      rel = wave_sum(rel);
      if (tid == 0) ag.part[PART_B + (size_t)bx * PART_STRIDE + 2] = rel;
    }
    return;
  }

  // ---- epilogue: tangent projection of the two poses, dots, mode-specific stores
  double zr = 0, rr = 0;
  if (tid < npose * R) {
    const int lp = tid / R, a = tid - lp * R;
    const size_t o = (size_t)(2 * bx + lp) * 4 * R;
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
        ag.buf[B_R0][o + c * R + a] = v;
        ag.buf[B_ETA][o + c * R + a] = 0.0;
        ag.buf[B_D0][o + c * R + a] = -z[c];
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

// ------------------------------------------------------------------------------------------------
// tCG body, part 1:  delta <- -z + beta delta (on the fly), Hd = Hess[delta], partial <delta, Hd>.
template <int R>
__global__ __launch_bounds__(64) void k_tcg_hv(const AgentDev *agents, const TeamDev *team, int sel, int sp,
                                               int max_inner) {
  const AgentDev &ag = agents[sel_cur(team, sel)];
  constexpr int PPB = 64 / R;
  __shared__ double Ysh[PPB * 4 * R], Esh[PPB * 4 * R], Wsh[PPB * 4 * R];
  const int lane = threadIdx.x, lp = lane / R, a = lane - lp * R;
  const int j = blockIdx.x * PPB + lp;
  if (blockIdx.x * PPB >= ag.n) return;
  const RtrState S = ag.st[sp];
  if (S.outer_done || !S.tcg_active) {
    if (blockIdx.x == 0 && lane == 0) ag.st[sp ^ 1] = S;
    return;
  }
  const double kappa = 0.1;  // tCG stop: |r| <= |r0| min(|r0|^theta, kappa) with theta = 1
  const int npb = precond_blocks(ag.N4);
  double zr_new, rr_new;
  sum_partials2(ag.part + PART_B, npb, PART_STRIDE, lane, zr_new, rr_new);
  RtrState T = S;
  double beta = 0;
  const bool fresh = (S.tcg_j == 0);
  if (fresh) {
    T.z_r = zr_new; T.d_Pd = zr_new; T.norm_r0 = sqrt(rr_new);
  } else {
    const double nr = sqrt(rr_new);
    const double thr = S.norm_r0;
    bool stop = false;
    if (nr <= S.norm_r0 * (thr < kappa ? thr : kappa)) { T.tcg_status = (kappa < thr) ? 3 : 4; stop = true; }
    else if (S.tcg_j >= max_inner) { T.tcg_status = 0; stop = true; }
    if (stop) {
      T.tcg_active = 0;
      if (blockIdx.x == 0 && lane == 0) ag.st[sp ^ 1] = T;
      return;
    }
    beta = zr_new / S.z_r;
    T.e_Pd = beta * (S.e_Pd + S.alpha * S.d_Pd);
    T.d_Pd = zr_new + beta * beta * S.d_Pd;
    T.z_r = zr_new;
  }
  T.hv_count = S.hv_count + 1;
  T.tcg_total = S.tcg_total + 1;
  if (blockIdx.x == 0 && lane == 0) ag.st[sp ^ 1] = T;

  const bool act = lp < PPB && j < ag.n;
  const int jp = S.tcg_j & 1;
  const double *Dold = ag.buf[jp ? B_D0 : B_D1];  // delta of iteration j-1
  double *Dnew = ag.buf[jp ? B_D1 : B_D0];        // delta of iteration j (T0 wrote D0 for j = 0)
  const double *Z = ag.buf[B_Z];
  double w[1][4] = {{0, 0, 0, 0}}, vrow[4] = {0, 0, 0, 0}, hrow[4];
  if (act) {
    spmm_row<R, 1>(ag, j, [&](int i, double(*x)[4]) {
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) {
        const size_t o = ((size_t)4 * i + cp) * R + a;
        x[0][cp] = fresh ? Dnew[o] : (-Z[o] + beta * Dold[o]);
      }
    }, w);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const size_t o = ((size_t)4 * j + c) * R + a;
      vrow[c] = fresh ? Dnew[o] : (-Z[o] + beta * Dold[o]);
      if (!fresh) Dnew[o] = vrow[c];
      Ysh[lp * 4 * R + c * R + a] = ag.buf[B_X][o];
      Esh[lp * 4 * R + c * R + a] = ag.buf[B_EGRAD][o];
    }
  }
  __syncthreads();
  hess_tail<R>(Ysh + lp * 4 * R, Esh + lp * 4 * R, Wsh + lp * 4 * R, a, w[0], vrow, hrow, act);
  double d = 0;
  if (act) {
    double *O = ag.buf[B_HD] + (size_t)j * 4 * R;
#pragma unroll
    for (int c = 0; c < 4; ++c) { O[c * R + a] = hrow[c]; d += vrow[c] * hrow[c]; }
  }
  d = wave_sum(d);
  if (lane == 0) ag.part[PART_A + (size_t)blockIdx.x * PART_STRIDE] = d;
}

// out = Retr_x(scale * eta).  guard_state >= 0: skip when the trust-region state says done.
template <int R>
__global__ __launch_bounds__(64) void k_retract(const AgentDev *agents, const TeamDev *team, int sel, int xb, int eb,
                                                double scale, int ob, int guard_state) {
  const AgentDev &ag = agents[sel_cur(team, sel)];
  if (guard_state >= 0) {
    const RtrState S = ag.st[guard_state];
    if (S.outer_done || S.tcg_active || S.need_init) return;  // only between the end of tCG and the accept step
  }
  const int j0 = blockIdx.x * 64, tid = threadIdx.x;
  if (j0 >= ag.n) return;
  const int cnt = min(64, ag.n - j0);
  __shared__ Tile<R> TA, TB;
  tile_in<R>(TA, ag.buf[xb], j0, cnt, tid);
  tile_in<R>(TB, ag.buf[eb], j0, cnt, tid);
  __syncthreads();
  if (tid < cnt) {
    double x[4 * R], e[4 * R];
    tile_get<R>(TA, tid, x);
    tile_get<R>(TB, tid, e);
#pragma unroll
    for (int i = 0; i < 4 * R; ++i) x[i] += scale * e[i];
    qf_inplace<R>(x);
    tile_put<R>(TA, tid, x);
  }
  __syncthreads();
  tile_out<R>(TA, ag.buf[ob], j0, cnt, tid);
}

// raw-pointer manifold ops (unit parity + set-up): OP 0 polar projection, 1 tangent projection, 2 retraction
template <int R, int OP>
__global__ __launch_bounds__(64) void k_raw_op(const double *X, const double *V, double *out, int n) {
  const int j0 = blockIdx.x * 64, tid = threadIdx.x;
  const int cnt = min(64, n - j0);
  __shared__ Tile<R> TA, TB;
  tile_in<R>(TA, X, j0, cnt, tid);
  if (OP != 0) tile_in<R>(TB, V, j0, cnt, tid);
  __syncthreads();
  if (tid < cnt) {
    double x[4 * R], v[4 * R];
    tile_get<R>(TA, tid, x);
    if (OP == 0) { polar_inplace<R>(x); tile_put<R>(TA, tid, x); }
    if (OP == 1) { tile_get<R>(TB, tid, v); tangent_inplace<R>(x, v); tile_put<R>(TA, tid, v); }
    if (OP == 2) {
      tile_get<R>(TB, tid, v);
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) x[i] += v[i];
      qf_inplace<R>(x);
      tile_put<R>(TA, tid, x);
    }
  }
  __syncthreads();
  tile_out<R>(TA, out, j0, cnt, tid);
}

template <int R>
__global__ __launch_bounds__(64) void k_nest_pre(const AgentDev *agents, TeamDev *team, int sel, int only_agent,
                                                 int num_robots, int restart_interval) {
  __shared__ Tile<R> TX, TV;
  nest_pre_body<R>(agents, team, sel, only_agent, num_robots, restart_interval, (int)blockIdx.x, (int)blockIdx.y, TX, TV);
}

// Heterogeneous launch that closes iteration k and opens iteration k+1 inside captured graphs: the first
// nest_tiles * num_agents workgroups run the Nesterov step of every agent (k_nest_pre), the rest evaluate
// f_opt / gradnorm_opt of the agent that just optimized on its snapshot B_X2 (k_eval, sel = stats_sel).
// The two halves touch disjoint data: the statistics read B_X2 / G of agent a, the Nesterov step writes
// X, Y, V, XPrev; one launch boundary per iteration disappears.
template <int R>
__global__ __launch_bounds__(64) void k_stats_nest(const AgentDev *agents, TeamDev *team, int nest_tiles, int num_agents,
                                                   int num_robots, int restart_interval) {
  __shared__ Tile<R> TX, TV;
  const int nb_nest = nest_tiles * num_agents;
  const int b = (int)blockIdx.x;
  if (b < nb_nest) {
    nest_pre_body<R>(agents, team, -1, -1, num_robots, restart_interval, b % nest_tiles, b / nest_tiles, TX, TV);
  } else {
    eval_body<R>(agents, team, -5, B_X2, B_EGRAD2, B_GF2, PART_A, 0, 0, b - nb_nest, TX.d, TV.d);
  }
}

// Pipelined accelerated RGD iterations (dpgo_team_run): two launches per iteration.
//   k_eval_stats      cost / gradient of the agent of iteration k (G from the neighbours' Y)  ||  final statistics of
//                     iteration k-1 on its snapshot  ||  end-of-iteration bookkeeping of k-1 (workgroup 0)
//   k_precond<PM_RGD> preconditioned step + Nesterov V of iteration k and the Nesterov step of iteration k+1 of the same
//                     poses (first wave of each workgroup)  ||  the Nesterov step of iteration k+1 of the workgroup's
//                     share of every other agent's poses (second wave)
// No workgroup reads what another workgroup of the same launch writes: this kernel's workgroups read next_sel /
// stats_sel (written by the previous step kernel) while workgroup 0 moves iter, cur_sel and the NestStates, which
// only the step kernel reads.
template <int R>
__global__ __launch_bounds__(64) void k_eval_stats(const AgentDev *agents, TeamDev *team, int nb_eval, int first,
                                                   int has_eval, int has_stats, int num_robots, int restart_interval) {
  constexpr int PPB = 64 / R;
  __shared__ double Ysh[PPB * 4 * R], Wsh[PPB * 4 * R];
  const int b = (int)blockIdx.x;
  if (b == (int)gridDim.x - 1) {  // the extra workgroup: bookkeeping only, so that no evaluation waits for it
    if (!first && threadIdx.x == 0) {
      for (int k = 0; k < team->num_agents; ++k) advance_agent(agents[k], 1, num_robots, restart_interval);
      team->iter += 1;
      if (has_eval) team->cur_sel = team->next_sel;
    }
    return;
  }
  if (has_eval && b < nb_eval) {
    eval_body<R>(agents, team, first ? -1 : -6, B_X, B_EGRAD, B_GF, PART_C, 2, 1, b, Ysh, Wsh);
  } else if (has_stats) {
    eval_body<R>(agents, team, -5, B_X2, B_EGRAD2, B_GF2, PART_A, 0, 0, b - (has_eval ? nb_eval : 0), Ysh, Wsh);
  }
}

// after the selected agent's local solve (unfused path):  V = proj(V + gamma' (X - Y)); on restart
// X = XPrev (the host then re-optimizes from XPrev and calls k_nest_reset).
template <int R>
__global__ __launch_bounds__(64) void k_nest_post(const AgentDev *agents, const TeamDev *team, int sel, int num_robots,
                                                  int restart_interval) {
  const AgentDev &ag = agents[sel_cur(team, sel)];
  const int j0 = blockIdx.x * 64, tid = threadIdx.x;
  if (j0 >= ag.n) return;
  const int cnt = min(64, ag.n - j0);
  const NestState ns = *ag.nest;
  const double Nr = (double)num_robots;
  const double gamma = (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * ns.gamma * ns.gamma)) / (2.0 * Nr);
  const bool restart = ((ns.iter + 2) % restart_interval) == 0;
  __shared__ Tile<R> TX, TV, TY;
  if (restart) {
    tile_in<R>(TX, ag.buf[B_XPREV], j0, cnt, tid);
    __syncthreads();
    tile_out<R>(TX, ag.buf[B_X], j0, cnt, tid);
    return;
  }
  tile_in<R>(TX, ag.buf[B_X], j0, cnt, tid);
  tile_in<R>(TV, ag.buf[B_V], j0, cnt, tid);
  tile_in<R>(TY, ag.buf[B_Y], j0, cnt, tid);
  __syncthreads();
  if (tid < cnt) {
    double x[4 * R], v[4 * R], y[4 * R];
    tile_get<R>(TX, tid, x);
    tile_get<R>(TV, tid, v);
    tile_get<R>(TY, tid, y);
#pragma unroll
    for (int i = 0; i < 4 * R; ++i) v[i] += gamma * (x[i] - y[i]);
    polar_inplace<R>(v);
    tile_put<R>(TV, tid, v);
  }
  __syncthreads();
  tile_out<R>(TV, ag.buf[B_V], j0, cnt, tid);
}

// V = X; Y = X  (restart tail / weight update)
__global__ void k_nest_reset(const AgentDev *agents, const TeamDev *team, int sel, int r) {
  const AgentDev &ag = agents[sel_cur(team, sel)];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ag.N4 * r) return;
  const double x = ag.buf[B_X][t];
  ag.buf[B_V][t] = x;
  ag.buf[B_Y][t] = x;
}

// end of an iteration (unfused paths): advance gamma/alpha/iter of every agent, and the team counter
__global__ void k_advance(const AgentDev *agents, TeamDev *team, int only_agent, int accel, int num_robots,
                          int restart_interval, int bump_team, int inc, int team_inc) {
  const int ai = only_agent >= 0 ? only_agent : (int)blockIdx.x;
  if (threadIdx.x != 0) return;
  advance_agent(agents[ai], accel, num_robots, restart_interval, inc);
  if (bump_team && ai == 0) team->iter += team_inc;
}

// PART_D partial [0] = |X - XPrev|_F^2 over a 64-pose tile (blockIdx.y = agent when sel == -3)
template <int R>
__global__ __launch_bounds__(64) void k_status(const AgentDev *agents, const TeamDev *team, int sel, int only_agent) {
  const int ai = only_agent >= 0 ? only_agent : (sel == -3 ? (int)blockIdx.y : sel_cur(team, sel));
  const AgentDev &ag = agents[ai];
  const int j0 = blockIdx.x * 64, tid = threadIdx.x;
  if (j0 >= ag.n) return;
  const size_t lo = (size_t)j0 * 4 * R, hi = (size_t)min(ag.n, j0 + 64) * 4 * R;
  double s = 0, xa[4 * R], xb[4 * R];
#pragma unroll
  for (int k = 0; k < 4 * R; ++k) {
    const size_t t = lo + tid + 64 * k;
    xa[k] = (t < hi) ? ag.buf[B_X][t] : 0.0;
    xb[k] = (t < hi) ? ag.buf[B_XPREV][t] : 0.0;
  }
#pragma unroll
  for (int k = 0; k < 4 * R; ++k) { const double d = xa[k] - xb[k]; s += d * d; }
  s = wave_sum(s);
  if (tid == 0) ag.part[PART_D + (size_t)blockIdx.x * PART_STRIDE] = s;
}

// buf[to] = buf[from] for one agent or (sel == -3) every agent (blockIdx.y).  As the first kernel of a
// non-accelerated iteration (publish != 0) it also publishes team->cur_sel.
__global__ void k_copy(const AgentDev *agents, TeamDev *team, int sel, int only_agent, int r, int from, int to,
                       int publish) {
  const int ai = only_agent >= 0 ? only_agent : (sel == -3 ? (int)blockIdx.y : sel_cur(team, sel));
  if (publish && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    team->cur_sel = team->sched[team->iter % team->sched_len];
  const AgentDev &ag = agents[ai];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ag.N4 * r) return;
  ag.buf[to][t] = ag.buf[from][t];
}

// trust-region set-up from the initial evaluation partials
template <int R>
__global__ void k_rtr_begin(const AgentDev *agents, const TeamDev *team, int sel, double Delta0, double tol,
                            int max_outer) {
  const AgentDev &ag = agents[sel_cur(team, sel)];
  const int lane = threadIdx.x;
  const int nb = spmm_blocks<R>(ag.n);
  const double f = sum_partials(ag.part + PART_A, nb, PART_STRIDE, lane);
  const double g = sum_partials(ag.part + PART_A + 1, nb, PART_STRIDE, lane);
  if (lane != 0) return;
  RtrState S = {};
  S.f1 = f; S.ngf = sqrt(g); S.Delta = Delta0;
  S.f_init = f; S.gn_init = S.ngf;
  S.outer_done = (S.ngf < tol) || (max_outer <= 0);
  S.need_init = 1;
  ag.st[0] = S;
  ag.st[1] = S;
}

// outer step, evaluation at the candidate x2 = Retr_x1(eta):
//   egrad2 = x2 Q + G, rgrad2, Heta = Hess_x1[eta];  partials [0] f2 [1] |rgrad2|^2 [2] <gf,eta> [3] <eta,Heta>
// (both SpMM rows share every Q block load)
template <int R>
__global__ __launch_bounds__(64) void k_rtr_eval2(const AgentDev *agents, const TeamDev *team, int sel, int sp) {
  const AgentDev &ag = agents[sel_cur(team, sel)];
  constexpr int PPB = 64 / R;
  __shared__ double Ysh[PPB * 4 * R], Esh[PPB * 4 * R], Wsh[PPB * 4 * R];
  const int lane = threadIdx.x, lp = lane / R, a = lane - lp * R;
  const int j = blockIdx.x * PPB + lp;
  if (blockIdx.x * PPB >= ag.n) return;
  {
    const RtrState S = ag.st[sp];
    if (S.outer_done || S.tcg_active || S.need_init) return;
  }
  const bool act = lp < PPB && j < ag.n;
  const double *X2 = ag.buf[B_X2], *ETA = ag.buf[B_ETA];
  double fpart = 0, gpart = 0, ge = 0, eh = 0;
  double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, vrow[4] = {0, 0, 0, 0}, hrow[4], eg[4] = {0, 0, 0, 0};
  if (act) {
    spmm_row<R, 2>(ag, j, [&](int i, double(*x)[4]) {
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) {
        x[0][cp] = X2[((size_t)4 * i + cp) * R + a];
        x[1][cp] = ETA[((size_t)4 * i + cp) * R + a];
      }
    }, acc);
    const bool pub = ag.pub_index[j] >= 0;
    const double *G = ag.buf[B_G] + (size_t)j * 4 * R;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const size_t o = ((size_t)4 * j + c) * R + a;
      const double xr = X2[o], g = pub ? G[c * R + a] : 0.0;
      fpart += (0.5 * acc[0][c] + g) * xr;
      eg[c] = acc[0][c] + g;
      ag.buf[B_EGRAD2][o] = eg[c];
      Ysh[lp * 4 * R + c * R + a] = xr;
      Wsh[lp * 4 * R + c * R + a] = eg[c];
    }
  }
  __syncthreads();
  if (act) {
    double o3[3];
    tangent_row<R>(Ysh + lp * 4 * R, Wsh + lp * 4 * R, a, o3);
    double *GF2 = ag.buf[B_GF2] + (size_t)j * 4 * R;
#pragma unroll
    for (int c = 0; c < 3; ++c) { GF2[c * R + a] = o3[c]; gpart += o3[c] * o3[c]; }
    GF2[3 * R + a] = eg[3];
    gpart += eg[3] * eg[3];
  }
  __syncthreads();
  if (act) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const size_t o = ((size_t)4 * j + c) * R + a;
      vrow[c] = ETA[o];
      Ysh[lp * 4 * R + c * R + a] = ag.buf[B_X][o];
      Esh[lp * 4 * R + c * R + a] = ag.buf[B_EGRAD][o];
    }
  }
  __syncthreads();
  hess_tail<R>(Ysh + lp * 4 * R, Esh + lp * 4 * R, Wsh + lp * 4 * R, a, acc[1], vrow, hrow, act);
  if (act) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const size_t o = ((size_t)4 * j + c) * R + a;
      ag.buf[B_HETA][o] = hrow[c];
      ge += ag.buf[B_GF][o] * vrow[c];
      eh += vrow[c] * hrow[c];
    }
  }
  fpart = wave_sum(fpart); gpart = wave_sum(gpart); ge = wave_sum(ge); eh = wave_sum(eh);
  if (lane == 0) {
    double *P = ag.part + PART_C + (size_t)blockIdx.x * PART_STRIDE;
    P[0] = fpart; P[1] = gpart; P[2] = ge; P[3] = eh;
  }
}

// outer step, acceptance test + radius update (ROPTLIB SolversTR constants: accept rho > 0.1,
// grow x2 when rho > 0.75 at the boundary, shrink x0.25 when rho < 0.25)
template <int R>
__global__ void k_rtr_accept(const AgentDev *agents, const TeamDev *team, int sel, int sp, double tol, int max_outer,
                             double max_radius) {
  const AgentDev &ag = agents[sel_cur(team, sel)];
  const RtrState S = ag.st[sp];
  if (S.outer_done || S.tcg_active || S.need_init) {
    if (blockIdx.x == 0 && threadIdx.x == 0) ag.st[sp ^ 1] = S;
    return;
  }
  const int lane = threadIdx.x & 63;
  const int nb = spmm_blocks<R>(ag.n);
  const double f2 = sum_partials(ag.part + PART_C, nb, PART_STRIDE, lane);
  const double g2 = sum_partials(ag.part + PART_C + 1, nb, PART_STRIDE, lane);
  const double ge = sum_partials(ag.part + PART_C + 2, nb, PART_STRIDE, lane);
  const double eh = sum_partials(ag.part + PART_C + 3, nb, PART_STRIDE, lane);
  const double rho = (S.f1 - f2) / (-ge - 0.5 * eh);
  const bool accept = rho > 0.1;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    RtrState T = S;
    if (rho > 0.75) {
      if (S.tcg_status == 1 || S.tcg_status == 2) T.Delta = fmin(2.0 * S.Delta, max_radius);
    } else if (rho < 0.25) {
      T.Delta = 0.25 * S.Delta;
    }
    if (accept) { T.f1 = f2; T.ngf = sqrt(g2); T.accepted = S.accepted + 1; }
    T.hv_count = S.hv_count + 1;
    T.outer_it = S.outer_it + 1;
    T.outer_done = (T.outer_it >= max_outer) || (T.ngf < tol);
    T.tcg_active = 0;
    T.need_init = 1;
    ag.st[sp ^ 1] = T;
  }
  if (!accept) return;
  const size_t len = (size_t)ag.n * 4 * R;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < len; t += (size_t)gridDim.x * blockDim.x) {
    ag.buf[B_X][t] = ag.buf[B_X2][t];
    ag.buf[B_EGRAD][t] = ag.buf[B_EGRAD2][t];
    ag.buf[B_GF][t] = ag.buf[B_GF2][t];
  }
}

// ------------------------------------------------------------------------------------------------
// exchange (a7): packed slabs in public_pose_ids / neighbor_pose_ids order
template <int R>
__global__ void k_pack(const double *X, const int *frames, int count, double *out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * 4 * R) return;
  const int q = t / (4 * R), k = t - q * 4 * R;
  out[t] = X[(size_t)frames[q] * 4 * R + k];
}
template <int R>
__global__ void k_unpack(double *slab, const int *slots, int count, const double *in) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * 4 * R) return;
  const int q = t / (4 * R), k = t - q * 4 * R;
  slab[(size_t)slots[q] * 4 * R + k] = in[t];
}

// per-edge residual sqrt(kappa |Y_j - Y_i R|^2 + tau |p_j - p_i - Y_i t|^2) (a8) and cost partials
template <int R>
__global__ void k_residuals(const AgentDev *agents, int ai) {
  const AgentDev &ag = agents[ai];
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ag.nedges) return;
  const EdgeDev &m = ag.edges[e];
  const double *Xi = m.i_local >= 0 ? ag.buf[B_X] + (size_t)m.i_local * 4 * R : ag.nbr[0] + (size_t)m.i_slot * 4 * R;
  const double *Xj = m.j_local >= 0 ? ag.buf[B_X] + (size_t)m.j_local * 4 * R : ag.nbr[0] + (size_t)m.j_slot * 4 * R;
  double sr = 0, st = 0;
#pragma unroll
  for (int x = 0; x < R; ++x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double v = Xj[c * R + x];
#pragma unroll
      for (int b = 0; b < 3; ++b) v -= Xi[b * R + x] * m.R[3 * b + c];
      sr += v * v;
    }
    double v = Xj[3 * R + x] - Xi[3 * R + x];
#pragma unroll
    for (int b = 0; b < 3; ++b) v -= Xi[b * R + x] * m.t[b];
    st += v * v;
  }
  ag.resid[e] = sqrt(m.kappa * sr + m.tau * st);
}

// scal[5] = sum over owned edges of w/2 * residual^2   (single workgroup, fixed order)
__global__ __launch_bounds__(256) void k_cost(const AgentDev *agents, int ai) {
  const AgentDev &ag = agents[ai];
  __shared__ double red[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  double s = 0;
  for (int e = tid; e < ag.nedges; e += 256) {
    const EdgeDev &m = ag.edges[e];
    if (m.count_in_cost) s += 0.5 * m.weight * ag.resid[e] * ag.resid[e];
  }
  s = wave_sum(s);
  if (lane == 0) red[w] = s;
  __syncthreads();
  if (tid == 0) ag.scal[5] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void k_noop(const AgentDev *agents, int ai) { (void)agents; (void)ai; }

// dense A = Q + shift I from the block-CSR (column-major N4 x N4; A must be zeroed first)
__global__ void k_bsr_to_dense(const int *rowptr, const int *col, const double *qval, int n, double shift, double *A) {
  const int j = blockIdx.x;  // block row = output pose = column block of A
  const int N4 = 4 * n;
  for (int p = rowptr[j] + threadIdx.x / 16; p < rowptr[j + 1]; p += blockDim.x / 16) {
    const int e = threadIdx.x % 16, cp = e % 4, c = e / 4;
    const int i = col[p];
    double v = qval[(size_t)16 * p + e];
    if (i == j && cp == c) v += shift;
    A[(size_t)(4 * j + c) * N4 + 4 * i + cp] = v;
  }
}

// ================================================================================================
// launch wrappers
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

void launch_buildG(const LaunchCtx &c, int sel, int max_npub, int aux, int pull) {
  if (max_npub <= 0) return;
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_buildG<R>, dim3(spmm_grid(c.r, max_npub), c.ny), dim3(64), 0, c.stream, c.agents,
                                          c.team, sel, aux, pull));
}
void launch_pull(const LaunchCtx &c, int dst, int nshared) {
  if (nshared <= 0) return;
  const int len = nshared * 4 * c.r;
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_pull<R>, dim3((len + 255) / 256), dim3(256), 0, c.stream, c.agents, dst));
}
void launch_eval(const LaunchCtx &c, int sel, int max_n, int xb, int egb, int gfb, int poff, const EvalOpts &o) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_eval<R>, dim3(spmm_grid(c.r, max_n), c.ny), dim3(64), 0, c.stream, c.agents,
                                          c.team, sel, xb, egb, gfb, poff, o.gmode, o.aux));
}
void launch_hess(const LaunchCtx &c, int sel, int max_n, int xb, int egb, int vb, int ob, int poff) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_hess<R>, dim3(spmm_grid(c.r, max_n), c.ny), dim3(64), 0, c.stream, c.agents,
                                          c.team, sel, xb, egb, vb, ob, poff));
}
void launch_precond(const LaunchCtx &c, int sel, int max_n, int mode, int xb, int vb, int zb, int sp, int max_inner,
                    double step, int accel, int num_robots, int advance, int restart_interval, int ahead) {
  const int grid = (((4 * max_n + 7) / 8) + 7) / 8 * 8;  // multiple of 8: see the XCD-aware block order in k_precond
#define PC_CALL(M)                                                                                                  \
  if (4 * max_n > 1024 && 4 * max_n <= 2048) {                                                                       \
    DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL((k_precond<R, M, 2048>), dim3(grid, c.ny), dim3(256), 0, c.stream, c.agents,   \
                                            c.team, sel, xb, vb, zb, sp, max_inner, step, accel, num_robots, advance,   \
                                            restart_interval, ahead));                                                \
  } else {                                                                                                           \
    DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL((k_precond<R, M, 1024>), dim3(grid, c.ny), dim3(256), 0, c.stream, c.agents,   \
                                            c.team, sel, xb, vb, zb, sp, max_inner, step, accel, num_robots, advance,   \
                                            restart_interval, ahead));                                                \
  }
  if (mode == PM_PLAIN_) { PC_CALL(PM_PLAIN_); }
  else if (mode == PM_TCG_INIT_) { PC_CALL(PM_TCG_INIT_); }
  else if (mode == PM_TCG_STEP_) { PC_CALL(PM_TCG_STEP_); }
  else { PC_CALL(PM_RGD_); }
#undef PC_CALL
}
void launch_eval_stats(const LaunchCtx &c, int max_n, int first, int has_eval, int has_stats, int num_robots,
                       int restart_interval) {
  const int nb = spmm_grid(c.r, max_n);
  const int grid = nb * ((has_eval ? 1 : 0) + (has_stats ? 1 : 0)) + 1;
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_eval_stats<R>, dim3(grid), dim3(64), 0, c.stream, c.agents, c.team, nb, first,
                                          has_eval, has_stats, num_robots, restart_interval));
}
void launch_tcg_hv(const LaunchCtx &c, int sel, int max_n, int sp, int max_inner) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_tcg_hv<R>, dim3(spmm_grid(c.r, max_n), c.ny), dim3(64), 0, c.stream, c.agents,
                                          c.team, sel, sp, max_inner));
}
void launch_retract(const LaunchCtx &c, int sel, int max_n, int xb, int eb, double scale, int ob, int guard_state) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_retract<R>, dim3((max_n + 63) / 64, c.ny), dim3(64), 0, c.stream, c.agents,
                                          c.team, sel, xb, eb, scale, ob, guard_state));
}
void launch_project_raw(const LaunchCtx &c, const double *X, double *out, int n) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL((k_raw_op<R, 0>), dim3((n + 63) / 64), dim3(64), 0, c.stream, X, X, out, n));
}
void launch_tangent_raw(const LaunchCtx &c, const double *X, const double *V, double *out, int n) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL((k_raw_op<R, 1>), dim3((n + 63) / 64), dim3(64), 0, c.stream, X, V, out, n));
}
void launch_retract_raw(const LaunchCtx &c, const double *X, const double *E, double *out, int n) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL((k_raw_op<R, 2>), dim3((n + 63) / 64), dim3(64), 0, c.stream, X, E, out, n));
}
void launch_nest_pre(const LaunchCtx &c, int sel, int only_agent, int num_agents, int max_n, int num_robots,
                     int restart_interval) {
  dim3 grid((max_n + 63) / 64, only_agent >= 0 ? 1 : num_agents);
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_nest_pre<R>, grid, dim3(64), 0, c.stream, c.agents, c.team, sel, only_agent,
                                          num_robots, restart_interval));
}
void launch_stats_nest(const LaunchCtx &c, int num_agents, int max_n, int num_robots, int restart_interval) {
  const int nest_tiles = (max_n + 63) / 64;
  const int grid = nest_tiles * num_agents + spmm_grid(c.r, max_n);
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_stats_nest<R>, dim3(grid), dim3(64), 0, c.stream, c.agents, c.team, nest_tiles,
                                          num_agents, num_robots, restart_interval));
}
void launch_nest_post(const LaunchCtx &c, int sel, int max_n, int num_robots, int restart_interval) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_nest_post<R>, dim3((max_n + 63) / 64), dim3(64), 0, c.stream, c.agents,
                                          c.team, sel, num_robots, restart_interval));
}
void launch_nest_reset(const LaunchCtx &c, int sel, int max_n) {
  const int len = max_n * 4 * c.r;
  hipLaunchKernelGGL(k_nest_reset, dim3((len + 255) / 256), dim3(256), 0, c.stream, c.agents, c.team, sel, c.r);
}
void launch_advance(const LaunchCtx &c, int only_agent, int num_agents, int accel, int num_robots, int restart_interval,
                    int bump_team, int inc, int team_inc) {
  hipLaunchKernelGGL(k_advance, dim3(only_agent >= 0 ? 1 : num_agents), dim3(64), 0, c.stream, c.agents, c.team,
                     only_agent, accel, num_robots, restart_interval, bump_team, inc, team_inc < 0 ? inc : team_inc);
}
void launch_status(const LaunchCtx &c, int sel, int only_agent, int num_agents, int max_n) {
  dim3 grid((max_n + 63) / 64, (sel == -3 && only_agent < 0) ? num_agents : 1);
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_status<R>, grid, dim3(64), 0, c.stream, c.agents, c.team, sel, only_agent));
}
void launch_rtr_begin(const LaunchCtx &c, int sel, double Delta0, double tol, int max_outer) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_rtr_begin<R>, dim3(1, c.ny), dim3(64), 0, c.stream, c.agents, c.team, sel, Delta0,
                                          tol, max_outer));
}
void launch_rtr_eval2(const LaunchCtx &c, int sel, int max_n, int sp) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_rtr_eval2<R>, dim3(spmm_grid(c.r, max_n), c.ny), dim3(64), 0, c.stream, c.agents,
                                          c.team, sel, sp));
}
void launch_rtr_accept(const LaunchCtx &c, int sel, int max_n, int sp, double tol, int max_outer, double max_radius) {
  const int len = max_n * 4 * c.r;
  int grid = (len + 255) / 256;
  if (grid > 64) grid = 64;
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_rtr_accept<R>, dim3(grid, c.ny), dim3(256), 0, c.stream, c.agents, c.team, sel,
                                          sp, tol, max_outer, max_radius));
}
void launch_pack(const LaunchCtx &c, const double *X, const int *frames, int count, double *out) {
  if (count <= 0) return;
  const int len = count * 4 * c.r;
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_pack<R>, dim3((len + 255) / 256), dim3(256), 0, c.stream, X, frames, count,
                                          out));
}
void launch_unpack(const LaunchCtx &c, double *slab, const int *slots, int count, const double *in) {
  if (count <= 0) return;
  const int len = count * 4 * c.r;
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_unpack<R>, dim3((len + 255) / 256), dim3(256), 0, c.stream, slab, slots,
                                          count, in));
}
void launch_residuals(const LaunchCtx &c, int ai, int nedges) {
  if (nedges <= 0) return;
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_residuals<R>, dim3((nedges + 63) / 64), dim3(64), 0, c.stream, c.agents, ai));
}
void launch_noop(const LaunchCtx &c, int grid, int block) {
  hipLaunchKernelGGL(k_noop, dim3(grid), dim3(block), 0, c.stream, c.agents, 0);
}
void launch_cost(const LaunchCtx &c, int ai) {
  hipLaunchKernelGGL(k_cost, dim3(1), dim3(256), 0, c.stream, c.agents, ai);
}
void launch_copy(const LaunchCtx &c, int sel, int only_agent, int num_agents, int max_n, int from, int to, int publish) {
  const int len = max_n * 4 * c.r;
  dim3 grid((len + 255) / 256, (sel == -3 && only_agent < 0) ? num_agents : 1);
  hipLaunchKernelGGL(k_copy, grid, dim3(256), 0, c.stream, c.agents, c.team, sel, only_agent, c.r, from, to, publish);
}
void launch_bsr_to_dense(hipStream_t s, const int *rowptr, const int *col, const double *qval, int n, double shift,
                         double *A) {
  (void)hipMemsetAsync(A, 0, sizeof(double) * (size_t)16 * n * n, s);
  hipLaunchKernelGGL(k_bsr_to_dense, dim3(n), dim3(64), 0, s, rowptr, col, qval, n, shift, A);
}

}  // namespace dpgo
