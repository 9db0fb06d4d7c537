// linesearch.hip -- Riemannian gradient descent with a backtracking (Armijo) line search on the retraction curve
// (north_star "RTR/RGD line search"; SURVEY App. B: "a backtracking variant exists" [UPSTREAM-RECALL]; the wrapper's
// knobs of the gradient method are src/PGOAgentROSNode.cpp:86-97, launch/PGOAgent.launch:16-17).
//
// A CPU line search evaluates one trial point after the other and stops at the first that passes.  Here the trial steps
// t_j = stepsize * shrink^j are known in advance, so ALL trial points are formed at once (k_ls_trials: J retractions per
// pose, one lane per pose) and ALL their costs come from ONE pass over the sparse operator with J right-hand sides
// (k_ls_cost: four trial points share every 4 x 4 block load; the operator is 0.3 MB and L2 resident, the cost of a pass
// is its dependent round trips, not its bytes).  The decision -- the first j with sufficient decrease, exactly the
// sequential rule -- is re-derived by every workgroup of k_ls_apply from the same partial sums in the same order (the
// idiom of the whole solver: no scalar visits the host, no atomics, bitwise reproducible), which then moves the chosen
// trial point into X.  Three launches whatever the number of back-offs; the host enqueues them blindly (graph-capturable).
#include "kernel_common.h"

namespace dpgo {

// the trial costs of one tile occupy one block of partials (PART_A[tile][0 .. J)), the slope PART_C[tile][2]
static_assert(LS_MAX_TRIALS <= PART_STRIDE, "a tile's trial costs must fit one block of partial sums");

// trial points live in work vectors that only the trust-region solve uses
__host__ __device__ __forceinline__ int ls_buf(int j) {
  switch (j) {
    case 0: return B_X2;
    case 1: return B_T0;
    case 2: return B_T1;
    case 3: return B_T2;
    case 4: return B_ETA;
    case 5: return B_R0;
    case 6: return B_R1;
    default: return B_D0;
  }
}

// X_j = Retr_X(-t_j d), j < ntrials: qf(Y + eta) on the rotation block, p + eta on the translation (a5)
template <int R>
__global__ __launch_bounds__(64) void k_ls_trials(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, int dirb,
                                                  double step0, double shrink, int ntrials) {
  const AgentDev &ag = agents[sel_cur(team, sel)];
  const int j0 = blockIdx.x * 64, tid = threadIdx.x;
  if (j0 >= ag.n) return;
  const int cnt = min(64, ag.n - j0);
  __shared__ Tile<R> TA, TB;
  tile_in<R>(TA, ag.buf[B_X], j0, cnt, tid);
  tile_in<R>(TB, ag.buf[dirb], j0, cnt, tid);
  __syncthreads();
  double x[4 * R], e[4 * R];
  tile_get<R>(TA, tid < cnt ? tid : 0, x);
  tile_get<R>(TB, tid < cnt ? tid : 0, e);
  double step = step0;
  for (int j = 0; j < ntrials; ++j) {
    double y[4 * R];
#pragma unroll
    for (int i = 0; i < 4 * R; ++i) y[i] = x[i] + (-step) * e[i];
    qf_inplace<R>(y);
    __syncthreads();  // (the previous trial point has left the tile)
    if (tid < cnt) tile_put<R>(TA, tid, y);
    __syncthreads();
    tile_out<R>(TA, ag.buf[ls_buf(j)], j0, cnt, tid);
    step *= shrink;
  }
}

// f(X_j) for every trial point and the slope <grad f(X), d>.  One lane per (pose, row a) as in k_eval.
// partials: PART_A[block][j] = f(X_j) share, PART_C[block][2] = slope share ([0], [1] hold f(X), |grad|^2 of the
// evaluation in front)
template <int R>
__global__ __launch_bounds__(64) void k_ls_cost(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, int dirb,
                                                int ntrials) {
  const AgentDev &ag = agents[sel_cur(team, sel)];
  constexpr int PPB = 64 / R;
  const int lane = threadIdx.x, lp = lane / R, a = lane - lp * R;
  const int j = blockIdx.x * PPB + lp;
  if (blockIdx.x * PPB >= ag.n) return;
  const bool act = lp < PPB && j < ag.n;
  double f[8] = {0, 0, 0, 0, 0, 0, 0, 0}, slope = 0;
  if (act) {
    double g[4] = {0, 0, 0, 0};
    if (gp(ag.pub_index)[j] >= 0) {
      const double *G = ag.buf[B_G] + (size_t)j * 4 * R;
#pragma unroll
      for (int c = 0; c < 4; ++c) g[c] = gp(G)[c * R + a];
    }
    for (int t0 = 0; t0 < ntrials; t0 += 4) {
      const double *Xt[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) Xt[v] = ag.buf[ls_buf(min(t0 + v, ntrials - 1))];
      double acc[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
      spmm_row<R, 4>(ag, j, [&](int i, double(*x)[4]) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int cp = 0; cp < 4; ++cp) x[v][cp] = gp(Xt[v])[((size_t)4 * i + cp) * R + a];
      }, acc);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        double s = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) s += (0.5 * acc[v][c] + g[c]) * gp(Xt[v])[((size_t)4 * j + c) * R + a];
        if (t0 + v < ntrials) f[t0 + v] = s;
      }
    }
    const double *GF = ag.buf[B_GF] + (size_t)j * 4 * R, *D = ag.buf[dirb] + (size_t)j * 4 * R;
#pragma unroll
    for (int c = 0; c < 4; ++c) slope += gp(GF)[c * R + a] * gp(D)[c * R + a];
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) f[t] = wave_sum(f[t]);
  slope = wave_sum(slope);
  if (lane == 0) {
    double *P = ag.part + PART_A + (size_t)blockIdx.x * PART_STRIDE;
#pragma unroll
    for (int t = 0; t < 8; ++t) gp(P)[t] = f[t];
    gp(ag.part)[PART_C + (size_t)blockIdx.x * PART_STRIDE + 2] = slope;
  }
}

// the decision and the move.  Every workgroup sums the same partials in the same order: the first j with
// f(X_j) <= f(X) - sigma t_j slope; X <- X_j (none: X stays).  Workgroup 0 leaves the record in the agent's scalars:
// [8] back-offs taken (ntrials: none qualified), [9] accepted, [10] f at the accepted point, [11] the accepted step.
// tail != 0 (the team schedule's non-restart iterations): the rest of the iteration rides in this launch, one lane per
// pose of the tile -- bit 1: the Nesterov sequence V <- proj(V + gamma (X - Y)) (k_nest_post; gamma as k_nest_pre
// published it in scal[6]); always: |X - XPrev|^2 of the tile into PART_D / PART_E (k_status, opt = 1); workgroup 0: the
// Nesterov scalars of every agent and the team's iteration counter advance (k_advance) -- three launches less.
template <int R>
__global__ __launch_bounds__(64) void k_ls_apply(const AgentDev *__restrict__ agents, TeamDev *team, int sel, double step0,
                                                 double shrink, double sigma, int ntrials, int tail, int num_robots,
                                                 int restart_interval) {
  const AgentDev &ag = agents[sel_cur(team, sel)];
  const int j0 = blockIdx.x * 64, tid = threadIdx.x;
  if (j0 >= ag.n) return;
  const int nb = spmm_blocks<R>(ag.n);
  double f0 = 0, sl = 0, fj[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int base = 0; base < nb; base += 64) {
    const int b = base + tid;
    const bool in = b < nb;
    const double *pc = ag.part + PART_C + (size_t)(in ? b : 0) * PART_STRIDE;
    const double *pa = ag.part + PART_A + (size_t)(in ? b : 0) * PART_STRIDE;
    const double w = in ? 1.0 : 0.0;
    double v[8];
    const double c0 = gp(pc)[0], c2 = gp(pc)[2];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = gp(pa)[t];
    f0 += w * c0; sl += w * c2;
#pragma unroll
    for (int t = 0; t < 8; ++t) fj[t] += w * v[t];
  }
  f0 = wave_sum(f0); sl = wave_sum(sl);
#pragma unroll
  for (int t = 0; t < 8; ++t) fj[t] = wave_sum(fj[t]);
  int chosen = -1;
  double step = step0, fsel = f0, tsel = 0;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t < ntrials && chosen < 0 && fj[t] <= f0 - sigma * step * sl) { chosen = t; fsel = fj[t]; tsel = step; }
    step *= shrink;
  }
  if (blockIdx.x == 0 && tid == 0) {
    ag.scal[8] = (double)(chosen < 0 ? ntrials : chosen);
    ag.scal[9] = chosen < 0 ? 0.0 : 1.0;
    ag.scal[10] = fsel;
    ag.scal[11] = tsel;
  }
  const int cnt = min(64, ag.n - j0);
  if (!tail) {
    if (chosen < 0) return;
    const size_t lo = (size_t)j0 * 4 * R, len = (size_t)cnt * 4 * R;
    const double *src = ag.buf[ls_buf(chosen)] + lo;
    double *dst = ag.buf[B_X] + lo;
    double tmp[4 * R];
#pragma unroll
    for (int k = 0; k < 4 * R; ++k) { const size_t e = tid + 64 * (size_t)k; tmp[k] = (e < len) ? src[e] : 0.0; }
#pragma unroll
    for (int k = 0; k < 4 * R; ++k) { const size_t e = tid + 64 * (size_t)k; if (e < len) dst[e] = tmp[k]; }
    return;
  }
  __shared__ Tile<R> TX, TV, TY, TP;
  tile_in<R>(TX, ag.buf[chosen < 0 ? B_X : ls_buf(chosen)], j0, cnt, tid);
  tile_in<R>(TP, ag.buf[B_XPREV], j0, cnt, tid);
  if (tail & 2) {
    tile_in<R>(TV, ag.buf[B_V], j0, cnt, tid);
    tile_in<R>(TY, ag.buf[B_Y], j0, cnt, tid);
  }
  __syncthreads();
  if (chosen >= 0) tile_out<R>(TX, ag.buf[B_X], j0, cnt, tid);
  double rel = 0;
  if (tid < cnt) {
    double x[4 * R], q[4 * R];
    tile_get<R>(TX, tid, x);
    tile_get<R>(TP, tid, q);
#pragma unroll
    for (int i = 0; i < 4 * R; ++i) { const double d = x[i] - q[i]; rel += d * d; }
    if (tail & 2) {
      const double gamma = ag.scal[6];
      double v[4 * R];
      tile_get<R>(TV, tid, v);
      tile_get<R>(TY, tid, q);
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) v[i] += gamma * (x[i] - q[i]);
      polar_inplace<R>(v);
      tile_put<R>(TV, tid, v);
    }
  }
  rel = wave_sum(rel);
  if (tid == 0) {
    ag.part[PART_D + (size_t)blockIdx.x * PART_STRIDE] = rel;
    ag.part[PART_E + (size_t)blockIdx.x * PART_STRIDE] = rel;
  }
  if (tail & 2) {
    __syncthreads();
    tile_out<R>(TV, ag.buf[B_V], j0, cnt, tid);
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    for (int k = 0; k < team->num_agents; ++k) advance_agent(agents[k], (tail & 2) ? 1 : 0, num_robots, restart_interval);
    team->iter += 1;
  }
}

void launch_ls_trials(const LaunchCtx &c, int sel, int max_n, int dirb, double step0, double shrink, int ntrials) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_ls_trials<R>, dim3((max_n + 63) / 64, c.ny), dim3(64), 0, c.stream, c.agents, c.team, sel,
                                          dirb, step0, shrink, ntrials));
}

void launch_ls_cost(const LaunchCtx &c, int sel, int max_n, int dirb, int ntrials) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_ls_cost<R>, dim3(spmm_grid(c.r, max_n), c.ny), dim3(64), 0, c.stream, c.agents, c.team,
                                          sel, dirb, ntrials));
}

void launch_ls_apply(const LaunchCtx &c, int sel, int max_n, double step0, double shrink, double sigma, int ntrials, int tail,
                     int num_robots, int restart_interval) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_ls_apply<R>, dim3((max_n + 63) / 64, c.ny), dim3(64), 0, c.stream, c.agents, c.team, sel,
                                          step0, shrink, sigma, ntrials, tail, num_robots, restart_interval));
}

}  // namespace dpgo
