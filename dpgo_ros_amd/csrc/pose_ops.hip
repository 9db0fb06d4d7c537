// pose_ops.hip -- per-pose manifold kernels (retraction, raw manifold operations, Nesterov sequences; SURVEY 8a rows
// a5, a6), status partials, the scalar trust-region state machine (a4), public-pose pack / unpack (a7), per-edge
// residuals and the global cost (a8), dense assembly of Q + shift I.
#include <cstdlib>

#include "kernel_common.h"

namespace dpgo {

// out = Retr_x(scale * eta).  guard_state >= 0: skip when the trust-region state says done.
template <int R>
__global__ __launch_bounds__(64) void k_retract(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, int xb, int eb,
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
    if (OP == 0) { polar_inplace<R, true>(x); tile_put<R>(TA, tid, x); }
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

// staged neighbour poses, pinned host memory -> slabs: elements first, first + stride, ... of the (n0 + n1) * 4R doubles.
// Reads from host memory cross PCIe (~2 us each): eight per lane are in flight before the first is stored (a plain loop
// would wait for every one in turn).
template <int R>
__device__ __forceinline__ void upload_slice(const AgentDev &ag, const int *slots, const double *in, int n0, int n1, int first,
                                             int stride) {
  const int total = (n0 + n1) * 4 * R;
  for (int base = first; base < total; base += 8 * stride) {
    double v[8];
    int sl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = min(base + u * stride, total - 1);
      v[u] = in[t];
      sl[u] = slots[t / (4 * R)];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int t = base + u * stride;
      if (t < total) {
        const int q = t / (4 * R), k = t - q * 4 * R;
        ag.nbr[q < n0 ? 0 : 1][(size_t)sl[u] * 4 * R + k] = v[u];
      }
    }
  }
}

// The same scatter with the Nesterov step of the launch BETWEEN its loads and its stores: the first (and, up to 8 poses per
// lane, only) trip's reads cross PCIe while the step runs instead of in front of it.
template <int R>
struct UploadTrip { double v[8]; int sl[8]; };

template <int R>
__device__ __forceinline__ void upload_request(const int *slots, const double *in, int n0, int n1, int first, int stride,
                                               UploadTrip<R> &tr) {
  const int total = (n0 + n1) * 4 * R;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int t = min(first + u * stride, total - 1);
    tr.v[u] = in[t];
    tr.sl[u] = slots[t / (4 * R)];
  }
}

template <int R>
__device__ __forceinline__ void upload_commit(const AgentDev &ag, const int *slots, const double *in, int n0, int n1, int first,
                                              int stride, const UploadTrip<R> &tr) {
  const int total = (n0 + n1) * 4 * R;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int t = first + u * stride;
    if (t < total) {
      const int q = t / (4 * R), k = t - q * 4 * R;
      ag.nbr[q < n0 ? 0 : 1][(size_t)tr.sl[u] * 4 * R + k] = tr.v[u];
    }
  }
  upload_slice<R>(ag, slots, in, n0, n1, first + 8 * stride, stride);  // (more than 8 per lane: the rest in turn)
}

template <int R>
__global__ __launch_bounds__(64) void k_nest_pre(const AgentDev *__restrict__ agents, TeamDev *team, int sel, int only_agent,
                                                 int num_robots, int restart_interval, int fused_restart, const int *up_slots,
                                                 const double *up_in, int up_n0, int up_n1) {
  __shared__ Tile<R> TX, TV;
  // (per-agent API, iterate(true): the neighbour poses staged on the host since the agent's last block update are
  // scattered into its slabs here -- nothing in this launch reads them, the evaluation behind it does)
  const bool up = up_n0 + up_n1 > 0 && only_agent >= 0;
  const int first = (int)blockIdx.x * 64 + (int)threadIdx.x, stride = 64 * (int)gridDim.x;
  UploadTrip<R> tr;
  bool requested = false;
  auto request = [&]() {
    if (up) { upload_request<R>(up_slots, up_in, up_n0, up_n1, first, stride, tr); requested = true; }
  };
  nest_pre_body<R>(agents, team, sel, only_agent, num_robots, restart_interval, (int)blockIdx.x, (int)blockIdx.y, TX, TV,
                   fused_restart, request);
  if (up) {
    if (requested) upload_commit<R>(agents[only_agent], up_slots, up_in, up_n0, up_n1, first, stride, tr);
    else upload_slice<R>(agents[only_agent], up_slots, up_in, up_n0, up_n1, first, stride);  // (a tile past the agent's poses)
  }
}

// after the selected agent's local solve (unfused path):  V = proj(V + gamma' (X - Y)); on restart
// X = XPrev (the host then re-optimizes from XPrev and calls k_nest_reset).
template <int R>
__global__ __launch_bounds__(64) void k_nest_post(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, int num_robots,
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
__global__ void k_nest_reset(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, int r) {
  const AgentDev &ag = agents[sel_cur(team, sel)];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ag.N4 * r) return;
  const double x = ag.buf[B_X][t];
  ag.buf[B_V][t] = x;
  ag.buf[B_Y][t] = x;
}

// end of an iteration (unfused paths): advance gamma/alpha/iter of every agent, and the team counter
__global__ void k_advance(const AgentDev *__restrict__ agents, TeamDev *team, int only_agent, int accel, int num_robots,
                          int restart_interval, int bump_team, int inc, int team_inc) {
  const int ai = only_agent >= 0 ? only_agent : (int)blockIdx.x;
  if (threadIdx.x != 0) return;
  advance_agent(agents[ai], accel, num_robots, restart_interval, inc);
  if (bump_team && ai == 0) team->iter += team_inc;
}

// PART_D partial [0] = |X - XPrev|_F^2 over a 64-pose tile (blockIdx.y = agent when sel == -3)
template <int R>
__global__ __launch_bounds__(64) void k_status(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, int only_agent, int opt) {
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
  if (tid == 0) {
    ag.part[PART_D + (size_t)blockIdx.x * PART_STRIDE] = s;
    if (opt) ag.part[PART_E + (size_t)blockIdx.x * PART_STRIDE] = s;  // status of the last block update (a9)
  }
}

// buf[to] = buf[from] for one agent or (sel == -3) every agent (blockIdx.y).  As the first kernel of a
// non-accelerated iteration (publish != 0) it also publishes team->cur_sel.
__global__ void k_copy(const AgentDev *__restrict__ agents, TeamDev *team, int sel, int only_agent, int r, int from, int to,
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
__global__ void k_rtr_begin(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, double Delta0, double tol,
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

// outer step, acceptance test + radius update (ROPTLIB SolversTR constants: accept rho > 0.1,
// grow x2 when rho > 0.75 at the boundary, shrink x0.25 when rho < 0.25)
template <int R>
__global__ void k_rtr_accept(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, int sp, double tol, int max_outer,
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
    if (S.outer_it < 4) T.tcg_o[S.outer_it] = S.tcg_j + 1;
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

// both sequences of every listed pose in one launch: out = [X of frames[0..count) | Y of the same frames]
template <int R>
__global__ void k_pack2(const double *X, const double *Y, const int *frames, int count, double *out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * 4 * R) return;
  const int q = t / (4 * R), k = t - q * 4 * R;
  const double *src = blockIdx.y ? Y : X;
  out[(size_t)blockIdx.y * count * 4 * R + t] = src[(size_t)frames[q] * 4 * R + k];
}

template <int R>
__global__ void k_unpack(double *slab, const int *slots, int count, const double *in) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count * 4 * R) return;
  const int q = t / (4 * R), k = t - q * 4 * R;
  slab[(size_t)slots[q] * 4 * R + k] = in[t];
}

// the slabs of one batch of rank-to-rank messages (rank_exchange.cpp): job blockIdx.y packs (UNPACK: scatters) count[j]
// poses between a pose array / neighbour slab and its place in the staging buffer of the message
template <int R, bool UNPACK>
__global__ void k_xfer_multi(XferSegs sg) {
  const int j = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= sg.count[j] * 4 * R) return;
  const int q = t / (4 * R), k = t - q * 4 * R;
  if (UNPACK) sg.src[j][(size_t)sg.idx[j][q] * 4 * R + k] = sg.buf[j][t];
  else sg.buf[j][t] = sg.src[j][(size_t)sg.idx[j][q] * 4 * R + k];
}

// ---- the host boundary of the per-agent API (the path a ROS wrapper drives), without copy engines or stream-wide waits:
// neighbour poses staged by updateNeighborPoses are scattered into the slabs straight FROM pinned host memory
// (slots / in: host pointers; counts[2]: poses of the main / auxiliary sequence, in that order)
template <int R>
__global__ void k_upload2(double *slab0, double *slab1, const int *slots, const double *in, int n0, int n1) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (n0 + n1) * 4 * R) return;
  const int q = t / (4 * R), k = t - q * 4 * R;
  double *slab = q < n0 ? slab0 : slab1;
  slab[(size_t)slots[q] * 4 * R + k] = in[t];
}

// ... and everything the wrapper asks for right after an iterate goes straight INTO pinned host memory, followed by a
// sequence word the host polls (one workgroup: its own stores are ordered by one system-scope fence):
//   out[0]            sequence number (written last)
//   out[1]            sum of the `stat_cnt` status partials part[stat_off + k * stat_stride]   (|X - XPrev|^2)
//   out[2 .. 5]       f_init, |grad|^2_init (PART_C), f_opt, |grad|^2_opt (PART_A) over `opt_nb` blocks
//   out[8 ..]         [X of the public frames | Y of the same frames], the layout of k_pack2
template <int R>
__global__ __launch_bounds__(512) void k_report(const AgentDev *__restrict__ agents, int ai, const int *frames, int count,
                                                  double *out, int stat_off, int stat_cnt, int stat_stride, int opt_nb,
                                                  unsigned long long *seq, int advance, int accel, int num_robots,
                                                  int restart_interval, const int *up_slots, const double *up_in, int up_n0,
                                                  int up_n1, int one_seq) {
  const AgentDev &ag = agents[ai];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // neighbour poses staged on the host that no launch of this iterate needed (iterate(false)): scattered here, one
  // launch less (what k_upload2 does)
  upload_slice<R>(ag, up_slots, up_in, up_n0, up_n1, tid, 512);
  // end of the iterate for this agent (what k_advance does in a launch of its own): nothing below reads the NestState
  if (advance && tid == 511) advance_agent(ag, accel, num_robots, restart_interval);
  const int len = count * 4 * R;
  // (one_seq: X == Y at the public poses -- an accelerated iterate(false) leaves X = Y --, so half the payload crosses
  // the bus and the host duplicates it)
  for (int t = tid; t < (one_seq ? 1 : 2) * len; t += 512) {
    const int sq = t >= len, u = t - sq * len, q = u / (4 * R), k = u - q * 4 * R;
    out[8 + t] = ag.buf[sq ? B_Y : B_X][(size_t)frames[q] * 4 * R + k];
  }
  if (wave == 0 && stat_cnt > 0) {
    const double s = sum_partials(ag.part + stat_off, stat_cnt, stat_stride, lane);
    if (lane == 0) out[1] = s;
  }
  if (wave >= 1 && wave <= 4 && opt_nb > 0) {
    const int w = wave - 1;  // 0 f_init, 1 g2_init, 2 f_opt, 3 g2_opt
    const double s = sum_partials(ag.part + ((w < 2) ? PART_C : PART_A) + (w & 1), opt_nb, PART_STRIDE, lane);
    if (lane == 0) out[2 + w] = s;
  }
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    const unsigned long long v = *seq + 1ull;
    *seq = v;
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(out), v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// The whole accelerated iterate(false) of one agent in ONE launch (the wrapper makes one such call per robot and
// iteration, and every launch in front of the report costs ~3-5 us of the host's wait): the Nesterov step of k_nest_pre
// on 64-pose tiles; every tile also scatters its share of the staged neighbour poses (read from pinned host memory) and
// writes the new X (= Y) of its PUBLIC poses straight into the pinned report (pubpos: where each public pose goes, one
// place per neighbour that shares it), makes that visible system-wide and takes a ticket; the workgroup that draws the
// last ticket advances the agent's Nesterov scalars and writes the sequence word the host polls.
template <int R>
__global__ __launch_bounds__(64) void k_iterate_false(const AgentDev *__restrict__ agents, TeamDev *team, int ai, int num_robots,
                                                      int restart_interval, const int *pubpos_ptr, const int *pubpos, double *out,
                                                      unsigned long long *seq, unsigned long long *ticket, const int *up_slots,
                                                      const double *up_in, int up_n0, int up_n1) {
  __shared__ Tile<R> TX, TV;
  const AgentDev &ag = agents[ai];
  const int tid = threadIdx.x, j0 = (int)blockIdx.x * 64;
  // this tile's share of the staged neighbour poses (nothing in this launch reads them)
  upload_slice<R>(ag, up_slots, up_in, up_n0, up_n1, (int)blockIdx.x * 64 + tid, 64 * (int)gridDim.x);
  int q = -1, p0 = 0, p1 = 0;
  if (j0 < ag.n && tid < min(64, ag.n - j0)) {
    q = ag.pub_index[j0 + tid];
    if (q >= 0) { p0 = pubpos_ptr[q]; p1 = pubpos_ptr[q + 1]; }
  }
  nest_pre_body<R>(agents, team, -2, ai, num_robots, restart_interval, (int)blockIdx.x, 0, TX, TV, 0);
  if (q >= 0) {
    double v[4 * R];
    tile_get<R>(TX, tid, v);  // the tile still holds what went to X (and Y)
    for (int p = p0; p < p1; ++p) {
      double *dst = out + 8 + (size_t)pubpos[p] * 4 * R;
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) dst[i] = v[i];
    }
  }
  __threadfence_system();
  __syncthreads();  // (one 64-thread workgroup is one wavefront on gfx950; the barrier keeps the ticket behind every lane's
                    // stores whatever the wave size)
  unsigned int tk = 0;
  if (tid == 0) tk = (unsigned int)__hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  tk = (unsigned int)__builtin_amdgcn_readfirstlane((int)tk);
  if (tk + 1u != gridDim.x) return;
  if (tid == 0) {
    __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    advance_agent(ag, 1, num_robots, restart_interval);
    const unsigned long long v = *seq + 1ull;
    *seq = v;
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(out), v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// per-edge residual sqrt(kappa |Y_j - Y_i R|^2 + tau |p_j - p_i - Y_i t|^2) (a8) and cost partials
template <int R>
__global__ void k_residuals(const AgentDev *__restrict__ agents, int ai) {
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
__global__ __launch_bounds__(256) void k_cost(const AgentDev *__restrict__ agents, int ai) {
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

__global__ void k_noop(const AgentDev *__restrict__ agents, int ai) { (void)agents; (void)ai; }

// ---- the UPDATE token across processes, on the device (dpgo_team_run_peer, capi.hip): every team owns a mailbox of
// 64-bit words that the OTHER teams write over peer access (HIP IPC: xGMI stores between GPUs) and that its own wait
// kernels poll in local memory.  A signal kernel runs behind the launches whose results it announces (their stores are
// device-visible at the kernel boundary); a wait kernel runs in front of the launches that read a peer's arrays in
// place, or that overwrite what a peer was reading.  Monotonic values, wall-clock time-out raised in a pinned host word.
__global__ void k_mail_signal(MailSignals s) {
  const int i = threadIdx.x;
  if (i >= s.count) return;
  __threadfence_system();
  __hip_atomic_store(s.word[i], s.value[i], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// timeout_ticks of the 100 MHz wall clock (DPGO_MAIL_TIMEOUT_S, default 20 s: a peer rank may be busy with dense
// inversions, graph instantiation or its first code-object load when this rank starts to wait)
__global__ void k_mail_wait(const unsigned long long *mail, MailWaits w, int *err, long long timeout_ticks) {
  const int i = threadIdx.x;
  if (i >= w.count) return;
  const long long t0 = (long long)wall_clock64();
  while (__hip_atomic_load(mail + w.index[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < w.value[i]) {
    __builtin_amdgcn_s_sleep(2);
    if ((long long)wall_clock64() - t0 > timeout_ticks) { *err = 4; break; }
  }
}

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
                     int restart_interval, int fused_restart) {
  dim3 grid((max_n + 63) / 64, only_agent >= 0 ? 1 : num_agents);
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_nest_pre<R>, grid, dim3(64), 0, c.stream, c.agents, c.team, sel, only_agent,
                                          num_robots, restart_interval, fused_restart, c.up_slots, c.up_in, c.up_n0, c.up_n1));
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

void launch_status(const LaunchCtx &c, int sel, int only_agent, int num_agents, int max_n, int opt) {
  dim3 grid((max_n + 63) / 64, (sel == -3 && only_agent < 0) ? num_agents : ((sel <= SEL_GROUP0 && only_agent < 0) ? c.ny : 1));
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_status<R>, grid, dim3(64), 0, c.stream, c.agents, c.team, sel, only_agent, opt));
}

void launch_rtr_begin(const LaunchCtx &c, int sel, double Delta0, double tol, int max_outer) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_rtr_begin<R>, dim3(1, c.ny), dim3(64), 0, c.stream, c.agents, c.team, sel, Delta0,
                                          tol, max_outer));
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

void launch_pack2(const LaunchCtx &c, const double *X, const double *Y, const int *frames, int count, double *out) {
  if (count <= 0) return;
  const int len = count * 4 * c.r;
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_pack2<R>, dim3((len + 255) / 256, 2), dim3(256), 0, c.stream, X, Y, frames, count,
                                          out));
}

void launch_unpack(const LaunchCtx &c, double *slab, const int *slots, int count, const double *in) {
  if (count <= 0) return;
  const int len = count * 4 * c.r;
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_unpack<R>, dim3((len + 255) / 256), dim3(256), 0, c.stream, slab, slots,
                                          count, in));
}

static int xfer_max_len(const XferSegs &sg, int r) {
  int m = 0;
  for (int j = 0; j < sg.n; ++j) m = std::max(m, sg.count[j] * 4 * r);
  return m;
}

void launch_pack_multi(const LaunchCtx &c, const XferSegs &sg) {
  const int len = xfer_max_len(sg, c.r);
  if (sg.n <= 0 || len <= 0) return;
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL((k_xfer_multi<R, false>), dim3((len + 255) / 256, sg.n), dim3(256), 0, c.stream, sg));
}

void launch_unpack_multi(const LaunchCtx &c, const XferSegs &sg) {
  const int len = xfer_max_len(sg, c.r);
  if (sg.n <= 0 || len <= 0) return;
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL((k_xfer_multi<R, true>), dim3((len + 255) / 256, sg.n), dim3(256), 0, c.stream, sg));
}

void launch_upload2(const LaunchCtx &c, double *slab0, double *slab1, const int *host_slots, const double *host_in, int n0, int n1) {
  if (n0 + n1 <= 0) return;
  const int len = (n0 + n1) * 4 * c.r;
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_upload2<R>, dim3((len + 255) / 256), dim3(256), 0, c.stream, slab0, slab1, host_slots,
                                          host_in, n0, n1));
}

void launch_report(const LaunchCtx &c, int ai, const int *frames, int count, double *host_out, int stat_off, int stat_cnt,
                   int stat_stride, int opt_nb, unsigned long long *seq, int advance, int accel, int num_robots,
                   int restart_interval, const int *up_slots, const double *up_in, int up_n0, int up_n1, int one_seq) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_report<R>, dim3(1), dim3(512), 0, c.stream, c.agents, ai, frames, count, host_out,
                                          stat_off, stat_cnt, stat_stride, opt_nb, seq, advance, accel, num_robots,
                                          restart_interval, up_slots, up_in, up_n0, up_n1, one_seq));
}

void launch_iterate_false(const LaunchCtx &c, int ai, int n, int num_robots, int restart_interval, const int *pubpos_ptr,
                          const int *pubpos, double *host_out, unsigned long long *seq, unsigned long long *ticket,
                          const int *up_slots, const double *up_in, int up_n0, int up_n1) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_iterate_false<R>, dim3((n + 63) / 64), dim3(64), 0, c.stream, c.agents, c.team, ai,
                                          num_robots, restart_interval, pubpos_ptr, pubpos, host_out, seq, ticket, up_slots, up_in,
                                          up_n0, up_n1));
}

void launch_residuals(const LaunchCtx &c, int ai, int nedges) {
  if (nedges <= 0) return;
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_residuals<R>, dim3((nedges + 63) / 64), dim3(64), 0, c.stream, c.agents, ai));
}

void launch_mail_signal(hipStream_t s, const MailSignals &sig) {
  if (sig.count > 0) hipLaunchKernelGGL(k_mail_signal, dim3(1), dim3(64), 0, s, sig);
}

void launch_mail_wait(hipStream_t s, const unsigned long long *mail, const MailWaits &w, int *err) {
  static const long long ticks = [] {
    const char *e = std::getenv("DPGO_MAIL_TIMEOUT_S");
    const double sec = e ? std::atof(e) : 20.0;
    return (long long)((sec > 0 ? sec : 20.0) * 1e8);
  }();
  if (w.count > 0) hipLaunchKernelGGL(k_mail_wait, dim3(1), dim3(64), 0, s, mail, w, err, ticks);
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

// Q's other device layouts refreshed FROM its block-CSR values on the device (a weight update: same pattern, new values) --
// the slot-major ELL copy, the CSR tail beyond its width, the lane-ordered copy of the one-launch iteration.  Pure copies
// (the values are bitwise the block-CSR ones, as when the host lays them out); slots a row does not have keep the zeros
// of the first, host-built upload.  One thread per (pose slot jt, 16-byte chunk q): assembly.hip says where each goes.
__global__ void k_q_layouts(const int *rowptr, const double *qval, int n, int EW, double *ell_val, const int *trowptr, double *tval,
                            int SW, int tiles, double *soa_val) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int jt = x >> 3, q = x & 7;
  if (jt >= tiles * 64 && jt >= n) return;
  const int j = min(jt, n - 1);
  const int p0 = rowptr[j], p1 = rowptr[j + 1];
  if (jt < n) {
    for (int u = 0; u < EW && p0 + u < p1; ++u) {
      const v2d_t v = *(const __attribute__((address_space(1))) v2d_t *)(qval + (size_t)16 * (p0 + u) + 2 * q);
      *(__attribute__((address_space(1))) v2d_t *)(ell_val + ((size_t)u * n + j) * 16 + 2 * q) = v;
    }
    for (int p = p0 + EW; p < p1; ++p) {
      const v2d_t v = *(const __attribute__((address_space(1))) v2d_t *)(qval + (size_t)16 * p + 2 * q);
      *(__attribute__((address_space(1))) v2d_t *)(tval + (size_t)16 * (trowptr[j] + (p - p0 - EW)) + 2 * q) = v;
    }
  }
  if (soa_val && jt < tiles * 64 && jt < n) {
    const int tile = jt / 64, lane = jt % 64;
    for (int u = 0; u < SW && p0 + u < p1; ++u) {
      const v2d_t v = *(const __attribute__((address_space(1))) v2d_t *)(qval + (size_t)16 * (p0 + u) + 2 * q);
      *(__attribute__((address_space(1))) v2d_t *)(soa_val + ((((size_t)tile * SW + u) * 8 + q) * 64 + lane) * 2) = v;
    }
  }
}

void launch_q_layouts(hipStream_t s, const int *rowptr, const double *qval, int n, int EW, double *ell_val, const int *trowptr,
                      double *tval, int SW, int tiles, double *soa_val) {
  const int slots = std::max(n, soa_val ? tiles * 64 : 0);
  hipLaunchKernelGGL(k_q_layouts, dim3((slots * 8 + 255) / 256), dim3(256), 0, s, rowptr, qval, n, EW, ell_val, trowptr, tval, SW, tiles,
                     soa_val);
}

void launch_bsr_to_dense(hipStream_t s, const int *rowptr, const int *col, const double *qval, int n, double shift,
                         double *A) {
  (void)hipMemsetAsync(A, 0, sizeof(double) * (size_t)16 * n * n, s);
  hipLaunchKernelGGL(k_bsr_to_dense, dim3(n), dim3(64), 0, s, rowptr, col, qval, n, shift, A);
}

}  // namespace dpgo
