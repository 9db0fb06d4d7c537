// spmm.hip -- block-sparse kernels: G assembly, cost / gradient, Hessian-vector products, the tCG Hessian step,
// the trust-region trial evaluation and the heterogeneous evaluation launches of the fused RGD iteration
// (SURVEY 8a rows a2, a3, a4).  One lane owns one (pose, row) pair; see kernel_common.h for the layout.
#include "kernel_common.h"

namespace dpgo {

// ------------------------------------------------------------------------------------------------
// stand-alone G assembly.  One lane per (public pose, row a).  pull != 0: read the neighbour's pose
// straight from the neighbour agent's X / Y array on this GPU (device-to-device exchange that
// replaces the PublicPoses topic) and refresh the slab; else read the slab filled by unpack.
template <int R>
__global__ __launch_bounds__(64) void k_buildG(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, int aux,
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
__global__ void k_pull(const AgentDev *__restrict__ agents, int dst) {
  const AgentDev &ag = agents[dst];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ag.nshared * 4 * R) return;
  const int e = t / (4 * R), k = t - e * 4 * R;
  const SharedEdgeDev &se = ag.se[e];
  if (!se.src[0]) return;  // neither co-resident nor imported: its poses arrive as messages
  ag.nbr[0][(size_t)se.slot * 4 * R + k] = se.src[0][k];
  ag.nbr[1][(size_t)se.slot * 4 * R + k] = se.src[1][k];
}

template <int R>
__global__ __launch_bounds__(64) void k_eval(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, int xb, int egb,
                                             int gfb, int poff, int gmode, int aux) {
  constexpr int PPB = 64 / R;
  __shared__ double Ysh[PPB * 4 * R], Wsh[PPB * 4 * R];
  eval_body<R>(agents, team, sel, xb, egb, gfb, poff, gmode, aux, (int)blockIdx.x, Ysh, Wsh, agents[0]);
}

// k_eval + the report of the per-agent API (ReportTail, kernels.h): the closing statistics evaluation of an RGD
// iterate(true) is the last launch of the call, and the report a launch of its own behind it cost the host's wait 7-10 us
// (one workgroup: cold start, two dependent loads per public pose, the sums, a PCIe round trip for the fence).  Here
//   * every workgroup first copies its share of the public poses (X, Y: final since the step kernel) into the pinned image
//     -- posted writes, on their way while the evaluation runs --,
//   * a system-scope fence behind the evaluation covers them and the workgroup's partial sums; one ticket per workgroup,
//   * the workgroup with the last ticket does what k_report's waves did: the sums (same routines, same order: the same
//     bits), the agent's end-of-iterate bookkeeping, the scalars, the sequence word.
template <int R>
__global__ __launch_bounds__(64) void k_eval_report(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, int xb, int egb,
                                                    int gfb, int poff, int gmode, int aux, int nb_eval, const ReportTail rt) {
  constexpr int PPB = 64 / R;
  __shared__ double Ysh[PPB * 4 * R], Wsh[PPB * 4 * R];
  const AgentDev &ra = agents[rt.ai];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= nb_eval) {
    // the copy workgroups (behind the evaluation's in the grid: memory reads return in order, so in front of an evaluation's
    // own loads these two dependent ones would hold it up)
    const int len = rt.count * 4 * R, total = 2 * len, stride = 64 * ((int)gridDim.x - nb_eval);
    for (int base = ((int)blockIdx.x - nb_eval) * 64 + tid; base < total; base += 4 * stride) {
      double v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = min(base + u * stride, total - 1);
        const int sq = t >= len, w = t - sq * len, q = w / (4 * R), k = w - q * 4 * R;
        v[u] = ra.buf[sq ? B_Y : B_X][(size_t)rt.frames[q] * 4 * R + k];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (base + u * stride < total) rt.out[8 + base + u * stride] = v[u];
    }
    __threadfence_system();  // (plain stores to host memory: complete once this workgroup's L2 has been written back)
  } else {
    eval_body<R>(agents, team, sel, xb, egb, gfb, poff, gmode, aux, (int)blockIdx.x, Ysh, Wsh, agents[0]);
    // an evaluation workgroup publishes two partial sums: written once more, THROUGH to memory (st_c), and waited for --
    // a release fence would write the whole L2 back in every one of them (what that costs: profiles/r06_agent_api.md)
    if (tid == 0) {
      double *P = ra.part + poff + (size_t)blockIdx.x * PART_STRIDE;
      const double f = gp(P)[0], g = gp(P)[1];
      st_c(P, f); st_c(P + 1, g);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  unsigned int tk = 0;
  if (tid == 0) tk = (unsigned int)__hip_atomic_fetch_add(rt.ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  tk = (unsigned int)__builtin_amdgcn_readfirstlane((int)tk);
  if (tk + 1u != gridDim.x) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (every lane of the wave reads the other workgroups' partials below)
  if (rt.advance && tid == 63) advance_agent(ra, rt.accel, rt.num_robots, rt.restart_interval);
  // the five sums k_report forms on five waves, here on one: all their loads first (sum_partials / sum_partials2 keep their
  // order of additions: the same bits)
  // (the partials of THIS launch were written by other workgroups, on other XCDs: read with agent-scope loads -- served
  // past this XCD's L2 whatever it holds of their lines, two workgroups' partials share a 128-byte line)
  auto ldc8 = [](const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  double sc[5] = {0, 0, 0, 0, 0};
  {
    double s0 = 0, c0 = 0, c1 = 0, a0 = 0, a1 = 0;
    const int most = max(rt.stat_cnt, rt.opt_nb);
    for (int base = 0; base < most; base += 512) {  // (sum_partials' trips and order of additions: the same bits)
      double sv[8], cv0[8], cv1[8], av0[8], av1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + tid + 64 * u;
        sv[u] = (i < rt.stat_cnt) ? ldc8(ra.part + rt.stat_off + (size_t)i * rt.stat_stride) : 0.0;
        const bool o = i < rt.opt_nb;
        const double *pc = ra.part + PART_C + (size_t)(o ? i : 0) * PART_STRIDE, *pa = ra.part + PART_A + (size_t)(o ? i : 0) * PART_STRIDE;
        cv0[u] = o ? ldc8(pc) : 0.0; cv1[u] = o ? ldc8(pc + 1) : 0.0;
        av0[u] = o ? ldc8(pa) : 0.0; av1[u] = o ? ldc8(pa + 1) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s0 += sv[u]; c0 += cv0[u]; c1 += cv1[u]; a0 += av0[u]; a1 += av1[u]; }
    }
    sc[0] = wave_sum(s0); sc[1] = wave_sum(c0); sc[2] = wave_sum(c1); sc[3] = wave_sum(a0); sc[4] = wave_sum(a1);
  }
  if (tid == 0) {
    if (rt.stat_cnt > 0) rt.out[1] = sc[0];
    if (rt.opt_nb > 0) { rt.out[2] = sc[1]; rt.out[3] = sc[2]; rt.out[4] = sc[3]; rt.out[5] = sc[4]; }
    __hip_atomic_store(rt.ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *rt.seq = rt.expect;
  }
  __threadfence_system();
  if (tid == 0) __hip_atomic_store(reinterpret_cast<unsigned long long *>(rt.out), rt.expect, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// k_eval for agents whose poses carry MANY shared edges (the tunnels robots: up to 20 inter-robot loop closures per pose,
// 3548 of the 4891 edges cross robots).  g_row_range chases them four at a time -- descriptor, then neighbour pose and
// coefficients, two dependent round trips per four edges -- behind the lane's own SpMM.  Here three helper waves fetch the
// operands of ALL shared edges of the tile into LDS while the tile's wave runs its SpMM: every 16-byte chunk of the
// coefficients and every edge descriptor in one round trip, every chunk of the neighbours' poses in a second one,
// whatever the number of edges; ONE barrier, the helpers leave, and the tile's lanes form G from LDS in g_row_range's
// order (bitwise the same sums).  A pulled neighbour pose also goes to the slab, as g_row_range does.
constexpr int EVS_NT = 256, EVS_NH = EVS_NT - 64;
template <int R>
__global__ __launch_bounds__(EVS_NT) void k_eval_staged(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, int xb, int egb,
                                                        int gfb, int poff, int gmode, int aux, int cap) {
  constexpr int PPB = 64 / R, EPE = 4 * R + 16, XCH = 2 * R;  // XCH: 16-byte chunks of a neighbour pose
  __shared__ double Ysh[PPB * 4 * R], Wsh[PPB * 4 * R];
  extern __shared__ __attribute__((aligned(16))) double Eop[];  // [cap][EPE]
#ifdef DPGO_EVS_TRACE
  const unsigned long long ts0 = wall_clock64();
#define EVS_STAMP(k) do { if ((threadIdx.x & 63) == 0) ag.part[PART_E + (4000 + (int)blockIdx.x) * PART_STRIDE + (k)] = (double)(wall_clock64() - ts0); } while (0)
#else
#define EVS_STAMP(k) do { } while (0)
#endif
  const AgentDev &ag = agents[sel_cur(team, sel)];
  const int bx = (int)blockIdx.x;
  const int j0 = bx * PPB;
  if (j0 >= ag.n) return;
  const int E0 = gp(ag.pose_eptr)[j0], E1 = gp(ag.pose_eptr)[min(j0 + PPB, ag.n)], cnt = E1 - E0;
  if (threadIdx.x >= 64) {
    // (every wave of the workgroup meets the same barriers: the one behind the staging when the tile has shared edges,
    // and eval_body's in front of the tangent projection -- no wave leaves a barrier behind it for the others)
    if (cnt == 0) { __syncthreads(); return; }
    const int t = (int)threadIdx.x - 64;
    const bool pull = gmode == 2;
    constexpr int CB = 8;  // chunks per lane and pass (the host keeps cap * XCH <= a few passes of EVS_NH * CB)
    const int nco = cnt * 8, nx = cnt * XCH;
    for (int base = 0; base < max(nco, nx); base += EVS_NH * CB) {
      // coefficient chunks and the descriptors of the edges whose pose chunks this lane takes: addresses known from the
      // edge index alone, one round trip
      double2 vc[CB];
      const double *src[CB];
      int slot[CB], ed[CB], part[CB];
#pragma unroll
      for (int u = 0; u < CB; ++u) {
        const int q = min(base + t + EVS_NH * u, nco - 1);
        vc[u] = ld2(ag.se[E0 + (q >> 3)].coef + 2 * (q & 7));
        const int qx = min(base + t + EVS_NH * u, nx - 1);
        ed[u] = qx / XCH; part[u] = qx - ed[u] * XCH;
        const auto *se = gp(ag.se) + (E0 + ed[u]);
        src[u] = se->src[aux]; slot[u] = se->slot;
      }
      // the neighbours' poses: one more round trip
      double2 vx[CB];
      bool cp[CB];
#pragma unroll
      for (int u = 0; u < CB; ++u) {
        cp[u] = pull && src[u];
        const double *xp = cp[u] ? src[u] : ag.nbr[aux] + (size_t)slot[u] * 4 * R;
        vx[u] = ld2(xp + 2 * part[u]);
      }
#pragma unroll
      for (int u = 0; u < CB; ++u) {
        const int q = base + t + EVS_NH * u;
        if (q < nco) *reinterpret_cast<double2 *>(Eop + (size_t)(q >> 3) * EPE + 4 * R + 2 * (q & 7)) = vc[u];
      }
#pragma unroll
      for (int u = 0; u < CB; ++u) {
        if (base + t + EVS_NH * u < nx) {
          *reinterpret_cast<double2 *>(Eop + (size_t)ed[u] * EPE + 2 * part[u]) = vx[u];
          if (cp[u]) { const v2d_t tv = {vx[u].x, vx[u].y}; *(__attribute__((address_space(1))) v2d_t *)(ag.nbr[aux] + (size_t)slot[u] * 4 * R + 2 * part[u]) = tv; }
        }
      }
    }
    EVS_STAMP(1 + (t >> 6));
    __syncthreads();  // (pairs with the tile's wave: eval_body, behind its SpMM)
    __syncthreads();  // (... and with eval_body's barrier in front of the tangent projection)
    return;
  }
  eval_body<R>(agents, team, sel, xb, egb, gfb, poff, gmode, aux, bx, Ysh, Wsh, agents[0], cnt > 0 ? Eop : nullptr, E0);
  EVS_STAMP(0);
#ifdef DPGO_EVS_TRACE
  if (threadIdx.x == 0) ag.part[PART_E + (4000 + (int)blockIdx.x) * PART_STRIDE + 4] = (double)cnt;
#endif
}

// generic Riemannian Hessian-vector product at point xb with Euclidean gradient egb:  ob = Hess[vb]
// partials: [0] <v, Hv>
template <int R>
__global__ __launch_bounds__(64) void k_hess(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, int xb, int egb,
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
      for (int cp = 0; cp < 4; ++cp) x[0][cp] = gp(V)[((size_t)4 * i + cp) * R + a];
    }, w);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      vrow[c] = gp(V)[((size_t)4 * j + c) * R + a];
      Ysh[lp * 4 * R + c * R + a] = gp(ag.buf[xb])[((size_t)4 * j + c) * R + a];
      Esh[lp * 4 * R + c * R + a] = gp(ag.buf[egb])[((size_t)4 * j + c) * R + a];
    }
  }
  __syncthreads();
  hess_tail<R>(Ysh + lp * 4 * R, Esh + lp * 4 * R, Wsh + lp * 4 * R, a, w[0], vrow, hrow, act);
  double d = 0;
  if (act) {
    double *O = ag.buf[ob] + (size_t)j * 4 * R;
#pragma unroll
    for (int c = 0; c < 4; ++c) { gp(O)[c * R + a] = hrow[c]; d += vrow[c] * hrow[c]; }
  }
  d = wave_sum(d);
  if (lane == 0) gp(ag.part)[poff + (size_t)blockIdx.x * PART_STRIDE] = d;
}

// ------------------------------------------------------------------------------------------------
// tCG body, part 1:  delta <- -z + beta delta (on the fly), Hd = Hess[delta], partial <delta, Hd>.
template <int R>
__global__ __launch_bounds__(64) void k_tcg_hv(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, int sp,
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
  const int npb = precond_nblk(ag);
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
        x[0][cp] = fresh ? gp(Dnew)[o] : (-gp(Z)[o] + beta * gp(Dold)[o]);
      }
    }, w);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const size_t o = ((size_t)4 * j + c) * R + a;
      vrow[c] = fresh ? gp(Dnew)[o] : (-gp(Z)[o] + beta * gp(Dold)[o]);
      if (!fresh) gp(Dnew)[o] = vrow[c];
      Ysh[lp * 4 * R + c * R + a] = gp(ag.buf[B_X])[o];
      Esh[lp * 4 * R + c * R + a] = gp(ag.buf[B_EGRAD])[o];
    }
  }
  __syncthreads();
  hess_tail<R>(Ysh + lp * 4 * R, Esh + lp * 4 * R, Wsh + lp * 4 * R, a, w[0], vrow, hrow, act);
  double d = 0;
  if (act) {
    double *O = ag.buf[B_HD] + (size_t)j * 4 * R;
#pragma unroll
    for (int c = 0; c < 4; ++c) { gp(O)[c * R + a] = hrow[c]; d += vrow[c] * hrow[c]; }
  }
  d = wave_sum(d);
  if (lane == 0) gp(ag.part)[PART_A + (size_t)blockIdx.x * PART_STRIDE] = d;
}

// Heterogeneous launch that closes iteration k and opens iteration k+1 inside captured graphs: the first
// nest_tiles * num_agents workgroups run the Nesterov step of every agent (k_nest_pre), the rest evaluate
// f_opt / gradnorm_opt of the agent that just optimized on its snapshot B_X2 (k_eval, sel = stats_sel).
// The two halves touch disjoint data: the statistics read B_X2 / G of agent a, the Nesterov step writes
// X, Y, V, XPrev; one launch boundary per iteration disappears.
template <int R>
__global__ __launch_bounds__(64) void k_stats_nest(const AgentDev *__restrict__ agents, TeamDev *team, int nest_tiles, int num_agents,
                                                   int num_robots, int restart_interval) {
  __shared__ Tile<R> TX, TV;
  const int nb_nest = nest_tiles * num_agents;
  const int b = (int)blockIdx.x;
  if (b < nb_nest) {
    nest_pre_body<R>(agents, team, -1, -1, num_robots, restart_interval, b % nest_tiles, b / nest_tiles, TX, TV);
  } else {
    eval_body<R>(agents, team, -5, B_X2, B_EGRAD2, B_GF2, PART_A, 0, 0, b - nb_nest, TX.d, TV.d, agents[0]);
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
template <int R, bool BAKED>
__global__ __launch_bounds__(64) void k_eval_stats(const AgentDev *__restrict__ agents, TeamDev *team, int nb_eval, int first,
                                                   int has_eval, int has_stats, int num_robots, int restart_interval,
                                                   int eval_sel, int stats_sel, const NestState *nest_copy, const AgentDev agv) {
  constexpr int PPB = 64 / R;
  __shared__ double Ysh[PPB * 4 * R], Wsh[PPB * 4 * R];
  const int b = (int)blockIdx.x;
  if (b == (int)gridDim.x - 1) {  // the extra workgroup: bookkeeping only, so that no evaluation waits for it
    if (!first && threadIdx.x == 0) {
      if (nest_copy) {
        // the iterations in front of this one were one-launch iterations (step_fused.hip): they moved iter / cur_sel
        // themselves and left the advanced NestStates next to the team's own
        for (int k = 0; k < team->num_agents; ++k) *agents[k].nest = nest_copy[k];
      } else {
        for (int k = 0; k < team->num_agents; ++k) advance_agent(agents[k], 1, num_robots, restart_interval);
        team->iter += 1;
        if (has_eval) team->cur_sel = team->next_sel;
      }
    }
    return;
  }
  if (has_eval && b < nb_eval) {
    eval_body<R, false, BAKED>(agents, team, eval_sel >= 0 ? eval_sel : (first ? -1 : -6), B_X, B_EGRAD, B_GF, PART_C, 2, 1, b, Ysh, Wsh, agv);
  } else if (has_stats) {
    eval_body<R>(agents, team, stats_sel >= 0 ? stats_sel : -5, B_X2, B_EGRAD2, B_GF2, PART_A, 0, 0,
                 b - (has_eval ? nb_eval : 0), Ysh, Wsh, agents[0]);
  }
}

// outer step, evaluation at the candidate x2 = Retr_x1(eta):
//   egrad2 = x2 Q + G, rgrad2, Heta = Hess_x1[eta];  partials [0] f2 [1] |rgrad2|^2 [2] <gf,eta> [3] <eta,Heta>
// (both SpMM rows share every Q block load)
template <int R>
__global__ __launch_bounds__(64) void k_rtr_eval2(const AgentDev *__restrict__ agents, const TeamDev *team, int sel, int sp) {
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

// (k_eval_staged's static LDS -- Ysh, Wsh: 2 x 64 x 4 doubles at most -- rounded up; assembly.hip admits a tile when its
// operands fit beside it in the device's limit)
size_t eval_staged_lds_bytes(int r, int cap) { return (size_t)cap * (4 * r + 16) * 8; }

void launch_eval(const LaunchCtx &c, int sel, int max_n, int xb, int egb, int gfb, int poff, const EvalOpts &o) {
  if (o.gmode != 0 && c.stage_cap > 0) {  // agents with many shared edges per pose: the edges' operands through LDS
    const size_t dyn = eval_staged_lds_bytes(c.r, c.stage_cap);
    hipError_t e = hipSuccess;
    DPGO_DISPATCH_R(c.r, {
      static bool configured = false;
      if (!configured) {
        e = hipFuncSetAttribute((const void *)k_eval_staged<R>, hipFuncAttributeMaxDynamicSharedMemorySize, c.max_lds - EVS_STATIC_LDS);
        configured = (e == hipSuccess);
      }
      if (e == hipSuccess)
        hipLaunchKernelGGL(k_eval_staged<R>, dim3(spmm_grid(c.r, max_n), c.ny), dim3(EVS_NT), dyn, c.stream, c.agents, c.team, sel, xb,
                           egb, gfb, poff, o.gmode, o.aux, c.stage_cap);
    });
    if (e == hipSuccess) return;
  }
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_eval<R>, dim3(spmm_grid(c.r, max_n), c.ny), dim3(64), 0, c.stream, c.agents,
                                          c.team, sel, xb, egb, gfb, poff, o.gmode, o.aux));
}

void launch_eval_report(const LaunchCtx &c, int sel, int max_n, int xb, int egb, int gfb, int poff, const EvalOpts &o,
                        const ReportTail &rt) {
  // evaluation workgroups + one copy workgroup per 256 doubles of public poses (four per lane; 16 at most)
  const int nb = spmm_grid(c.r, max_n), ncopy = std::max(1, std::min(16, (2 * rt.count * 4 * c.r + 255) / 256));
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_eval_report<R>, dim3(nb + ncopy, 1), dim3(64), 0, c.stream, c.agents,
                                          c.team, sel, xb, egb, gfb, poff, o.gmode, o.aux, nb, rt));
}

void launch_hess(const LaunchCtx &c, int sel, int max_n, int xb, int egb, int vb, int ob, int poff) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_hess<R>, dim3(spmm_grid(c.r, max_n), c.ny), dim3(64), 0, c.stream, c.agents,
                                          c.team, sel, xb, egb, vb, ob, poff));
}

void launch_eval_stats(const LaunchCtx &c, int max_n, int first, int has_eval, int has_stats, int num_robots,
                       int restart_interval, int eval_sel, int stats_sel, const NestState *nest_copy) {
  const int nb = spmm_grid(c.r, max_n);
  const int grid = nb * ((has_eval ? 1 : 0) + (has_stats ? 1 : 0)) + 1;
  if (has_eval && eval_sel >= 0 && c.host_agents && c.bake_desc) {
    const AgentDev &d = c.host_agents[eval_sel];
    DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL((k_eval_stats<R, true>), dim3(grid), dim3(64), 0, c.stream, c.agents, c.team, nb, first,
                                            has_eval, has_stats, num_robots, restart_interval, eval_sel, stats_sel, nest_copy, d));
    return;
  }
  AgentDev none{};
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL((k_eval_stats<R, false>), dim3(grid), dim3(64), 0, c.stream, c.agents, c.team, nb, first,
                                          has_eval, has_stats, num_robots, restart_interval, eval_sel, stats_sel, nest_copy, none));
}

void launch_tcg_hv(const LaunchCtx &c, int sel, int max_n, int sp, int max_inner) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_tcg_hv<R>, dim3(spmm_grid(c.r, max_n), c.ny), dim3(64), 0, c.stream, c.agents,
                                          c.team, sel, sp, max_inner));
}

void launch_stats_nest(const LaunchCtx &c, int num_agents, int max_n, int num_robots, int restart_interval) {
  const int nest_tiles = (max_n + 63) / 64;
  const int grid = nest_tiles * num_agents + spmm_grid(c.r, max_n);
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_stats_nest<R>, dim3(grid), dim3(64), 0, c.stream, c.agents, c.team, nest_tiles,
                                          num_agents, num_robots, restart_interval));
}

void launch_rtr_eval2(const LaunchCtx &c, int sel, int max_n, int sp) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL(k_rtr_eval2<R>, dim3(spmm_grid(c.r, max_n), c.ny), dim3(64), 0, c.stream, c.agents,
                                          c.team, sel, sp));
}

}  // namespace dpgo
