// step_persist.hip -- K deep-carried accelerated-RGD iterations in ONE persistent launch (k_step_pd; round 6, the form the
// round-5 verdict asked for, built on top of step_deep.hip's pipeline).  The body of an iteration is k_step_fd's, role for
// role and expression for expression (generated from it: the iterates are bitwise those of the per-launch form and of the
// two-launch sequence); what changes is what sits between two iterations:
//   * no kernel boundary: a grid hand-off (every workgroup arrives on its XCD's counter, the last one per XCD on the top
//     counter, the last of those releases eight generation words -- rtr_fused.hip's tree) instead of end-of-kernel, dispatch
//     and a cold start;
//   * everything an iteration publishes for the OTHER workgroups (poses, their twins, V, the carried arrays) is stored
//     write-through (st_c) and read with L1-bypassing loads (ldc / CVec) -- per-XCD L2s are not coherent with each other;
//   * what stays inside a workgroup stays in LDS: the partial sums of the next agent's product (the same workgroup owns
//     the same columns of every agent), the Nesterov scalars (every workgroup advances its own copy);
//   * the descriptors of the agents come from the device array (scalar loads; static data).
// The grid must be resident at once (one 512-thread workgroup of 150 KB LDS per CU, <= 256 of them): the host launches it
// only while it holds the device's persistent-kernel lock (solve.hip), every spin is bounded, and a hand-off that times out
// raises the team's error word -- the run is then invalid and reported as such.
#include "kernel_common.h"
#include "step_deep_dev.h"
#include <algorithm>

namespace dpgo {

// a value another workgroup of THIS launch may have written: served by the L2 / the fabric, never by this CU's L1
__device__ __forceinline__ double ldc(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// fd_cur_request with the partial sums in LDS
template <int R, int M0, int NC>
__device__ __forceinline__ void pd_cur_request(const AgentDev &ag, const double *pacc_lds, int bx, int l, FdCur<R, NC> &cu) {
  const int cg = (l >> 5) & 7, kl = l & 31;
  const int N4 = ag.N4, col = 8 * bx + cg;
  const double *Mc = ag.M + (size_t)((col < N4) ? col : 0) * N4;
#pragma unroll
  for (int i = 0; i < NC; ++i) cu.mc[i] = ld2_nt(Mc + min(2 * kl + 64 * (int)ag.fe_ord[M0 + i], N4 - 2));
#pragma unroll
  for (int a = 0; a < R; ++a) cu.pa[a] = pacc_lds[a * 256 + l];
}

// fd_xn_request with L1-bypassing loads (the neighbours' poses are what the previous iteration of this launch wrote)
template <int R>
__device__ __forceinline__ void pd_xn_request(const AgentDev &ag, const FeBases &fb, int parity, int ln, FdXn<R> &xr) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    unsigned wsel = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const unsigned wk = ag.fe_code[(32 * q + k < FE_MAX_EDGES / 2) ? 32 * q + k : FE_MAX_EDGES / 2 - 1];
      wsel = ((ln >> 1) == k) ? wk : wsel;
    }
    const unsigned code = (ln & 1) ? (wsel >> 16) : (wsel & 0xffffu);
    const int sa = (int)(code >> 12), sf = (int)(code & 0xfffu);
    const double *yb = fb.ybase[0];
    int yn = fb.npose[0];
#pragma unroll
    for (int k = 1; k < LOOKAHEAD_MAX_AGENTS; ++k) { yb = (sa == k) ? fb.ybase[k] : yb; yn = (sa == k) ? fb.npose[k] : yn; }
    const double *xp = yb + (parity ? (size_t)B_ALT * 4 * R * yn : (size_t)0) + (size_t)sf * 4 * R;
#pragma unroll
    for (int k = 0; k < 2 * R; ++k) xr.v[q][k] = make_double2(ldc(xp + 2 * k), ldc(xp + 2 * k + 1));
  }
}

// ---- the hand-off between two iterations (rtr_fused.hip's counter tree): bar[g * 16] arrivals of XCD g, bar[8 * 16] the XCDs
// that are complete, bar[(9 + g) * 16] generation word of XCD g, bar[17 * 16 + 1] abort.  The words are zero at launch.
constexpr int PB_LINE = 16, PB_TOP = 8 * PB_LINE, PB_GEN = 9 * PB_LINE, PB_ABORT = 17 * PB_LINE + 1;
constexpr long long PB_TIMEOUT_TICKS = 20000000;  // 0.2 s of the 100 MHz wall clock

struct PdBar {
  unsigned long long *bar;
  unsigned long long epoch;
  int g, size_g, ngroups;
  int *err, *ok;
};

__device__ __forceinline__ bool pd_grid_sync(PdBar &gb) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // every wave: its write-through stores have left the CU
  __syncthreads();
  gb.epoch += 1ull;
  if (threadIdx.x == 0) {
    const unsigned long long old = __hip_atomic_fetch_add(&gb.bar[gb.g * PB_LINE], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1ull == (unsigned long long)gb.size_g * gb.epoch) {
      const unsigned long long old2 = __hip_atomic_fetch_add(&gb.bar[PB_TOP], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old2 + 1ull == (unsigned long long)gb.ngroups * gb.epoch) {
#pragma unroll
        for (int q = 0; q < 8; ++q) __hip_atomic_store(&gb.bar[PB_GEN + q * PB_LINE], gb.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    const long long t_start = (long long)wall_clock64();
    while (__hip_atomic_load(&gb.bar[PB_GEN + gb.g * PB_LINE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gb.epoch) {
      __builtin_amdgcn_s_sleep(1);
      if ((long long)wall_clock64() - t_start > PB_TIMEOUT_TICKS ||
          __hip_atomic_load(&gb.bar[PB_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) {
        *gb.err = 5;
        *gb.ok = 0;
        __hip_atomic_store(&gb.bar[PB_ABORT], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
  return *gb.ok != 0;
}

// -DDPGO_PD_TRACE: wall-clock stamps of iteration DPGO_PD_TRACE_IT of hardware workgroup 100 (LDS, flushed at the end into
// agent 0's partial-sum scratch, PART_E words [4000 ..]: [wave][16]) -- profiles/experiments/pd_trace.py
#ifdef DPGO_PD_TRACE
#ifndef DPGO_PD_TRACE_IT
#define DPGO_PD_TRACE_IT 40
#endif
#define PD_TRACE_DECL __shared__ unsigned long long pd_stamps[8 * 16];
#define PD_STAMP(k) do { if (it == DPGO_PD_TRACE_IT && (threadIdx.x & 63) == 0) pd_stamps[(threadIdx.x >> 6) * 16 + (k)] = wall_clock64(); } while (0)
#else
#define PD_TRACE_DECL
#define PD_STAMP(k) do { } while (0)
#endif

template <int R, int M0>
__global__ __launch_bounds__(512) void k_step_pd(const AgentDev *__restrict__ agents, TeamDev *team, const int *__restrict__ sched, int sched_len,
                                                 int it0, int K, int B, int L, double step, int num_robots, int restart_interval,
                                                 const NestState *nest_src, NestState *nest_dst, const FeBases fb, int nblk_all, int num_agents,
                                                 unsigned long long *bar, int *err) {
  const int hb = (int)blockIdx.x, gq = (int)gridDim.x / 8;
  const int bx = (hb % 8) * gq + hb / 8;  // XCD-aware block order, as in k_precond
  const int tid0 = threadIdx.x;
  if (bx >= nblk_all) return;
  constexpr int KC = FD_KC, MREG = KC / 64, NC = MREG - M0;
  __shared__ double vs[R * KC];
  __shared__ double zs[8 * R];
  __shared__ double red[32 * (8 * R + 1)];
  __shared__ double Ysh[2 * 4 * R];
  __shared__ double Esh[2][2 * 4 * R];
  __shared__ double Es[FE_MAX_EDGES * (4 * R + 16)];
  __shared__ double tl_x[2 * 4 * R], tl_v[2 * 4 * R], tl_y[2 * 4 * R], tl_s[2 * 16];
  __shared__ double Ex[2 * 3 * 4 * R];
  __shared__ double Psh[2 * 4 * R], tl_rel[2];
  __shared__ double pacc_lds[R * 256];  // the partial sums this workgroup's streamers leave for its waves 4-7 of the next iteration
  __shared__ NestState nsl[LOOKAHEAD_MAX_AGENTS], nsl_next[LOOKAHEAD_MAX_AGENTS];
  __shared__ int sy[FD_SY_COUNT];
  __shared__ int bar_ok;
  PD_TRACE_DECL
  if (tid0 < FD_SY_COUNT) sy[tid0] = 0;
  if (tid0 < LOOKAHEAD_MAX_AGENTS) { nsl[tid0] = nest_src[min(tid0, num_agents - 1)]; nsl_next[tid0] = nsl[tid0]; }
  if (tid0 == 0) bar_ok = __hip_atomic_load(&bar[PB_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull;
  __syncthreads();
  if (!bar_ok) return;
  PdBar gb;
  gb.bar = bar; gb.err = err; gb.ok = &bar_ok; gb.epoch = 0ull;
  gb.g = hb % 8;
  gb.size_g = max(0, min(gq, nblk_all - gb.g * gq));
  gb.ngroups = min(8, (nblk_all + gq - 1) / gq);
  constexpr int EPE = 4 * R + 16;
  const int cwv = __builtin_amdgcn_readfirstlane(tid0 >> 6);

  // iterations -2 and -1 only produce (the row products of sel(0); then its private partial sums and the row products of
  // sel(1)): enqueue_fe_deep's two opening launches
  for (int it = -2; it < K; ++it) {
    // (the lane's index is made opaque once per iteration: everything derived from it is then formed where it is used --
    // hoisted out of the loop those values stay live across all eight roles and spill)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int ln = tid & 63;
    PD_STAMP(0);
    const int ep = it + 3;  // hand-offs inside the workgroup count iterations (the counters in LDS only grow)
    const int rep = max(it, 0);
    const int sel = sched[(it0 + rep) % sched_len], next_sel = sched[(it0 + rep + 1) % sched_len];
    const int dsel = sched[(it0 + max(it + 1, 0)) % sched_len], esel = sched[(it0 + max(it + 2, 0)) % sched_len];
    const int next3_sel = sched[(it0 + rep + 3) % sched_len];
    const int flags = (it == -2) ? FD_W : (it == -1) ? (FD_P | FD_W)
                      : (FD_IN | (it + 1 < K ? FD_P : 0) | (it + 2 < K ? FD_W : 0) | (it + 3 < K ? FD_Y : 0) |
                         (it >= B - L ? FD_STATS : 0) | ((it + 1 < B && it + 1 >= B - L) ? FD_LASTAT : 0));
    const int parity = (it < 0) ? 0 : (it & 1);
    const AgentDev &ag = agents[sel], &agd = agents[dsel], &age = agents[esel];
    const int N4 = ag.N4, n = ag.n;
    const int nblk = (N4 + 7) / 8;
    const bool own = bx < nblk, in = (flags & FD_IN) != 0;
    const int pj0 = own ? 2 * bx : 0, pj1 = (own && 2 * bx + 1 < n) ? 2 * bx + 1 : -1;
    const int npose = own ? ((pj1 >= 0) ? 2 : 1) : 0;

    if (cwv < 4) {
      // ================================================================ streamers
      const int cg = (tid >> 5) & 7, kl = tid & 31;
      for (int t = N4 * R + tid; t < KC * R; t += 256) vs[t] = 0.0;  // rows beyond the agent's
      // the rows of this agent's carried gradient that the last NC chunks meet: positions [M0 * 64 R, N4 R)
      constexpr int NGC = ((KC - 64 * M0) * R / 2 + 255) / 256;
      double2 gc[NGC];
      {
        const CVec Gc(ag.buf[B_CARRY_G], N4 * R);
#pragma unroll
        for (int u = 0; u < NGC; ++u) gc[u] = Gc.ld2(min(M0 * 64 * R + 2 * (tid + 256 * u), N4 * R - 2));
      }
      const int cold = 8 * bx + cg;
      const int N4d = agd.N4;
      const double *Md = agd.M + (size_t)((cold < N4d) ? cold : 0) * N4d;
      double2 mn[M0];
#pragma unroll
      for (int i = 0; i < FD_HEAD; ++i) mn[i] = ld2_nt(Md + min(2 * kl + 64 * (int)agd.fe_ord[i], N4d - 2));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < NGC; ++u) {
        const int tt = M0 * 64 * R + 2 * (tid + 256 * u);
        if (tt < N4 * R) *reinterpret_cast<double2 *>(&vs[tt]) = gc[u];
      }
      fd_signal(&sy[FD_SY_C]);
      PD_STAMP(2);
      fd_wait(&sy[FD_SY_RQ], 4 * ep);  // A: waves 4-7 have requested all they need -- the stream queues behind it, not in front
      PD_STAMP(11);
      // (nothing of the stream is requested in front of this hand-off: a wave stays at the issue of such loads, and the chain
      // waits for C -- the scheduler otherwise hoists the requests above the LDS writes)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" ::: "memory");
      // ---- P: the private chunks of the next agent's M against the private rows of its carried gradient
      constexpr int NGN = (M0 * 64 * R / 2 + 255) / 256;
      double2 gn[NGN];
      const CVec Gd(agd.buf[B_CARRY_G], agd.N4 * R);
#pragma unroll
      for (int u = 0; u < NGN; ++u) gn[u] = Gd.ld2(min(2 * (tid + 256 * u), M0 * 64 * R - 2));
#pragma unroll
      for (int i = FD_HEAD; i < M0; ++i) mn[i] = ld2_nt(Md + min(2 * kl + 64 * (int)agd.fe_ord[i], N4d - 2));
#pragma unroll
      for (int u = 0; u < NGN; ++u) {
        const int tt = 2 * (tid + 256 * u);
        if (tt < M0 * 64 * R) *reinterpret_cast<double2 *>(&vs[tt]) = gn[u];
      }
      fd_signal(&sy[FD_SY_N]);
      fd_wait(&sy[FD_SY_N], 4 * ep);
      fd_wait(&sy[FD_SY_F], 4 * ep);  // (the chain's product first: the two share the LDS)
      PD_STAMP(12);
      double acc[R];
#pragma unroll
      for (int a = 0; a < R; ++a) acc[a] = 0;
#pragma unroll
      for (int i = 0; i < M0; ++i) {
        const int k = 2 * kl + 64 * i;
        double wv[2 * R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
          const double2 t2 = *reinterpret_cast<const double2 *>(&vs[k * R + 2 * q]);
          wv[2 * q] = t2.x; wv[2 * q + 1] = t2.y;
        }
#pragma unroll
        for (int a = 0; a < R; ++a) acc[a] = __builtin_fma(wv[R + a], mn[i].y, __builtin_fma(wv[a], mn[i].x, acc[a]));
      }
      if (flags & FD_P) {
#pragma unroll
        for (int a = 0; a < R; ++a) pacc_lds[a * 256 + tid] = acc[a];  // (the same workgroup continues them in the next iteration)
        PD_STAMP(13);
      }
    } else if (cwv < 6) {
      // the poses live twice (step_fused.hip): this iteration reads the copy of its parity and writes the other one
      const double *__restrict__ Xr = ag.buf[parity ? B_XALT : B_X];
      const double *__restrict__ Yr = ag.buf[parity ? B_YALT : B_Y];
      double *__restrict__ Xw = ag.buf[parity ? B_X : B_XALT];
      double *__restrict__ Yw = ag.buf[parity ? B_Y : B_YALT];
      // ================================================================ the chain: gradient of the public poses, product over the
      // last chunks, step of the workgroup's poses
      const int g = cwv - 4;
      const int npub = ag.npub;
      const int pq = 64 * g + ln;
      const bool pact = pq < npub;
      const int pqc = pact ? pq : 0;
      const int pj = ag.pub_pose[pqc];
      const int pe0 = ag.pub_ptr[pqc], pe1 = ag.pub_ptr[pqc + (pact ? 1 : 0)];
      double w[4 * R], x[4 * R];
      {
        const double *Wc = ag.buf[B_CARRY_W], *Xc = ag.buf[B_CARRY_X];
#pragma unroll
        for (int i = 0; i < 2 * R; ++i) {
          // ([entry pair][public pose][2]: one 16-byte load per pair, a wave's 64 lanes one contiguous KB -- half the requests)
          w[2 * i] = ldc(Wc + ((size_t)i * npub + pqc) * 2); w[2 * i + 1] = ldc(Wc + ((size_t)i * npub + pqc) * 2 + 1);
          x[2 * i] = ldc(Xc + ((size_t)i * npub + pqc) * 2); x[2 * i + 1] = ldc(Xc + ((size_t)i * npub + pqc) * 2 + 1);
        }
      }
      FdCur<R, NC> cu;
      pd_cur_request<R, M0, NC>(ag, pacc_lds, bx, tid - 256, cu);
      // W (wave 5, behind its quarter of the product): the indices of the rows now (wave 4's copies are never used)
      const int npw = (age.n + nblk_all - 1) / nblk_all;  // (<= 2)
      const int wls = ln / (4 * R), we = ln - wls * (4 * R);
      const int pw = bx * npw + wls;
      const bool pv = wls < 3 && wls < npw && pw < age.n;
      const int pwc = pv ? pw : 0;
      const int wtile = pwc >> 6, wpl = pwc & 63, wdn = age.soa_w;
      int ii[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) ii[u] = gp(age.soa_col)[((size_t)wtile * wdn + min(u, wdn - 1)) * 64 + wpl];
      const int qi = gp(age.pub_index)[pwc];
      // operands of the tail (the lanes of wave 4 that will hold them; wave 5's copies are never used)
      const int tl = tid - 256 - 64 * g;
      const size_t own_off = (size_t)((tl >= 4 * R) ? max(pj1, 0) : pj0) * 4 * R + (size_t)(tl % (4 * R));
      double pre_x = 0, pre_v = 0, pre_y = 0, pre_p = 0;
      if (tl < npose * 4 * R) {
        pre_x = ldc(Xr + own_off);
        pre_v = ldc(ag.buf[B_V] + own_off);
        pre_y = ldc(Yr + own_off);
        pre_p = ldc(ag.buf[B_XPREV] + own_off);
      }
      const NestState ns = nsl[sel];
      __builtin_amdgcn_s_setprio(3);  // (the chain's instructions go first: the streamers' product shares the LDS with it)
      fd_signal_requested(&sy[FD_SY_RQ]);
      PD_STAMP(10);
      const int vs_off = fd_pos_off<R>(ag.fe_ord, pj);  // (where the pose's row goes: looked up while the edges' operands land)
      fd_wait(&sy[FD_SY_E], 2 * ep);  // the operands of the shared edges are in LDS
      PD_STAMP(1);
      if (pact) {
        // G_j from LDS: g[c][a] -= x[cp][a] coef[cp + 4c], edge after edge and cp after cp for every entry (g_row_range's order)
        double gg[4 * R];
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) gg[i] = 0.0;
        for (int e = pe0; e < pe1; ++e) {
          const double *E = Es + (size_t)e * EPE;
          double xn[4 * R];
#pragma unroll
          for (int i = 0; i < 2 * R; ++i) { const double2 t = *reinterpret_cast<const double2 *>(E + 2 * i); xn[2 * i] = t.x; xn[2 * i + 1] = t.y; }
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const double2 c01 = *reinterpret_cast<const double2 *>(E + 4 * R + 4 * c), c23 = *reinterpret_cast<const double2 *>(E + 4 * R + 4 * c + 2);
            const double cfc[4] = {c01.x, c01.y, c23.x, c23.y};
#pragma unroll
            for (int cp = 0; cp < 4; ++cp)
#pragma unroll
              for (int a = 0; a < R; ++a) gg[c * R + a] -= xn[cp * R + a] * cfc[cp];
          }
        }
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) w[i] = w[i] + gg[i];
        tangent_inplace<R>(x, w);
      }
      fd_wait(&sy[FD_SY_C], 4 * ep);  // the carried rows are in LDS: the public ones are overwritten now
      PD_STAMP(3);
      if (pact) {
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) vs[vs_off + i] = w[i];
      }
      fd_signal(&sy[FD_SY_D]);
      PD_STAMP(4);
      fd_wait(&sy[FD_SY_D], 2 * ep);
      fd_cur_product<R, M0, NC>(cu, vs, red, tid - 256);
      if (tl >= 0 && tl < npose * 4 * R) { Ysh[tl] = pre_x; Esh[0][tl] = pre_v; Esh[1][tl] = pre_y; Psh[tl] = pre_p; }
      fd_signal(&sy[FD_SY_F]);
      PD_STAMP(5);
      if (g == 1) {
        // ---- W: the row products of agent e at the point B_CARRY_Y holds (left complete by the previous launch), for this
        // workgroup's share of its poses -- one (pose, entry) per lane, fe_block's expression slot after slot (bitwise the sums
        // a self-forming launch would make), then their tangent projection at the point
        __builtin_amdgcn_s_setprio(0);
        const int c = we / R, a = we - c * R;
        const double *__restrict__ Y2 = age.buf[B_CARRY_Y];
        double xv[8][4], bv[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double *xp = Y2 + (size_t)4 * R * ii[u] + a;
          const double *bp = age.soa_val + ((size_t)wtile * wdn + min(u, wdn - 1)) * 1024 + (2 * c) * 128 + 2 * wpl;
          xv[u][0] = ldc(xp); xv[u][1] = ldc(xp + R); xv[u][2] = ldc(xp + 2 * R); xv[u][3] = ldc(xp + 3 * R);
          bv[u][0] = gp(bp)[0]; bv[u][1] = gp(bp)[1]; bv[u][2] = gp(bp)[128]; bv[u][3] = gp(bp)[129];
        }
        const double xe_ = ldc(Y2 + (size_t)4 * R * pwc + we);
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double t = fma4(xv[u][0], bv[u][0], xv[u][1], bv[u][1], xv[u][2], bv[u][2], xv[u][3], bv[u][3], acc);
          acc = (u < wdn) ? t : acc;
        }
        const bool wr = pv && (flags & FD_W);
        if (wr && qi >= 0) {  // (a public pose: its launch finishes it -- row product and point, [entry][public pose])
          st_c(age.buf[B_CARRY_W] + ((size_t)(we >> 1) * age.npub + qi) * 2 + (we & 1), acc);
          st_c(age.buf[B_CARRY_X] + ((size_t)(we >> 1) * age.npub + qi) * 2 + (we & 1), xe_);
        }
        if (wls < 3) { Ex[wls * 4 * R + we] = acc; Ex[3 * 4 * R + wls * 4 * R + we] = xe_; }
        WSYNC();
        if (wr && we == 0) {
          double ww[4 * R], xx[4 * R];
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) { ww[i] = Ex[wls * 4 * R + i]; xx[i] = Ex[3 * 4 * R + wls * 4 * R + i]; }
          tangent_inplace<R>(xx, ww);
          double *Gn = age.buf[B_CARRY_G] + fd_pos_off<R>(age.fe_ord, pw);  // (chunk-ordered: the launches that consume it copy straight ranges)
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) st_c(Gn + i, ww[i]);
        }
      } else {
      fd_wait(&sy[FD_SY_F], 4 * ep);
      PD_STAMP(6);
      if (ln < 8 * R) {
        double s = 0;
#pragma unroll
        for (int q = 0; q < 32; ++q) s += red[q * (8 * R + 1) + ln];
        zs[ln] = s;
      }
      WSYNC();
      const FdNest nn = fd_nest(ns, num_robots, restart_interval);
      const bool restart_now = nn.restart_now, restart_next = nn.restart_next;
      const double nest_gamma = nn.nest_gamma, ahead_alpha = nn.ahead_alpha;
      const bool ahead_opt = next_sel == sel;
      const bool stats = in && (flags & FD_STATS) != 0, lastat = in && (flags & FD_LASTAT) != 0;
      if (ln < 2) tl_rel[ln] = 0.0;
      // ---- the step of the workgroup's two poses: one pose on 16 lanes, the pose in LDS (k_step_fe's tail, device_math.h
      // lane-parallel forms: bitwise the serial routines)
      {
        const int lp = ln >> 4, s = ln & 15;
        if (lp < npose) {
          const size_t o = (size_t)(lp ? pj1 : pj0) * 4 * R;
          double *xs = tl_x + lp * 4 * R, *vsv = tl_v + lp * 4 * R, *ys = tl_y + lp * 4 * R, *Ss = tl_s + lp * 16;
          double *zz = zs + lp * 4 * R;
          const double *x0 = Ysh + lp * 4 * R, *v0 = Esh[0] + lp * 4 * R, *y0 = Esh[1] + lp * 4 * R;
          const int i0 = s, i1 = s + 16;
          const bool h0 = i0 < 4 * R, h1 = i1 < 4 * R;
          tangent_lanes<R>(x0, zz, Ss, s);
          if (h0) xs[i0] = x0[i0] - step * zz[i0];
          if (h1) xs[i1] = x0[i1] - step * zz[i1];
          lanes_sync();
          qf_lanes<R>(xs, Ss, s);
          PD_STAMP(7);
          if (stats) {
            // what k_precond's bit 3 leaves: the snapshot the closing statistics evaluate, |X - XPrev|^2 of the pose (one lane,
            // the serial loop's order)
            if (h0) gp(ag.buf[B_X2])[o + i0] = xs[i0];
            if (h1) gp(ag.buf[B_X2])[o + i1] = xs[i1];
            if (s == 0) {
              const double *p0 = Psh + lp * 4 * R;
              double rel = 0;
#pragma unroll
              for (int i = 0; i < 4 * R; ++i) { const double d = xs[i] - p0[i]; rel += d * d; }
              tl_rel[lp] = rel;
            }
          }
          if (lastat) {
            if (h0) st_c(ag.buf[B_XPREV] + o + i0, xs[i0]);
            if (h1) st_c(ag.buf[B_XPREV] + o + i1, xs[i1]);
          }
          const bool reset = restart_now;
          if (reset) {
            if (h0) vsv[i0] = xs[i0];
            if (h1) vsv[i1] = xs[i1];
            lanes_sync();
          } else {
            const double gamma = nest_gamma;
            if (h0) vsv[i0] = v0[i0] + gamma * (xs[i0] - y0[i0]);
            if (h1) vsv[i1] = v0[i1] + gamma * (xs[i1] - y0[i1]);
            lanes_sync();
            polar_lanes<R>(vsv, Ss, s);
          }
          if (restart_next) {
            if (h0 && in) {
              st_c(Xw + o + i0, xs[i0]);
              if (!ahead_opt) { st_c(Yw + o + i0, xs[i0]); vsv[i0] = xs[i0]; }
              else st_c(Yw + o + i0, reset ? xs[i0] : y0[i0]);
            }
            if (h1 && in) {
              st_c(Xw + o + i1, xs[i1]);
              if (!ahead_opt) { st_c(Yw + o + i1, xs[i1]); vsv[i1] = xs[i1]; }
              else st_c(Yw + o + i1, reset ? xs[i1] : y0[i1]);
            }
            if (lastat && !ahead_opt && s == 0) gp(ag.part)[PART_D + (lp ? pj1 : pj0)] = 0.0;
          } else {
            if (h0) ys[i0] = (1.0 - ahead_alpha) * xs[i0] + ahead_alpha * vsv[i0];
            if (h1) ys[i1] = (1.0 - ahead_alpha) * xs[i1] + ahead_alpha * vsv[i1];
            lanes_sync();
            polar_lanes<R>(ys, Ss, s);
            if (h0 && in) { st_c(Yw + o + i0, ys[i0]); st_c(Xw + o + i0, ys[i0]); }
            if (h1 && in) { st_c(Yw + o + i1, ys[i1]); st_c(Xw + o + i1, ys[i1]); }
            if (lastat && !ahead_opt && s == 0) {  // look-ahead steps leave |Y' - X|^2 per pose
              double rel2 = 0;
#pragma unroll
              for (int i = 0; i < 4 * R; ++i) { const double d = ys[i] - xs[i]; rel2 += d * d; }
              gp(ag.part)[PART_D + (lp ? pj1 : pj0)] = rel2;
            }
          }
          lanes_sync();
          if (h0 && in) st_c(ag.buf[B_V] + o + i0, vsv[i0]);
          if (h1 && in) st_c(ag.buf[B_V] + o + i1, vsv[i1]);
        }
      }
      if (stats) {
        // (k_precond: the two poses' sums through wave_sum, lane 0 stores)
        WSYNC();
        double rl = (ln < npose) ? tl_rel[ln] : 0.0;
        rl = wave_sum(rl);
        if (ln == 0 && own) gp(ag.part)[PART_B + (size_t)bx * PART_STRIDE + 2] = rl;
      }
      }
    } else if (cwv == 6) {
      FdXn<R> er;
      pd_xn_request<R>(ag, fb, parity, ln, er);
      FdCur<R, NC> cu;
      pd_cur_request<R, M0, NC>(ag, pacc_lds, bx, tid - 256, cu);
      fd_signal_requested(&sy[FD_SY_RQ]);
      PD_STAMP(10);
      __builtin_amdgcn_s_setprio(3);
      fd_xn_to_lds<R>(ag, ln, er, Es);
      fd_signal(&sy[FD_SY_E]);
      fd_wait(&sy[FD_SY_D], 2 * ep);  // this wave's quarter of the product over the last chunks
      fd_cur_product<R, M0, NC>(cu, vs, red, tid - 256);
      fd_signal(&sy[FD_SY_F]);
      PD_STAMP(5);
      if (ln < LOOKAHEAD_MAX_AGENTS && in) {
        // the books (advance_agent, accelerated): every workgroup keeps its own copy of the Nesterov scalars (the same
        // arithmetic on the same values in every one of them); the next iteration reads it behind the grid hand-off
        const int k = ln;
        const double Nr = (double)num_robots;
        if (k < num_agents) {
          NestState s2 = nsl[k];
          const bool restart = ((s2.iter + 2) % restart_interval) == 0;
          if (restart) { s2.gamma = 0; s2.alpha = 0; }
          else {
            s2.gamma = (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * s2.gamma * s2.gamma)) / (2.0 * Nr);
            s2.alpha = 1.0 / (s2.gamma * Nr);
          }
          s2.iter += 1;
          nsl_next[k] = s2;
        }
      }
    } else {
      // ---- wave 7: look-ahead of the other agents' poses (k_step_fd's wave 7)
      FdCf er;
      fd_cf_request(ag, ln, er);
      FdCur<R, NC> cu;
      pd_cur_request<R, M0, NC>(ag, pacc_lds, bx, tid - 256, cu);
      // (the coefficients go to LDS as soon as they are here -- the chain waits for them --, the look-ahead operands are
      // requested behind that)
#if DPGO_FD_E_EARLY
      __builtin_amdgcn_sched_barrier(0);
      fd_cf_to_lds<R>(ag, ln, er, Es);
      fd_signal(&sy[FD_SY_E]);
      __builtin_amdgcn_sched_barrier(0);
#endif
      int pre[LOOKAHEAD_MAX_AGENTS + 1];
      pre[0] = 0;
#pragma unroll
      for (int k = 0; k < LOOKAHEAD_MAX_AGENTS; ++k) pre[k + 1] = pre[k] + ((k < num_agents) ? fb.npose[k] : 0);
      const int total = pre[LOOKAHEAD_MAX_AGENTS] - n;
      const int per = (total + nblk_all - 1) / nblk_all;  // <= 64, checked by the host
      const int q = bx * per + ln;
      const bool lact = ln < per && q < total;
      int self_lo = 0;
#pragma unroll
      for (int k = 0; k < LOOKAHEAD_MAX_AGENTS; ++k) if (k == sel) self_lo = pre[k];
      const int gq = lact ? (q < self_lo ? q : q + n) : 0;
      int a = 0, lo = 0, na_ = fb.npose[0];
      const double *ya = fb.ybase[0];
#pragma unroll
      for (int k = 1; k < LOOKAHEAD_MAX_AGENTS; ++k)
        if (k < num_agents && gq >= pre[k]) { a = k; lo = pre[k]; ya = fb.ybase[k]; na_ = fb.npose[k]; }
      // (the work vectors of an agent are one allocation, NBUF x (4r n): everything from its Y array)
      const size_t vlen = (size_t)4 * R * na_;
      double *yp = const_cast<double *>(ya);
      double *xa = yp - (size_t)(B_Y - B_X) * vlen, *va = yp + (size_t)(B_V - B_Y) * vlen;
      const size_t da = (size_t)B_ALT * vlen;
      const double *xr = parity ? xa + da : xa, *yr = parity ? yp + da : yp;
      double *oX = parity ? xa : xa + da, *oY = parity ? yp : yp + da, *oV = va;
      const int la_pose = gq - lo;
      const bool la_opt = next_sel == a;
      const size_t o = (size_t)la_pose * 4 * R;
      double la_x[4 * R], la_v[4 * R];
#pragma unroll
      for (int i = 0; i < 2 * R; ++i) {
        la_x[2 * i] = ldc(xr + o + 2 * i); la_x[2 * i + 1] = ldc(xr + o + 2 * i + 1);
        la_v[2 * i] = ldc(va + o + 2 * i); la_v[2 * i + 1] = ldc(va + o + 2 * i + 1);
      }
      const NestState ns = nsl[sel];
      fd_signal_requested(&sy[FD_SY_RQ]);
      PD_STAMP(10);
      __builtin_amdgcn_s_setprio(3);
#if !DPGO_FD_E_EARLY
      fd_cf_to_lds<R>(ag, ln, er, Es);
      fd_signal(&sy[FD_SY_E]);
#endif
      fd_wait(&sy[FD_SY_D], 2 * ep);  // this wave's quarter of the product over the last chunks
      fd_cur_product<R, M0, NC>(cu, vs, red, tid - 256);
      fd_signal(&sy[FD_SY_F]);
      PD_STAMP(5);
      __builtin_amdgcn_s_setprio(0);
      const FdNest nn = fd_nest(ns, num_robots, restart_interval);
      if (lact) {
        const bool st = in, lastat = in && (flags & FD_LASTAT) != 0;
        double *pa_ = fb.part[0];
#pragma unroll
        for (int k = 1; k < LOOKAHEAD_MAX_AGENTS; ++k) pa_ = (a == k) ? fb.part[k] : pa_;
        if (lastat) {
          double *xprev = yp - (size_t)(B_Y - B_XPREV) * vlen;
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) st_c(xprev + o + i, la_x[i]);
        }
        if (nn.restart_next) {
          // (X stays; Y = V = X unless the agent optimizes next -- then Y stays too: both are carried into the other copy)
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) {
            if (st) st_c(oX + o + i, la_x[i]);
            if (!la_opt) { if (st) { st_c(oY + o + i, la_x[i]); st_c(oV + o + i, la_x[i]); } la_v[i] = la_x[i]; }
            else if (st) st_c(oY + o + i, ldc(yr + o + i));
          }
          if (lastat && !la_opt) gp(pa_)[PART_D + la_pose] = 0.0;
        } else {
          double y[4 * R];
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) y[i] = (1.0 - nn.ahead_alpha) * la_x[i] + nn.ahead_alpha * la_v[i];
          polar_inplace<R>(y);
          if (lastat && !la_opt) {
            double r2 = 0;
#pragma unroll
            for (int i = 0; i < 4 * R; ++i) { const double d = y[i] - la_x[i]; r2 += d * d; }
            gp(pa_)[PART_D + la_pose] = r2;
          }
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) { if (st) { st_c(oY + o + i, y[i]); st_c(oX + o + i, y[i]); } la_x[i] = y[i]; }
        }
        if ((flags & FD_Y) && a == next3_sel) {
          // Y: the point the agent of iteration k+3 will be evaluated at -- what the look-ahead waves of the next two launches
          // will leave in its X array (the same expressions on the same operands: bitwise), formed two launches early.  The
          // agent rests in k+1 and k+2; la_x / la_v hold its X and V after iteration k+1 here.
          if (nn.restart_next2) {
#pragma unroll
            for (int i = 0; i < 4 * R; ++i) la_v[i] = la_x[i];  // (k+2 restarts: X stays, V = Y = X)
          } else {
            double y[4 * R];
#pragma unroll
            for (int i = 0; i < 4 * R; ++i) y[i] = (1.0 - nn.ahead2_alpha) * la_x[i] + nn.ahead2_alpha * la_v[i];
            polar_inplace<R>(y);
#pragma unroll
            for (int i = 0; i < 4 * R; ++i) la_x[i] = y[i];
          }
          if (!nn.restart_next3) {
            double y[4 * R];
#pragma unroll
            for (int i = 0; i < 4 * R; ++i) y[i] = (1.0 - nn.ahead3_alpha) * la_x[i] + nn.ahead3_alpha * la_v[i];
            polar_inplace<R>(y);
#pragma unroll
            for (int i = 0; i < 4 * R; ++i) la_x[i] = y[i];
          }
          double *py3 = yp + (size_t)(B_CARRY_Y - B_Y) * (long long)vlen;
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) st_c(py3 + o + i, la_x[i]);
        }
      }
    }
    PD_STAMP(14);
    if (it + 1 < K) {
      if (!pd_grid_sync(gb)) return;  // (the hand-off timed out: the error word is raised)
      PD_STAMP(15);
      if (tid < LOOKAHEAD_MAX_AGENTS) nsl[tid] = nsl_next[tid];
      __syncthreads();
    }
  }
#ifdef DPGO_PD_TRACE
  if (hb == 100 && (tid0 & 63) == 0)
    for (int k_ = 0; k_ < 16; ++k_) agents[0].part[PART_E + 4000 * PART_STRIDE + (tid0 >> 6) * 16 + k_] = (double)pd_stamps[(tid0 >> 6) * 16 + k_];
#endif
  // the books of the whole run: the Nesterov scalars into the buffer the launch behind this one reads, the team's counters
  __syncthreads();
  if (bx == 0 && tid0 < num_agents) nest_dst[tid0] = nsl_next[tid0];
  if (bx == 0 && tid0 == 0) {
    team->iter += K;
    team->stats_sel = sched[(it0 + K - 1) % sched_len];
    team->next_sel = sched[(it0 + K) % sched_len];
    team->cur_sel = team->next_sel;
  }
}

// K deep-carried iterations from the state k_nest_pre + k_fd_prime leave, in one launch.  bar: >= 18 * 16 zeroed 64-bit words
void launch_step_pd(const LaunchCtx &c, int m0, const int *d_sched, int sched_len, int it0, int K, int B, int L, double step, int num_robots,
                    int restart_interval, const NestState *nest_src, NestState *nest_dst, unsigned long long *bar, int *err) {
  int nblk_all = 0;
  for (int k = 0; k < c.num_agents; ++k) nblk_all = std::max(nblk_all, (c.host_agents[k].N4 + 7) / 8);
  const int grid = (nblk_all + 7) / 8 * 8;
  FeBases fb = {};
  for (int k = 0; k < c.num_agents && k < LOOKAHEAD_MAX_AGENTS; ++k) { fb.ybase[k] = c.host_agents[k].buf[B_Y]; fb.npose[k] = c.host_agents[k].n; fb.part[k] = c.host_agents[k].part; }
  if (m0 != 24) return;
  switch (c.r) {
    case 3: hipLaunchKernelGGL((k_step_pd<3, 24>), dim3(grid), dim3(512), 0, c.stream, c.agents, c.team, d_sched, sched_len, it0, K, B, L, step, num_robots, restart_interval, nest_src, nest_dst, fb, nblk_all, c.num_agents, bar, err); break;
    case 4: hipLaunchKernelGGL((k_step_pd<4, 24>), dim3(grid), dim3(512), 0, c.stream, c.agents, c.team, d_sched, sched_len, it0, K, B, L, step, num_robots, restart_interval, nest_src, nest_dst, fb, nblk_all, c.num_agents, bar, err); break;
    case 5: hipLaunchKernelGGL((k_step_pd<5, 24>), dim3(grid), dim3(512), 0, c.stream, c.agents, c.team, d_sched, sched_len, it0, K, B, L, step, num_robots, restart_interval, nest_src, nest_dst, fb, nblk_all, c.num_agents, bar, err); break;
    default: break;
  }
}

}  // namespace dpgo
