// step_deep.hip -- the one-launch accelerated-RGD iteration with the stream taken OFF its critical path ("deep carry",
// round 6).  k_step_fe<R, 0> (step_fused.hip) removed every dependent round trip but one
// from the front of the launch; what was left of its 14.2 us was a chain: cold round trip 3.4 -> the 128 KB slab of
// M = (Q + shift I)^-1 per CU 4.9 -> product 1.1 -> per-pose tail 3.0 -> launch gap 2.2.  The slab is STATIC data and
// 80 % of the vector it meets is known a launch early, so the chain need not contain it:
//
//   * the rows of the gradient of agent c that belong to PRIVATE poses (no shared edge) are the tangent projection of the
//     carried row products -- complete two launches before c's turn (B_CARRY_G, formed from the evaluation point the
//     look-ahead wave left three launches before).  Only the public poses (the first and last 50 of a 500-pose agent of
//     sphere2500 / 5) need the neighbours' poses of the launch before.
//   * a column of the product  z[col] = sum_k M[k, col] g[k]  is accumulated by 32 lanes, lane kl taking rows
//     2 kl + 64 m (+ 1) for m = 0 .. 31 in one fixed order of the 64-row chunks m: the agent's chunk order (AgentDev::fe_ord,
//     private chunks first; k_precond follows the same order).  The first M0 chunks of every agent are private.  So launch
//     k - 1 streams those M0 chunks of M_c against the carried gradient and leaves the 256 x R partial sums of every
//     workgroup (PACC); launch k loads them and continues the SAME chain of fused multiply-adds over the remaining
//     32 - M0 chunks: bitwise the sums k_precond forms in one go.
//
// Launch k (agent c = sel(k); d = sel(k+1), e = sel(k+2), f = sel(k+3); four different agents in a row):
//     IN  finish the gradient of c's public poses, continue PACC(c) over the last 32 - M0 chunks, step c's poses
//         (tangent projection, QF retraction, Nesterov V, look-ahead Y), look-ahead of everybody else
//     P   stream the M0 private chunks of M_d against carry-G(d): PACC(d)                    (consumed by launch k + 1)
//     W   row products of e at its evaluation point, their tangent projection: carry-W/X/G(e) (k + 1: P, k + 2: IN)
//     Y   the point f will be evaluated at: three look-ahead maps in a row, carry-Y(f)        (k + 1: W)
// A run opens with k_fd_prime (the points of sel(0), sel(1), sel(2)), k_fd_rows (W for sel(0) and sel(1)) and k_fd_partial
// (P for sel(0)): three short launches.
//
// Eight waves, eight roles, ONE workgroup barrier.  A wave that requests the stream stays at the issue of its loads until
// most of them have landed (step_fused.hip), and every s_barrier behind that point waits for it -- the critical chain
// (gradient of the public poses -> product over the last chunks -> tail) must not meet the streamers again.  So the only
// s_barrier stands at the kernel's head (it zeroes the counters); every hand-off is a counter in LDS that only the waves
// concerned wait at -- the first of them (RQ) orders the requests: waves 4-7 count themselves in behind their last load, the
// streamers release the stream when all four have (everything the chain needs is in the CU's memory queue in front of it):
//     waves 0-3  streamers: the carried gradient of c's last chunks -> LDS (C); then the whole P part on their own (N: their
//                own hand-off; they start their product behind F, the chain's -- the two share the LDS)
//     waves 4-5  the chain: one public pose per lane (G_j from LDS, projection; rows -> LDS: D) -> their quarters of the
//                product over the last chunks (F) -> wave 4: reduction, step of the two poses on 16 lanes each (as
//                k_step_fe); wave 5: W
//     wave 6     neighbour poses of the shared edges -> LDS (E); its quarter of the product behind D (F); the books
//     wave 7     coefficients of the shared edges -> LDS (E); its quarter of the product (F); look-ahead of the other
//                agents' poses + Y
//   Waves 4-7 run at raised priority while they are on the chain.
// Where a launch's 12.6 us go (profiles/r06_deep_carry.md): 2.8 until everything is requested (134 KB per CU in front of
// the stream -- every workgroup finishes ALL public poses itself, the price of no exchange inside the launch), 1.0 until the
// edges' operands have landed, 1.9 gradient of the public poses, 0.9 product over the last chunks, 0.5 reduction, 3.2 tail,
// 2.2 from the last workgroup's end to the next launch's first instruction.  The stream (128 KB per CU) lands under all
// of it: the streamers are done 3 us before the tail.
// Same arithmetic on the same operands in the same order as k_step_fe / the two-launch sequence: the iterates are BITWISE
// theirs (tests/test_gpu_fused_step.py, profiles/experiments/fe_fuzz.py).
#include "kernel_common.h"
#include "step_deep_dev.h"
#include <algorithm>

namespace dpgo {

#ifdef DPGO_FE_TRACE
#ifndef DPGO_FE_TRACE_BLOCK
#define DPGO_FE_TRACE_BLOCK 100
#endif
#define FD_TRACE_DECL __shared__ unsigned long long fd_stamps[8 * 16];
#ifndef DPGO_FD_STAMP_MASK
#define DPGO_FD_STAMP_MASK 0xFFFF
#endif
#define FD_STAMP(k) do { if (((DPGO_FD_STAMP_MASK >> (k)) & 1) && (threadIdx.x & 63) == 0) fd_stamps[(threadIdx.x >> 6) * 16 + (k)] = wall_clock64(); } while (0)
#define FD_FLUSH() do { if ((threadIdx.x & 63) == 0) { const int w_ = threadIdx.x >> 6; \
    if (blockIdx.x == DPGO_FE_TRACE_BLOCK) for (int k_ = 0; k_ < 16; ++k_) ag.part[PART_E + 4000 * PART_STRIDE + w_ * 16 + k_] = (double)fd_stamps[w_ * 16 + k_]; \
    if (w_ == 4) { ag.part[PART_E + (4100 + 2 * (int)blockIdx.x) * PART_STRIDE] = (double)fd_stamps[4 * 16]; ag.part[PART_E + (4100 + 2 * (int)blockIdx.x) * PART_STRIDE + 1] = (double)fd_stamps[4 * 16 + 15]; } \
    if (w_ == 0) { ag.part[PART_E + (4100 + 2 * (int)blockIdx.x) * PART_STRIDE + 2] = (double)fd_stamps[15]; } } } while (0)
#elif defined(DPGO_FD_ENDS)
// -DDPGO_FD_ENDS: nothing but the time every wave of workgroup 100 leaves, straight to memory (one store per wave: the
// build runs within 0.1 us of the product build) -- PART_E words [4000 + 16 w + 15]; wave 4 also leaves its start [64]
#define FD_TRACE_DECL
#define FD_STAMP(k) do { if ((k) == 0 && blockIdx.x == 100 && threadIdx.x == 256) ag.part[PART_E + 4000 * PART_STRIDE + 64] = (double)wall_clock64(); } while (0)
#define FD_FLUSH() do { if (blockIdx.x == 100 && (threadIdx.x & 63) == 0) ag.part[PART_E + 4000 * PART_STRIDE + (threadIdx.x >> 6) * 16 + 15] = (double)wall_clock64(); } while (0)
#else
#define FD_TRACE_DECL
#define FD_STAMP(k) do { } while (0)
#define FD_FLUSH() do { } while (0)
#endif

template <int R, int M0>
__global__ __launch_bounds__(512) void k_step_fd(TeamDev *team, int sel, int next_sel, double step, int num_robots, int restart_interval,
                                                 const NestState *nest_src, NestState *nest_dst, int parity, const AgentDev agv,
                                                 int next3_sel, int flags, const FdNext nx, const FeBases fb,
                                                 const double *__restrict__ pacc_in, double *__restrict__ pacc_out, int nblk_all,
                                                 int num_agents) {
  const AgentDev &ag = agv;
#if DPGO_FD_KA_PREFETCH
  {
    // The launch's arguments (1.5 KB: the agent's descriptor by value, what it needs of the next two) sit in memory nobody
    // has touched: fetched where they are first used, every scalar load misses in turn (traced: the wave that needs the
    // edges' codes issued its first request 2.7 us into the launch).  One word of every 64-byte line now, all in flight at
    // once: what follows hits the scalar cache.
    const unsigned *ka = (const unsigned *)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int KA_LINES = (int)((sizeof(AgentDev) + sizeof(FdNext) + sizeof(FeBases) + 160 + 63) / 64);
    unsigned touch = 0;
#pragma unroll
    for (int i = 0; i < KA_LINES; ++i) touch |= ka[16 * i];
    asm volatile("" ::"s"(touch));
  }
#endif
  // the poses live twice (step_fused.hip): this launch reads the copy of its parity and writes the other one
  const double *__restrict__ Xr = ag.buf[parity ? B_XALT : B_X];
  const double *__restrict__ Yr = ag.buf[parity ? B_YALT : B_Y];
  double *__restrict__ Xw = ag.buf[parity ? B_X : B_XALT];
  double *__restrict__ Yw = ag.buf[parity ? B_Y : B_YALT];
  const int hb = (int)blockIdx.x;
  const int bx = (hb % 8) * ((int)gridDim.x / 8) + hb / 8;  // XCD-aware block order, as in k_precond
  const int tid = threadIdx.x;
  const int N4 = ag.N4, n = ag.n;
  const int nblk = (N4 + 7) / 8;
  if (bx >= nblk_all) return;
  // (the grid covers the largest agent of the team: a workgroup beyond this agent's columns still takes its share of the
  // other agents' work)
  const bool own = bx < nblk, in = (flags & FD_IN) != 0;
  constexpr int KC = FD_KC, MREG = KC / 64, NC = MREG - M0;
  __shared__ double vs[R * KC];  // chunk-ordered: [0, M0 * 64 R) the private rows of the NEXT agent's carried gradient, behind
                                 // them the rows of THIS agent's gradient that the last NC chunks meet
  __shared__ double zs[8 * R];
  __shared__ double red[32 * (8 * R + 1)];
  __shared__ double Ysh[2 * 4 * R];
  __shared__ double Esh[2][2 * 4 * R];
  __shared__ double Es[FE_MAX_EDGES * (4 * R + 16)];
  __shared__ double tl_x[2 * 4 * R], tl_v[2 * 4 * R], tl_y[2 * 4 * R], tl_s[2 * 16];
  __shared__ double Ex[2 * 3 * 4 * R];
  __shared__ double Psh[2 * 4 * R], tl_rel[2];  // XPrev of the two poses, their |X - XPrev|^2 (steps that leave statistics)
  __shared__ int sy[FD_SY_COUNT];
  FD_TRACE_DECL
  FD_STAMP(0);
  if (tid < FD_SY_COUNT) sy[tid] = 0;
  lds_barrier();  // Z: the counters are zero (the only workgroup barrier; every wave is here within its first instructions)
  const int pj0 = own ? 2 * bx : 0, pj1 = (own && 2 * bx + 1 < n) ? 2 * bx + 1 : -1;
  const int npose = own ? ((pj1 >= 0) ? 2 : 1) : 0;
  constexpr int EPE = 4 * R + 16;
  const int cwv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;

  if (cwv < 4) {
    // ================================================================ streamers
    const int cg = (tid >> 5) & 7, kl = tid & 31;
    for (int t = N4 * R + tid; t < KC * R; t += 256) vs[t] = 0.0;  // rows beyond the agent's
    // the rows of this agent's carried gradient that the last NC chunks meet: positions [M0 * 64 R, N4 R)
    constexpr int NGC = ((KC - 64 * M0) * R / 2 + 255) / 256;
    double2 gc[NGC];
#if !DPGO_FD_GC_LATE
    {
      const double *Gc = ag.buf[B_CARRY_G];
#pragma unroll
      for (int u = 0; u < NGC; ++u) gc[u] = ld2(Gc + min(M0 * 64 * R + 2 * (tid + 256 * u), N4 * R - 2));
    }
#endif
    const int cold = 8 * bx + cg;
    const int N4d = nx.N4d;
    const double *Md = nx.Md + (size_t)((cold < N4d) ? cold : 0) * N4d;
    double2 mn[M0];
#pragma unroll
    for (int i = 0; i < FD_HEAD; ++i) mn[i] = ld2_nt(Md + min(2 * kl + 64 * (int)nx.ord_d[i], N4d - 2));
    FD_STAMP(10);
    __builtin_amdgcn_sched_barrier(0);
#if DPGO_FD_GC_LATE
    {
      const double *Gc = ag.buf[B_CARRY_G];
#pragma unroll
      for (int u = 0; u < NGC; ++u) gc[u] = ld2(Gc + min(M0 * 64 * R + 2 * (tid + 256 * u), N4 * R - 2));
    }
#endif
#pragma unroll
    for (int u = 0; u < NGC; ++u) {
      const int tt = M0 * 64 * R + 2 * (tid + 256 * u);
      if (tt < N4 * R) *reinterpret_cast<double2 *>(&vs[tt]) = gc[u];
    }
    fd_signal(&sy[FD_SY_C]);
    FD_STAMP(1);
    fd_wait(&sy[FD_SY_RQ], 4);  // A: waves 4-7 have requested all they need -- the stream queues behind it, not in front
    FD_STAMP(11);
    // (nothing of the stream is requested in front of this hand-off: a wave stays at the issue of such loads, and the chain
    // waits for C -- the scheduler otherwise hoists the requests above the LDS writes)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    // ---- P: the private chunks of the next agent's M against the private rows of its carried gradient
    constexpr int NGN = (M0 * 64 * R / 2 + 255) / 256;
    double2 gn[NGN];
#pragma unroll
    for (int u = 0; u < NGN; ++u) gn[u] = ld2(nx.Gd + min(2 * (tid + 256 * u), M0 * 64 * R - 2));
#pragma unroll
    for (int i = FD_HEAD; i < M0; ++i) mn[i] = ld2_nt(Md + min(2 * kl + 64 * (int)nx.ord_d[i], N4d - 2));
    FD_STAMP(12);
#pragma unroll
    for (int u = 0; u < NGN; ++u) {
      const int tt = 2 * (tid + 256 * u);
      if (tt < M0 * 64 * R) *reinterpret_cast<double2 *>(&vs[tt]) = gn[u];
    }
    fd_signal(&sy[FD_SY_N]);
    fd_wait(&sy[FD_SY_N], 4);
    fd_wait(&sy[FD_SY_F], 4);  // (the chain's product first: the two share the LDS)
    FD_STAMP(13);
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
    if ((flags & FD_P) && bx < nx.nblk_d) {
#pragma unroll
#if DPGO_FD_PACC_WT
      for (int a = 0; a < R; ++a) st_c(pacc_out + ((size_t)bx * R + a) * 256 + tid, acc[a]);  // (write-through: not left dirty in L2 for the kernel boundary)
#else
      for (int a = 0; a < R; ++a) gp(pacc_out)[((size_t)bx * R + a) * 256 + tid] = acc[a];
#endif
    }
    FD_STAMP(15);
    FD_FLUSH();
    return;
  }

  if (cwv < 6) {
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
        const double2 tw = ld2(Wc + ((size_t)i * npub + pqc) * 2), tx = ld2(Xc + ((size_t)i * npub + pqc) * 2);
        w[2 * i] = tw.x; w[2 * i + 1] = tw.y; x[2 * i] = tx.x; x[2 * i + 1] = tx.y;
      }
    }
    FD_STAMP(14);
    FdCur<R, NC> cu;
    fd_cur_request<R, M0, NC>(ag, pacc_in, bx, nblk, tid - 256, cu);
    // W (wave 5, behind its quarter of the product): the indices of the rows now (wave 4's copies are never used)
    const int npw = (nx.n_e + nblk_all - 1) / nblk_all;  // (<= 2)
    const int wls = ln / (4 * R), we = ln - wls * (4 * R);
    const int pw = bx * npw + wls;
    const bool pv = wls < 3 && wls < npw && pw < nx.n_e;
    const int pwc = pv ? pw : 0;
    const int wtile = pwc >> 6, wpl = pwc & 63, wdn = nx.soa_w_e;
    int ii[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ii[u] = gp(nx.soa_col_e)[((size_t)wtile * wdn + min(u, wdn - 1)) * 64 + wpl];
    const int qi = gp(nx.pub_index_e)[pwc];
    // operands of the tail (the lanes of wave 4 that will hold them; wave 5's copies are never used)
    const int tl = tid - 256 - 64 * g;
    const size_t own_off = (size_t)((tl >= 4 * R) ? max(pj1, 0) : pj0) * 4 * R + (size_t)(tl % (4 * R));
    double pre_x = 0, pre_v = 0, pre_y = 0, pre_p = 0;
    if (tl < npose * 4 * R) {
      pre_x = gp(Xr)[own_off];
      pre_v = gp(ag.buf[B_V])[own_off];
      pre_y = gp(Yr)[own_off];
      pre_p = gp(ag.buf[B_XPREV])[own_off];
    }
    const NestState ns = nest_src[sel];
    __builtin_amdgcn_s_setprio(3);  // (the chain's instructions go first: the streamers' product shares the LDS with it)
    FD_STAMP(10);
    fd_signal_requested(&sy[FD_SY_RQ]);
    FD_STAMP(11);
    const int vs_off = fd_pos_off<R>(ag.fe_ord, pj);  // (where the pose's row goes: looked up while the edges' operands land)
    fd_wait(&sy[FD_SY_E], 2);  // the operands of the shared edges are in LDS
    FD_STAMP(1);
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
    FD_STAMP(3);
    fd_wait(&sy[FD_SY_C], 4);  // the carried rows are in LDS: the public ones are overwritten now
    FD_STAMP(2);
    if (pact) {
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) vs[vs_off + i] = w[i];
    }
    FD_STAMP(12);
    fd_signal(&sy[FD_SY_D]);
    FD_STAMP(13);
    fd_wait(&sy[FD_SY_D], 2);
    FD_STAMP(4);
    fd_cur_product<R, M0, NC>(cu, vs, red, tid - 256);
    if (tl >= 0 && tl < npose * 4 * R) { Ysh[tl] = pre_x; Esh[0][tl] = pre_v; Esh[1][tl] = pre_y; Psh[tl] = pre_p; }
    FD_STAMP(5);
    fd_signal(&sy[FD_SY_F]);
    if (g == 1) {
      // ---- W: the row products of agent e at the point B_CARRY_Y holds (left complete by the previous launch), for this
      // workgroup's share of its poses -- one (pose, entry) per lane, fe_block's expression slot after slot (bitwise the sums
      // a self-forming launch would make), then their tangent projection at the point
      __builtin_amdgcn_s_setprio(0);
      const int c = we / R, a = we - c * R;
      const double *__restrict__ Y2 = nx.Ye;
      double xv[8][4], bv[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double *xp = Y2 + (size_t)4 * R * ii[u] + a;
        const double *bp = nx.soa_val_e + ((size_t)wtile * wdn + min(u, wdn - 1)) * 1024 + (2 * c) * 128 + 2 * wpl;
        xv[u][0] = gp(xp)[0]; xv[u][1] = gp(xp)[R]; xv[u][2] = gp(xp)[2 * R]; xv[u][3] = gp(xp)[3 * R];
        bv[u][0] = gp(bp)[0]; bv[u][1] = gp(bp)[1]; bv[u][2] = gp(bp)[128]; bv[u][3] = gp(bp)[129];
      }
      const double xe_ = gp(Y2)[(size_t)4 * R * pwc + we];
      double acc = 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double t = fma4(xv[u][0], bv[u][0], xv[u][1], bv[u][1], xv[u][2], bv[u][2], xv[u][3], bv[u][3], acc);
        acc = (u < wdn) ? t : acc;
      }
      const bool wr = pv && (flags & FD_W);
      if (wr && qi >= 0) {  // (a public pose: its launch finishes it -- row product and point, [entry][public pose])
        gp(nx.We)[((size_t)(we >> 1) * nx.npub_e + qi) * 2 + (we & 1)] = acc;
        gp(nx.Xe)[((size_t)(we >> 1) * nx.npub_e + qi) * 2 + (we & 1)] = xe_;
      }
      if (wls < 3) { Ex[wls * 4 * R + we] = acc; Ex[3 * 4 * R + wls * 4 * R + we] = xe_; }
      WSYNC();
      if (wr && we == 0) {
        double ww[4 * R], xx[4 * R];
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) { ww[i] = Ex[wls * 4 * R + i]; xx[i] = Ex[3 * 4 * R + wls * 4 * R + i]; }
        tangent_inplace<R>(xx, ww);
        double *Gn = nx.Ge + fd_pos_off<R>(nx.ord_e, pw);  // (chunk-ordered: the launches that consume it copy straight ranges)
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) gp(Gn)[i] = ww[i];
      }
      FD_STAMP(14);
      FD_FLUSH();
      return;
    }
    fd_wait(&sy[FD_SY_F], 4);
    if (ln < 8 * R) {
      double s = 0;
#pragma unroll
      for (int q = 0; q < 32; ++q) s += red[q * (8 * R + 1) + ln];
      zs[ln] = s;
    }
    FD_STAMP(6);
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
        FD_STAMP(7);
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
          if (h0) gp(ag.buf[B_XPREV])[o + i0] = xs[i0];
          if (h1) gp(ag.buf[B_XPREV])[o + i1] = xs[i1];
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
        FD_STAMP(8);
        if (restart_next) {
          if (h0 && in) {
            Xw[o + i0] = xs[i0];
            if (!ahead_opt) { Yw[o + i0] = xs[i0]; vsv[i0] = xs[i0]; }
            else Yw[o + i0] = reset ? xs[i0] : y0[i0];
          }
          if (h1 && in) {
            Xw[o + i1] = xs[i1];
            if (!ahead_opt) { Yw[o + i1] = xs[i1]; vsv[i1] = xs[i1]; }
            else Yw[o + i1] = reset ? xs[i1] : y0[i1];
          }
          if (lastat && !ahead_opt && s == 0) gp(ag.part)[PART_D + (lp ? pj1 : pj0)] = 0.0;
        } else {
          if (h0) ys[i0] = (1.0 - ahead_alpha) * xs[i0] + ahead_alpha * vsv[i0];
          if (h1) ys[i1] = (1.0 - ahead_alpha) * xs[i1] + ahead_alpha * vsv[i1];
          lanes_sync();
          polar_lanes<R>(ys, Ss, s);
          FD_STAMP(9);
          if (h0 && in) { Yw[o + i0] = ys[i0]; Xw[o + i0] = ys[i0]; }
          if (h1 && in) { Yw[o + i1] = ys[i1]; Xw[o + i1] = ys[i1]; }
          if (lastat && !ahead_opt && s == 0) {  // look-ahead steps leave |Y' - X|^2 per pose
            double rel2 = 0;
#pragma unroll
            for (int i = 0; i < 4 * R; ++i) { const double d = ys[i] - xs[i]; rel2 += d * d; }
            gp(ag.part)[PART_D + (lp ? pj1 : pj0)] = rel2;
          }
        }
        lanes_sync();
        if (h0 && in) ag.buf[B_V][o + i0] = vsv[i0];
        if (h1 && in) ag.buf[B_V][o + i1] = vsv[i1];
      }
    }
    if (stats) {
      // (k_precond: the two poses' sums through wave_sum, lane 0 stores)
      WSYNC();
      double rl = (ln < npose) ? tl_rel[ln] : 0.0;
      rl = wave_sum(rl);
      if (ln == 0 && own) gp(ag.part)[PART_B + (size_t)bx * PART_STRIDE + 2] = rl;
    }
    FD_STAMP(15);
    FD_FLUSH();
    return;
  }

  // ================================================================ waves 6 and 7: the operands of the shared edges into LDS,
  // then the row products of the agent two iterations ahead (6) / the look-ahead of the other agents' poses (7)
  const int h = cwv - 6;
  if (h == 0) {
    FdXn<R> er;
    fd_xn_request<R>(ag, fb, parity, ln, er);
    FD_STAMP(9);
    FdCur<R, NC> cu;
    fd_cur_request<R, M0, NC>(ag, pacc_in, bx, nblk, tid - 256, cu);
    FD_STAMP(10);
    fd_signal_requested(&sy[FD_SY_RQ]);
    FD_STAMP(11);
    __builtin_amdgcn_s_setprio(3);
    fd_xn_to_lds<R>(ag, ln, er, Es);
    fd_signal(&sy[FD_SY_E]);
    FD_STAMP(8);
    fd_wait(&sy[FD_SY_D], 2);  // this wave's quarter of the product over the last chunks
    fd_cur_product<R, M0, NC>(cu, vs, red, tid - 256);
    fd_signal(&sy[FD_SY_F]);
    if (bx == 0 && ln < LOOKAHEAD_MAX_AGENTS && in) {
      // the books (advance_agent, accelerated) into the OTHER state buffer: every workgroup of this launch reads nest_src
      const int k = ln;
      const double Nr = (double)num_robots;
      if (k < num_agents) {
        NestState s2 = nest_src[k];
        const bool restart = ((s2.iter + 2) % restart_interval) == 0;
        if (restart) { s2.gamma = 0; s2.alpha = 0; }
        else {
          s2.gamma = (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * s2.gamma * s2.gamma)) / (2.0 * Nr);
          s2.alpha = 1.0 / (s2.gamma * Nr);
        }
        s2.iter += 1;
        nest_dst[k] = s2;
      }
      if (k == 0) {
        team->iter += 1;
        team->stats_sel = sel;
        team->next_sel = next_sel;
        team->cur_sel = next_sel;
      }
    }
    FD_STAMP(14);
    FD_FLUSH();
    return;
  }

  // ---- wave 7: look-ahead Nesterov step of iteration k+1 for this workgroup's share of the OTHER agents' poses (one lane
  // per pose, k_step_fe's second wave), every address from the launch's arguments
  {
    FdCf er;
    fd_cf_request(ag, ln, er);
    FD_STAMP(9);
    FdCur<R, NC> cu;
    fd_cur_request<R, M0, NC>(ag, pacc_in, bx, nblk, tid - 256, cu);
    // (the coefficients go to LDS as soon as they are here -- the chain waits for them --, the look-ahead operands are
    // requested behind that)
#if DPGO_FD_E_EARLY
    __builtin_amdgcn_sched_barrier(0);
    fd_cf_to_lds<R>(ag, ln, er, Es);
    fd_signal(&sy[FD_SY_E]);
    FD_STAMP(8);
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
      const double2 tx = ld2(xr + o + 2 * i), tv = ld2(va + o + 2 * i);
      la_x[2 * i] = tx.x; la_x[2 * i + 1] = tx.y; la_v[2 * i] = tv.x; la_v[2 * i + 1] = tv.y;
    }
    const NestState ns = nest_src[sel];
    FD_STAMP(10);
    fd_signal_requested(&sy[FD_SY_RQ]);
    FD_STAMP(11);
    __builtin_amdgcn_s_setprio(3);
#if !DPGO_FD_E_EARLY
    fd_cf_to_lds<R>(ag, ln, er, Es);
    fd_signal(&sy[FD_SY_E]);
    FD_STAMP(8);
#endif
    fd_wait(&sy[FD_SY_D], 2);  // this wave's quarter of the product over the last chunks
    fd_cur_product<R, M0, NC>(cu, vs, red, tid - 256);
    fd_signal(&sy[FD_SY_F]);
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
        for (int i = 0; i < 4 * R; ++i) gp(xprev)[o + i] = la_x[i];
      }
      if (nn.restart_next) {
        // (X stays; Y = V = X unless the agent optimizes next -- then Y stays too: both are carried into the other copy)
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) {
          if (st) oX[o + i] = la_x[i];
          if (!la_opt) { if (st) { oY[o + i] = la_x[i]; oV[o + i] = la_x[i]; } la_v[i] = la_x[i]; }
          else if (st) oY[o + i] = yr[o + i];
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
        for (int i = 0; i < 4 * R; ++i) { if (st) { oY[o + i] = y[i]; oX[o + i] = y[i]; } la_x[i] = y[i]; }
      }
      FD_STAMP(7);
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
        for (int i = 0; i < 4 * R; ++i) gp(py3)[o + i] = la_x[i];
      }
    }
    FD_STAMP(15);
    FD_FLUSH();
  }
}

// the evaluation points of the first three agents of a run, from the state k_nest_pre leaves (every agent at its point of
// iteration 0): role 0 -- sel(0): the point itself; role 1 -- sel(1): one look-ahead map; role 2 -- sel(2): two.  The
// expressions are the look-ahead waves' (bitwise what launches 0 and 1 will leave in those agents' X arrays).
template <int R>
__global__ __launch_bounds__(64) void k_fd_prime(const AgentDev *__restrict__ agents, int s0, int s1, int s2, int num_robots,
                                                 int restart_interval, const NestState *nest_src) {
  const int role = (int)blockIdx.y;
  const int ai = role == 0 ? s0 : (role == 1 ? s1 : s2);
  const AgentDev &ag = agents[ai];
  const int j = (int)blockIdx.x * 64 + (int)threadIdx.x;
  if (j >= ag.n) return;
  const size_t o = (size_t)j * 4 * R;
  double x[4 * R], v[4 * R];
#pragma unroll
  for (int i = 0; i < 4 * R; ++i) { x[i] = gp(ag.buf[B_X])[o + i]; v[i] = gp(ag.buf[B_V])[o + i]; }
  const FdNest nn = fd_nest(nest_src[s0], num_robots, restart_interval);
  if (role == 2) {
    if (nn.restart_next) {
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) v[i] = x[i];
    } else {
      double y[4 * R];
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) y[i] = (1.0 - nn.ahead_alpha) * x[i] + nn.ahead_alpha * v[i];
      polar_inplace<R>(y);
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) x[i] = y[i];
    }
  }
  if (role >= 1) {
    const bool rs = role == 1 ? nn.restart_next : nn.restart_next2;
    const double al = role == 1 ? nn.ahead_alpha : nn.ahead2_alpha;
    if (!rs) {
      double y[4 * R];
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) y[i] = (1.0 - al) * x[i] + al * v[i];
      polar_inplace<R>(y);
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) x[i] = y[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 4 * R; ++i) gp(ag.buf[B_CARRY_Y])[o + i] = x[i];
}

// The two producers a run opens with, as launches of their own (they used to be two full k_step_fd launches whose iteration
// part computed on nothing and stored nothing: 12 us each where these take a few).
// k_fd_rows: W for the agents sel(0) and sel(1) (blockIdx.y) from the points k_fd_prime left -- the W wave of k_step_fd,
// expression for expression, one wave per workgroup's share of the agent's poses.
template <int R>
__global__ __launch_bounds__(64) void k_fd_rows(const AgentDev *__restrict__ agents, int e0, int e1, int nblk_all) {
  const AgentDev &age = agents[blockIdx.y == 0 ? e0 : e1];
  const int bx = (int)blockIdx.x, ln = (int)threadIdx.x;
  __shared__ double Ex[2 * 3 * 4 * R];
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
  const int c = we / R, a = we - c * R;
  const double *__restrict__ Y2 = age.buf[B_CARRY_Y];
  double xv[8][4], bv[8][4];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const double *xp = Y2 + (size_t)4 * R * ii[u] + a;
    const double *bp = age.soa_val + ((size_t)wtile * wdn + min(u, wdn - 1)) * 1024 + (2 * c) * 128 + 2 * wpl;
    xv[u][0] = gp(xp)[0]; xv[u][1] = gp(xp)[R]; xv[u][2] = gp(xp)[2 * R]; xv[u][3] = gp(xp)[3 * R];
    bv[u][0] = gp(bp)[0]; bv[u][1] = gp(bp)[1]; bv[u][2] = gp(bp)[128]; bv[u][3] = gp(bp)[129];
  }
  const double xe_ = gp(Y2)[(size_t)4 * R * pwc + we];
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const double t = fma4(xv[u][0], bv[u][0], xv[u][1], bv[u][1], xv[u][2], bv[u][2], xv[u][3], bv[u][3], acc);
    acc = (u < wdn) ? t : acc;
  }
  if (pv && qi >= 0) {
    gp(age.buf[B_CARRY_W])[((size_t)(we >> 1) * age.npub + qi) * 2 + (we & 1)] = acc;
    gp(age.buf[B_CARRY_X])[((size_t)(we >> 1) * age.npub + qi) * 2 + (we & 1)] = xe_;
  }
  if (wls < 3) { Ex[wls * 4 * R + we] = acc; Ex[3 * 4 * R + wls * 4 * R + we] = xe_; }
  WSYNC();
  if (pv && we == 0) {
    double ww[4 * R], xx[4 * R];
#pragma unroll
    for (int i = 0; i < 4 * R; ++i) { ww[i] = Ex[wls * 4 * R + i]; xx[i] = Ex[3 * 4 * R + wls * 4 * R + i]; }
    tangent_inplace<R>(xx, ww);
    double *Gn = age.buf[B_CARRY_G] + fd_pos_off<R>(age.fe_ord, pw);
#pragma unroll
    for (int i = 0; i < 4 * R; ++i) gp(Gn)[i] = ww[i];
  }
}

// k_fd_partial: P for the agent sel(0) -- the streamers of k_step_fd, expression for expression: the first M0 chunks of its M
// against the private rows of its carried gradient, the partial sums of every workgroup into pacc_out
template <int R, int M0>
__global__ __launch_bounds__(256) void k_fd_partial(const AgentDev *__restrict__ agents, int d, double *__restrict__ pacc_out) {
  const AgentDev &agd = agents[d];
  const int hb = (int)blockIdx.x;
  const int bx = (hb % 8) * ((int)gridDim.x / 8) + hb / 8;
  const int tid = threadIdx.x;
  const int N4d = agd.N4;
  if (bx >= (N4d + 7) / 8) return;
  __shared__ double vs[M0 * 64 * R];
  const int cg = (tid >> 5) & 7, kl = tid & 31;
  const int cold = 8 * bx + cg;
  const double *Md = agd.M + (size_t)((cold < N4d) ? cold : 0) * N4d;
  constexpr int NGN = (M0 * 64 * R / 2 + 255) / 256;
  double2 gn[NGN];
#pragma unroll
  for (int u = 0; u < NGN; ++u) gn[u] = ld2(agd.buf[B_CARRY_G] + min(2 * (tid + 256 * u), M0 * 64 * R - 2));
  double2 mn[M0];
#pragma unroll
  for (int i = 0; i < M0; ++i) mn[i] = ld2_nt(Md + min(2 * kl + 64 * (int)agd.fe_ord[i], N4d - 2));
#pragma unroll
  for (int u = 0; u < NGN; ++u) {
    const int tt = 2 * (tid + 256 * u);
    if (tt < M0 * 64 * R) *reinterpret_cast<double2 *>(&vs[tt]) = gn[u];
  }
  lds_barrier();
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
#pragma unroll
  for (int a = 0; a < R; ++a) gp(pacc_out)[((size_t)bx * R + a) * 256 + tid] = acc[a];
}

// the deep-carried form needs: every agent's first M0 chunks private, its public poses on two waves (<= 128), its shared
// edges on two waves' two slots (<= FE_MAX_EDGES, all co-resident: fe_code_ok), its rows in the lane-ordered copy
int step_fd_pick_m0(int min_private_chunks) {
  // (fewer private chunks: the waves of the chain would carry too much of the stream.  Measured, round 6: M0 = 20 spills 14
  // registers at r = 5 and runs 15.7 us per launch on sphere2500 / 5 -- slower than round 5's form (14.3); M0 = 16 spills 35:
  // 0.0182 ms per iteration on sphere2500 / 6 against 0.0142)
  const int want[] = {24};
  for (int m : want) if (min_private_chunks >= m) return m;
  return 0;
}

void launch_fd_prime(const LaunchCtx &c, int s0, int s1, int s2, int max_n, int num_robots, int restart_interval, const NestState *nest_src) {
  DPGO_DISPATCH_R(c.r, hipLaunchKernelGGL((k_fd_prime<R>), dim3((max_n + 63) / 64, 3), dim3(64), 0, c.stream, c.agents, s0, s1, s2,
                                          num_robots, restart_interval, nest_src));
}

// what a run opens with behind k_fd_prime: the row products of sel(0) and sel(1), then the private partial sums of sel(0)
void launch_fd_open(const LaunchCtx &c, int m0, int s0, int s1, double *pacc_out) {
  int nblk_all = 0;
  for (int k = 0; k < c.num_agents; ++k) nblk_all = std::max(nblk_all, (c.host_agents[k].N4 + 7) / 8);
  const int gridp = ((c.host_agents[s0].N4 + 7) / 8 + 7) / 8 * 8;
  if (m0 != 24) return;
  switch (c.r) {
    case 3: hipLaunchKernelGGL((k_fd_rows<3>), dim3(nblk_all, 2), dim3(64), 0, c.stream, c.agents, s0, s1, nblk_all);
            hipLaunchKernelGGL((k_fd_partial<3, 24>), dim3(gridp), dim3(256), 0, c.stream, c.agents, s0, pacc_out); break;
    case 4: hipLaunchKernelGGL((k_fd_rows<4>), dim3(nblk_all, 2), dim3(64), 0, c.stream, c.agents, s0, s1, nblk_all);
            hipLaunchKernelGGL((k_fd_partial<4, 24>), dim3(gridp), dim3(256), 0, c.stream, c.agents, s0, pacc_out); break;
    case 5: hipLaunchKernelGGL((k_fd_rows<5>), dim3(nblk_all, 2), dim3(64), 0, c.stream, c.agents, s0, s1, nblk_all);
            hipLaunchKernelGGL((k_fd_partial<5, 24>), dim3(gridp), dim3(256), 0, c.stream, c.agents, s0, pacc_out); break;
    default: break;
  }
}

// sel .. : agents c, d, e, f of the head of the file (d, e, f: any valid agent where the flag is off)
void launch_step_fd(const LaunchCtx &c, int m0, int sel, int next_sel, int next2_sel, int next3_sel, double step, int num_robots,
                    int restart_interval, const NestState *nest_src, NestState *nest_dst, int parity, int flags,
                    const double *pacc_in, double *pacc_out) {
  const AgentDev &d = c.host_agents[sel], &dd = c.host_agents[next_sel], &de = c.host_agents[next2_sel];
  int nblk_all = 0;
  for (int k = 0; k < c.num_agents; ++k) nblk_all = std::max(nblk_all, (c.host_agents[k].N4 + 7) / 8);
  const int grid = (nblk_all + 7) / 8 * 8;
  FeBases fb = {};
  for (int k = 0; k < c.num_agents && k < LOOKAHEAD_MAX_AGENTS; ++k) { fb.ybase[k] = c.host_agents[k].buf[B_Y]; fb.npose[k] = c.host_agents[k].n; fb.part[k] = c.host_agents[k].part; }
  FdNext nx = {};
  nx.Md = dd.M; nx.N4d = dd.N4; nx.nblk_d = (dd.N4 + 7) / 8; nx.Gd = dd.buf[B_CARRY_G];
  for (int i = 0; i < 32; ++i) { nx.ord_d[i] = dd.fe_ord[i]; nx.ord_e[i] = de.fe_ord[i]; }
  nx.n_e = de.n; nx.soa_w_e = de.soa_w; nx.npub_e = de.npub; nx.soa_col_e = de.soa_col; nx.soa_val_e = de.soa_val;
  nx.pub_index_e = de.pub_index; nx.Ye = de.buf[B_CARRY_Y]; nx.We = de.buf[B_CARRY_W]; nx.Xe = de.buf[B_CARRY_X]; nx.Ge = de.buf[B_CARRY_G];
#define FD_LAUNCH(RR, MM)                                                                                                   \
  hipLaunchKernelGGL((k_step_fd<RR, MM>), dim3(grid), dim3(512), 0, c.stream, c.team, sel, next_sel, step, num_robots,        \
                     restart_interval, nest_src, nest_dst, parity, d, next3_sel, flags, nx, fb, pacc_in, pacc_out, nblk_all, \
                     c.num_agents)
#define FD_LAUNCH_M(RR)                                   \
  switch (m0) {                                           \
    case 24: FD_LAUNCH(RR, 24); break;                    \
    default: break;                                       \
  }
  switch (c.r) {
    case 3: FD_LAUNCH_M(3); break;
    case 4: FD_LAUNCH_M(4); break;
    case 5: FD_LAUNCH_M(5); break;
    default: break;
  }
#undef FD_LAUNCH_M
#undef FD_LAUNCH
}

}  // namespace dpgo
