// step_fused.hip -- ONE launch per pipelined accelerated-RGD iteration: the step
// kernel of precond.hip with the evaluation of spmm.hip's k_eval_stats folded in, without any hand-off between
// workgroups.  Every workgroup (512 threads, one per CU) holds the WHOLE Riemannian gradient of the selected agent in its
// LDS -- the vector its slab of the dense inverse is about to meet -- and there are two ways it gets there.
//
// k_step_fe<R, WD>, WD = 5 .. 8 (round 3; the first two one-launch iterations of a run, and teams that cannot carry rows):
//   1. X of the agent (80 KB at 500 poses) is staged in LDS; the operands of the shared edges (neighbour poses,
//      coefficients) follow it into LDS, one double per lane and trip.
//   2. one lane per pose: W_j = sum_i X_i Q_ij over the row -- the 4 x 4 blocks from a [tile of 64 poses][slot][16-byte
//      chunk][lane] copy of the ELL part (every load of a wave is one contiguous KB; padded slots hold zero blocks),
//      through a ring of four slots in registers; X_i gathered from LDS.  Behind its row the lane of a pose with shared
//      edges forms the linear term G_j from LDS.  Tangent projection in registers; barrier; the result replaces X in LDS.
//   3. the 128 KB slab of M = (Q + shift I)^-1 is requested in four parts, each as soon as a slot of the ring is free for
//      good, then: slab x vector, partial sums, and exactly the tail of k_precond<PM_RGD>: wave 0 finishes the step of
//      the two poses the workgroup owns (step, QF retraction, Nesterov V, look-ahead Y), wave 1 takes the look-ahead step
//      of its share of the other agents' poses.
//
// k_step_fe<R, 0>, "carried rows" (round 5; every other one-launch iteration of a run over >= 3 agents):
//   The row products W_j = sum_i X_i Q_ij of an agent depend on that agent's own poses only, and between two of its
//   turns those move by the look-ahead Nesterov step -- a per-pose map of X, V and scalars that follow a fixed recurrence.
//   So the point agent c will be evaluated at in iteration k is known two launches earlier, and its row products one
//   launch earlier:
//     launch k - 2  (look-ahead wave, FE_CARRY_Y): for the poses of c = sel(k) it applies the look-ahead map a second time
//                   -- the same expressions the look-ahead wave of launch k - 1 will run, on the same operands -- and leaves
//                   the point in B_CARRY_Y.  (c rests in k - 2 and k - 1: three different agents in a row.)
//     launch k - 1  (waves 4-7 behind the partial-sum barrier, FE_CARRY_W): every workgroup forms the row products of ITS
//                   SHARE of c's poses (two or three poses, one (pose, entry) per lane, fe_block's expression slot after
//                   slot) from the complete B_CARRY_Y, and their tangent projection at the point: B_CARRY_G, the gradient
//                   of a pose without shared edges as launch k would form it, bit for bit.  For the public poses it
//                   leaves W_j and the point itself as [entry][public pose] arrays.
//     launch k      (FE_CARRY_IN): waves 0-3 request the head of the slab and B_CARRY_G at once, the bulk of the slab behind
//                   barrier A, and do nothing else until the product; waves 4-7 finish the public poses, one per lane --
//                   W_j + G_j, projection -- from operands whose addresses need no descriptor round trip (16-bit codes in
//                   the launch's descriptor) and overwrite their rows of the vector.  ONE round trip in front of the stream
//                   instead of three; the sparse operator is read once per launch instead of once per workgroup.
//   Same arithmetic on the same operands in the same order: the iterates are still BITWISE those of the two-launch sequence
//   (tests/test_gpu_fused_step.py, profiles/experiments/fe_fuzz.py).  19.8 -> 14.1 us per launch on sphere2500 / 5
//   (profiles/r05_carried_rows.md has the phase traces and what each step bought).
//
// What was learned building it (profiles/r03_fused_step.md, profiles/r05_carried_rows.md):
//   * a CU serves its vector-memory requests in order.  A slab requested first holds every later load of the same CU
//     back until it has landed (X staged at 6.5 us instead of 2): the evaluation cannot hide under the stream, its loads
//     must be IN FRONT of the slab's in the queue.  A pointer that was loaded from memory makes a flat load (waited for
//     with vmcnt(0)), and a value the compiler may resolve early (src ? src : slab) is resolved -- and waited for -- early:
//     every one of these turns "requested now, used later" into a wait for the whole queue (asm volatile pins below).
//   * loads under a wave-uniform `if` cost twice (round 5, read off the ISA): a select on the loaded value inside the
//     branch is waited for inside the branch, and behind the join the compiler's wait counts are those of the path that
//     skipped the loads -- every later wait for an OLDER load then waits for these as well.  Rounds 3-4 requested the slab
//     under `tid < 256`: the stream ran as four exposed round trips.  Every wave issues the same loads now (<R, WD>), or
//     takes one of two whole code paths (<R, 0>).
//   * a wave stays at the issue of its loads for as long as the CU's memory pipe is full: a wave that requests 32 KB
//     executes its next instruction when most of that has landed.  Whoever requests the slab takes part in nothing else
//     until the product; whoever must get a request in front of the slab's bulk does so before barrier A.
//   * __syncthreads() waits for every global load in flight (one counter for loads and stores on gfx9): the barriers
//     here order LDS only (lds_barrier).
//   * one lane per pose reading 128-byte blocks touches 64 cache lines per load instruction: the blocks are stored
//     once more in the order the lanes read them (4x on the evaluation).
//   * the scheduler hoists every gather above the first multiply and spills: the accumulators are pinned per block.
//   * the workgroup that keeps the books must not keep them in front of a barrier the stream waves wait at: workgroup 0
//     finished 1.6 us behind the other 249, and a launch lasts as long as its last workgroup.
//
// What no longer holds "for free" behind a kernel boundary, and how it is kept:
//   * the evaluation reads X of the whole agent and the neighbours' auxiliary poses, the tails / look-ahead steps of the
//     SAME launch write poses.  Round 4: the poses live twice (B_X / B_Y and their twins B_XALT / B_YALT, dpgo_dev.h); a
//     launch reads the copy of its parity and writes the other one -- every pose of every agent moves or is carried over,
//     so the copy a launch leaves is complete -- and the next launch does the opposite; a run of one-launch iterations
//     has an even length, so the state ends in the primary arrays.  (Round 3 kept one copy: every workgroup counted itself
//     in once its gradient was formed and nobody stored before the counter was complete -- which needed every workgroup
//     of a launch resident at once, the device's lock, and a spin that could only give up.)  The carried arrays are
//     written for one agent and read for another in any one launch, and live inside one captured run.
//   * the Nesterov scalars advance between iterations: they are double-buffered -- workgroup 0 writes the next state
//     next to the one every workgroup of this launch reads; the launch that leaves the fused run copies it back
//     (k_eval_stats, nest_copy).
// Mid-run iterations only (ahead == 3: nothing a status query reads is left behind); the last iterations of a run take
// the two-launch sequence, which leaves the statistics.  Dense agents of 32 .. 512 poses where every iteration finds carried
// rows, 449 .. 512 otherwise (the bound where each form beats two launches; DPGO_FE_MIN_N sets it by hand), r <= 5, rows
// of <= 8 blocks, <= 160 shared edges; DPGO_FUSED_EVAL=0 keeps the two-launch sequence everywhere, DPGO_FE_CARRY=0 the
// round-3 form of the one-launch iteration.
#include "kernel_common.h"
#include <algorithm>

namespace dpgo {

// -DDPGO_FE_TRACE: per-wave wall-clock stamps kept in LDS (no registers: the kernel sits at the VGPR limit) and written to
// the agent's partial-sum scratch (PART_E, words [4000 ..]: workgroup DPGO_FE_TRACE_BLOCK, [wave][16]; [4100 + 2 * block]:
// start / end of every workgroup's first wave) when a wave leaves -- profiles/experiments/fe_trace.py
#ifdef DPGO_FE_TRACE
#ifndef DPGO_FE_TRACE_BLOCK
#define DPGO_FE_TRACE_BLOCK 100
#endif
#define FE_TRACE_DECL __shared__ unsigned long long fe_stamps[8 * 16];
#ifndef DPGO_FE_STAMP_MASK
#define DPGO_FE_STAMP_MASK 0xFFFF
#endif
#define FE_STAMP(k) do { if (((DPGO_FE_STAMP_MASK >> (k)) & 1) && (threadIdx.x & 63) == 0) fe_stamps[(threadIdx.x >> 6) * 16 + (k)] = wall_clock64(); } while (0)
#define FE_FLUSH() do { if ((threadIdx.x & 63) == 0) { const int w_ = threadIdx.x >> 6; \
    if (blockIdx.x == DPGO_FE_TRACE_BLOCK) for (int k_ = 0; k_ < 16; ++k_) ag.part[PART_E + 4000 * PART_STRIDE + w_ * 16 + k_] = (double)fe_stamps[w_ * 16 + k_]; \
    if (w_ == 0) { ag.part[PART_E + (4100 + 2 * (int)blockIdx.x) * PART_STRIDE] = (double)fe_stamps[0]; ag.part[PART_E + (4100 + 2 * (int)blockIdx.x) * PART_STRIDE + 1] = (double)fe_stamps[15]; } } } while (0)
#else
#define FE_TRACE_DECL
#define FE_STAMP(k) do { } while (0)
#define FE_FLUSH() do { } while (0)
#endif

constexpr int FE_KC = 2048;
#ifndef DPGO_FE_HEAD
#define DPGO_FE_HEAD 8
#endif
constexpr int FE_HEAD = DPGO_FE_HEAD;  // carried rows: 16-byte loads per lane of the slab requested before the gradient waves' second trip
#ifndef DPGO_FE_PARTS
#define DPGO_FE_PARTS 4
#endif
constexpr int FE_PARTS = DPGO_FE_PARTS;  // the slab is requested in this many parts, behind the last blocks of the row

// one block of the row: W += X_i Q_ij, X_i gathered from the staged copy of X
template <int R>
__device__ __forceinline__ void fe_block(const double *Xs, int i, const double2 *B, double *W) {
  const double *xp = Xs + (size_t)4 * R * i;
  double x[4 * R];
#pragma unroll
  for (int t = 0; t < 2 * R; ++t) { const double2 v = *reinterpret_cast<const double2 *>(xp + 2 * t); x[2 * t] = v.x; x[2 * t + 1] = v.y; }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int a = 0; a < R; ++a)
      W[c * R + a] = fma4(x[a], B[2 * c].x, x[R + a], B[2 * c].y, x[2 * R + a], B[2 * c + 1].x, x[3 * R + a], B[2 * c + 1].y, W[c * R + a]);
}

// the accumulators of a row as operands of an empty volatile asm: everything that feeds them is computed in front of it,
// nothing behind it moves up -- the compiler otherwise gathers every X_i of the row before the first multiply and spills
template <int N>
__device__ __forceinline__ void fe_pin(double *w) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(w[i]));
  asm volatile("" ::: "memory");
}

// WD = ag.soa_w (5 .. 8): the launch forms the row products itself.  WD = 0: "carried rows" -- W_j = sum_i X_i Q_ij of this
// agent was left in B_CARRY_W by the PREVIOUS launch (flags & FE_CARRY_W there), see the head of the file.
template <int R, int WD>
__global__ __launch_bounds__(512) void k_step_fe(const AgentDev *__restrict__ agents, TeamDev *team, int sel, int next_sel, double step,
                                                 int num_robots, int restart_interval, const NestState *nest_src, NestState *nest_dst,
                                                 int parity, const AgentDev agv, int next2_sel, int flags, const AgentDev agn, const FeBases fb) {
  const AgentDev &ag = agv;
  constexpr bool CARRIED = WD == 0;
  // The poses live twice (B_X / B_Y and their twins B_XALT / B_YALT): this launch reads the copy of its parity and writes
  // the other one -- every pose of every agent, so the copy it leaves is complete -- and the next launch does the
  // opposite.  Nobody overwrites what another workgroup of the same launch still reads: no arrival counter, no wait, no
  // co-residency condition, no device lock (rounds 3's form counted the workgroups in and spun).
  const double *__restrict__ Xr = ag.buf[parity ? B_XALT : B_X];
  const double *__restrict__ Yr = ag.buf[parity ? B_YALT : B_Y];
  double *__restrict__ Xw = ag.buf[parity ? B_X : B_XALT];
  double *__restrict__ Yw = ag.buf[parity ? B_Y : B_YALT];
  // XCD-aware block order, as in k_precond
  const int hb = (int)blockIdx.x;
  const int bx = (hb % 8) * ((int)gridDim.x / 8) + hb / 8;
  const int tid = threadIdx.x;
  const int N4 = ag.N4, n = ag.n;
  const int nblk = (N4 + 7) / 8;
  if (bx >= nblk) return;
  constexpr int KC = FE_KC, MREG = KC / 64;
  __shared__ double vs[R * KC];          // X of the agent, then its Riemannian gradient: [k][a], the arrays' own layout
  __shared__ double zs[8 * R];
  __shared__ double red[32 * (8 * R + 1)];
  __shared__ double Ysh[2 * 4 * R];
  __shared__ double Esh[2][2 * 4 * R];   // V, Yaux of the two poses
  __shared__ double Es[FE_MAX_EDGES * (4 * R + 16)];  // operands of the shared edges: neighbour pose, coefficients
  __shared__ double tl_x[2 * 4 * R], tl_v[2 * 4 * R], tl_y[2 * 4 * R], tl_s[2 * 16];  // the two poses of the tail, lane-parallel
  FE_TRACE_DECL
  FE_STAMP(0);
  const int pj0 = 2 * bx, pj1 = (2 * bx + 1 < n) ? 2 * bx + 1 : -1;
  const int npose = (pj1 >= 0) ? 2 : 1;

  // operands of the tail (consumed 10 us from here)
  const size_t own_off = (size_t)((tid >= 4 * R) ? max(pj1, 0) : pj0) * 4 * R + (size_t)(tid % (4 * R));
  double pre_x = 0, pre_v = 0, pre_y = 0;
  if (tid < npose * 4 * R) {
    pre_x = Xr[own_off];
    pre_v = ag.buf[B_V][own_off];
    pre_y = Yr[own_off];
  }
  NestState ns = {};
  if (tid < 128) ns = nest_src[sel];
  constexpr int EPE = 4 * R + 16;                              // doubles per shared edge in LDS: neighbour pose, coefficients
  const int cg = (tid >> 5) & 7, kl = tid & 31;
  const int col = 8 * bx + cg;
  const bool cact = col < N4 && tid < 256;
  const double *Mc = ag.M + (size_t)(cact ? col : 0) * N4;
  double2 mreg[MREG];
  if constexpr (CARRIED) {
    // ================================================================ carried rows (WD = 0): two kinds of waves from the
    // first instruction on, and ONE round trip in front of the stream.
    //   * waves 0-3 (stream): request the head of their 128 KB slab of M and the carried gradient at once, the rest of the
    //     slab behind barrier A.  A wave stays at the issue of its loads for as long as the CU's memory pipe is full, i.e.
    //     until most of what it asked for has landed (traced in round 5: with the slab requested behind barrier 1 by all
    //     waves, the next instruction of every wave ran 5 us later) -- so these waves take part in nothing else until the
    //     product.  The carried gradient -- the previous launch left, for every pose of this agent, the tangent projection of
    //     its row product at the evaluation point, which IS the Riemannian gradient of a pose without shared edges, bit for
    //     bit (B_CARRY_G, [pose][4r]: the layout of the vector in LDS) -- goes into LDS when they are through.
    //   * waves 4-7 (gradient) finish the public poses, one per lane (wave g: public poses 64 g ..): W_j and X_j from the
    //     [entry][public pose] arrays the previous launch left, the coefficients of the lane's shared edges (one or two edges
    //     per lane: the edges of the wave's 64 public poses, fe_eptr) and the neighbour poses they meet -- whose addresses
    //     come out of the 16-bit codes in the descriptor (scalar loads and a select chain: no descriptor round trip).
    //     Everything is requested in front of barrier A; G_j, the projection, and behind barrier B the row of the vector in
    //     LDS is overwritten.
    // Barrier A: the gradient waves have requested all they need (the bulk of the slab queues behind it, not in front);
    // B: the carried gradient is in LDS; B2: the rows of the public poses are.  (The two kinds of waves are whole waves --
    // the branch below is on the wave index -- and each meets exactly A, B and B2, from its own code path: the two paths
    // cannot share the barriers' program points without keeping both register sets alive at once.)
    const int cwv = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (cwv < 4) {
      for (int t = N4 * R + tid; t < KC * R; t += 256) vs[t] = 0.0;  // rows of the vector beyond the agent's
#pragma unroll
      for (int m = 0; m < FE_HEAD; ++m) mreg[m] = ld2_nt(Mc + min(2 * kl + 64 * (int)ag.fe_ord[m], N4 - 2));
      constexpr int NG = (KC * R / 2 + 255) / 256;
      double2 gv[NG];
      {
        const double *Gc = ag.buf[B_CARRY_G];
#pragma unroll
        for (int u = 0; u < NG; ++u) gv[u] = ld2(Gc + min(2 * (tid + 256 * u), N4 * R - 2));
      }
      FE_STAMP(10);
      lds_barrier();  // A
      FE_STAMP(11);
#pragma unroll
      for (int m = FE_HEAD; m < MREG; ++m) mreg[m] = ld2_nt(Mc + min(2 * kl + 64 * (int)ag.fe_ord[m], N4 - 2));
      FE_STAMP(12);
#pragma unroll
      for (int u = 0; u < NG; ++u) {
        const int tt = 2 * (tid + 256 * u);
        if (tt < N4 * R) *reinterpret_cast<double2 *>(&vs[tt]) = gv[u];
      }
      lds_barrier();  // B
    } else {
      const int g = cwv - 4, ln = tid & 63;
      const int npub = ag.npub;
      const int t0 = ag.fe_eptr[g], t1 = ag.fe_eptr[g + 1], ne = t1 - t0;  // shared edges of this wave's public poses (uniform; <= 128, host)
      // the lane's edges (ln, ln + 64 of the wave's): the neighbour pose from its code, the 16 coefficients
      double2 cf[2][8], xe[2][2 * R];
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (64 * q < ne) {  // (uniform)
          const int ei = t0 + min(ln + 64 * q, ne - 1);
          // word ei >> 1 of the code table: the wave's words sit in scalar registers, the lane picks its own
          const int wbase = (t0 + 64 * q) >> 1, wrel = (ei >> 1) - wbase;  // 0 .. 32
          unsigned wsel = 0;
#pragma unroll
          for (int k = 0; k <= 32; ++k) {
            const unsigned wk = ag.fe_code[min(wbase + k, FE_MAX_EDGES / 2 - 1)];
            wsel = (wrel == k) ? wk : wsel;
          }
          const unsigned code = (ei & 1) ? (wsel >> 16) : (wsel & 0xffffu);
          const int sa = (int)(code >> 12), sf = (int)(code & 0xfffu);
          const double *yb = fb.ybase[0];
          int yn = fb.npose[0];
#pragma unroll
          for (int k = 1; k < LOOKAHEAD_MAX_AGENTS; ++k) { yb = (sa == k) ? fb.ybase[k] : yb; yn = (sa == k) ? fb.npose[k] : yn; }
          const double *xp = yb + (parity ? (size_t)B_ALT * 4 * R * yn : (size_t)0) + (size_t)sf * 4 * R;
#pragma unroll
          for (int k = 0; k < 2 * R; ++k) {
            const v2d_t t = *(const __attribute__((address_space(1))) v2d_t *)(xp + 2 * k);
            xe[q][k] = make_double2(t.x, t.y);
          }
          const double *cp_ = ag.se[ei].coef;
#pragma unroll
          for (int k = 0; k < 8; ++k) cf[q][k] = ld2(cp_ + 2 * k);
        }
      // the lane's public pose: its row products and the point, its edges
      const int pq = 64 * g + ln;
      const bool pact = pq < npub;
      const int pqc = pact ? pq : 0;
      const int pj = ag.pub_pose[pqc];
      const int pe0 = ag.pub_ptr[pqc], pe1 = ag.pub_ptr[pqc + (pact ? 1 : 0)];
      double w[4 * R], x[4 * R];
      {
        const double *Wc = ag.buf[B_CARRY_W], *Xc = ag.buf[B_CARRY_X];
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) { w[i] = Wc[(size_t)i * npub + pqc]; x[i] = Xc[(size_t)i * npub + pqc]; }
      }
      FE_STAMP(10);
      lds_barrier();  // A
      FE_STAMP(11);
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (64 * q < ne && ln + 64 * q < ne) {
          double *E = Es + (size_t)(t0 + ln + 64 * q) * EPE;
#pragma unroll
          for (int k = 0; k < 2 * R; ++k) *reinterpret_cast<double2 *>(E + 2 * k) = xe[q][k];
#pragma unroll
          for (int k = 0; k < 8; ++k) *reinterpret_cast<double2 *>(E + 4 * R + 2 * k) = cf[q][k];
        }
      WSYNC();
      FE_STAMP(1);
      if (pact) {
        // G_j from LDS: g[c][a] -= x[cp][a] coef[cp + 4c], edge after edge and cp after cp for every entry (g_row_range's
        // order); the operands of an edge are read once
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
        FE_STAMP(2);
        tangent_inplace<R>(x, w);
        FE_STAMP(3);
      }
      lds_barrier();  // B
      if (pact) {
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) vs[(size_t)4 * R * pj + i] = w[i];
      }
    }
    lds_barrier();  // B2: the gradient is in LDS
  } else {
  const int kslab = (tid < 256) ? 2 * kl : 0, kstep = (tid < 256) ? 64 : 0;  // (waves 4-7: always row 0)
  // ================================================================ the evaluation: one lane per pose, all 8 waves
  // W_j = sum_i X_i Q_ij + G_j (see the head of the file)
  const int j = tid;
  const bool act = j < n;
  const int jj = act ? j : 0;
  // The linear term G (the neighbours' poses through the shared edges): the operands of every shared edge -- the
  // neighbour's auxiliary pose, 4R doubles, and the edge's 16 coefficients -- are copied into LDS by all lanes, one double
  // per lane and trip (consecutive lanes read consecutive doubles of an edge); behind its row the lane of a pose with
  // shared edges forms G_j from LDS, edge after edge in g_row_range's order (the host checks nshared <= FE_MAX_EDGES).
  // The neighbour's pose comes straight from its agent's Y array where it is co-resident (or imported), from the
  // neighbour slab otherwise (g_row_range, aux = 1, pull).
  constexpr int NEI = (FE_MAX_EDGES * EPE + 511) / 512;        // trips that cover FE_MAX_EDGES edges
  const int nsh = ag.nshared, etotal = nsh * EPE;
  const double *esrc[NEI];
  int eslot[NEI];
#pragma unroll
  for (int i = 0; i < NEI; ++i) {
    esrc[i] = nullptr; eslot[i] = 0;
    if (i * 512 < etotal) {  // (uniform; lanes beyond the last double re-read the last edge and drop it)
      const SharedEdgeDev &se = ag.se[min((tid + 512 * i) / EPE, nsh - 1)];
      esrc[i] = parity ? se.src_yalt : se.src[1]; eslot[i] = se.slot;
    }
  }
  const int e0 = ag.pose_eptr[jj], e1 = ag.pose_eptr[jj + 1];  // the pose's shared edges (empty for most poses)
  constexpr int NSTG = (KC * R / 2 + 511) / 512;
  double2 xv[NSTG];
  {
    const double *X = Xr;
#pragma unroll
    for (int u = 0; u < NSTG; ++u) {
      const int tt = 2 * (tid + 512 * u);  // N4 * R is even
      const double2 t = ld2(X + min(tt, N4 * R - 2));
      xv[u] = (tt < N4 * R) ? t : make_double2(0.0, 0.0);
    }
  }
  // (every wave waits for ITS OWN share of X before it requests the ring: a CU serves its requests in order)
#pragma unroll
  for (int u = 0; u < NSTG; ++u) {
    const int tt = 2 * (tid + 512 * u);
    if (tt < KC * R) *reinterpret_cast<double2 *>(&vs[tt]) = xv[u];
  }
  // second trip of the edge operands (their descriptors were requested first and are back), in front of the ring
  double ev[NEI];
#pragma unroll
  for (int i = 0; i < NEI; ++i) {
    ev[i] = 0.0;
    if (i * 512 < etotal) {
      const int t = tid + 512 * i, e = min(t / EPE, nsh - 1), k = t - (t / EPE) * EPE;
      // (the descriptor is LOOKED AT here and not earlier: left alone, the compiler resolves each pointer right behind
      // its loads -- one exposed round trip per descriptor)
      asm volatile("" : "+v"(esrc[i]), "+v"(eslot[i]));
      const double *xp = esrc[i] ? esrc[i] : ag.nbr[1] + (size_t)eslot[i] * 4 * R;
      const double *src = (k < 4 * R) ? xp + k : ag.se[e].coef + (k - 4 * R);
      // (a pointer that was loaded from memory is a generic pointer to the compiler; a flat load may return out of
      // order, so its data would be waited for with vmcnt(0): behind the whole ring.  These are global addresses.)
      ev[i] = *(const __attribute__((address_space(1))) double *)src;
    }
  }
  lds_barrier();  // #1: X is in LDS.  (LDS-only barrier: the edge operands stay in flight)
  FE_STAMP(1);
  // the blocks of the row travel through a ring of FE_RING slots: the first FE_RING are requested here -- BEHIND the
  // barrier: a wave is held at the issue of its loads for as long as the CU's address unit is busy with everybody's
  // (36 requests of 1 KB per wave, 8 waves), and nothing in front of the barrier may wait for that --, slot u + FE_RING
  // as soon as slot u has been used
  constexpr int RING = 4;
  constexpr int Wd = WD;  // = ag.soa_w, 5 .. 8 (the launch picks the instance)
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;
  const double *qt = ag.soa_val + (size_t)wv * Wd * 1024 + 2 * ln;  // this wave's tile of blocks, this lane's 16 bytes
  const int *ct = ag.soa_col + (size_t)wv * Wd * 64 + ln;
  int idx[RING];
  double2 B[RING][8];
#pragma unroll
  for (int k = 0; k < RING; ++k) {
    idx[k] = ct[k * 64];
#pragma unroll
    for (int q = 0; q < 8; ++q) B[k][q] = ld2(qt + (k * 8 + q) * 128);
  }
  // the slab of M (waves 0-3): requested in four parts, each as soon as a slot of the ring is free for good -- behind
  // every load of the evaluation in the CU's queue (a slab requested earlier holds them back until it has landed:
  // measured, X staged at 6.5 us instead of 2), and as early as the registers allow: it streams under the rest of the row
  // the edge operands were requested in front of the ring and are back before its first slot: into LDS now (their
  // registers are free for the row)
#pragma unroll
  for (int i = 0; i < NEI; ++i)
    if (i * 512 < etotal) { const int t = tid + 512 * i; if (t < etotal) Es[t] = ev[i]; }
  double w[4 * R];
#pragma unroll
  for (int i = 0; i < 4 * R; ++i) w[i] = 0.0;
#pragma unroll
  for (int u = 0; u < WD; ++u) {
    const int k = u % RING;
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise hoists every gather and refill above the first block and spills)
    fe_block<R>(vs, idx[k], B[k], w);
    fe_pin<4 * R>(w);
    if (u + RING < WD) {
      idx[k] = ct[(u + RING) * 64];
#pragma unroll
      for (int q = 0; q < 8; ++q) B[k][q] = ld2(qt + ((u + RING) * 8 + q) * 128);
    } else if (u >= WD - FE_PARTS) {
      // EVERY wave issues these loads, and nothing is selected on their data here.  Under `tid < 256` the compiler's wait
      // counts behind the join are those of the path that skipped the loads: every block of the row behind a slab part
      // then waited for that part to LAND, and a select on the loaded value inside the branch waited for it on the spot --
      // the stream ran as four exposed round trips (round 5, read off the ISA).  Waves 4-7 read one 16-byte word of M over
      // and over (their copies are never used); rows k >= N4 meet zeros of the vector, columns beyond the last pose feed
      // sums nobody reads.
      constexpr int PART = MREG / FE_PARTS;
      const int part = u - (WD - FE_PARTS);  // 0 .. FE_PARTS - 1
#pragma unroll
      for (int m = part * PART; m < (part + 1) * PART; ++m) mreg[m] = ld2_nt(Mc + min(kslab + kstep * (int)ag.fe_ord[m], N4 - 2));
    }
    if (u == 1) lds_barrier();  // #1a: the edge operands are in LDS
    __builtin_amdgcn_sched_barrier(0);
  }
  if (act && e1 > e0) {
    // G_j from LDS: g[c][a] -= x[cp][a] coef[cp + 4c], edge after edge and cp after cp for every entry (g_row_range's
    // order), one column c at a time (the slab's registers are in flight: few are free)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double g[R];
#pragma unroll
      for (int a = 0; a < R; ++a) g[a] = 0.0;
      for (int e = e0; e < e1; ++e) {
        const double *E = Es + (size_t)e * EPE;
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
          const double cf = E[4 * R + cp + 4 * c];
#pragma unroll
          for (int a = 0; a < R; ++a) g[a] -= E[cp * R + a] * cf;
        }
      }
#pragma unroll
      for (int a = 0; a < R; ++a) w[c * R + a] = w[c * R + a] + g[a];
    }
  }
  FE_STAMP(2);
  {
    double y[4 * R];
#pragma unroll
    for (int i = 0; i < 4 * R; ++i) y[i] = vs[(size_t)4 * R * (act ? j : 0) + i];
    tangent_inplace<R>(y, w);
  }
  FE_STAMP(3);
  lds_barrier();  // #2: nobody reads X in LDS any more
  if (act) {
#pragma unroll
    for (int i = 0; i < 4 * R; ++i) vs[(size_t)4 * R * j + i] = w[i];
  }
  lds_barrier();  // #3: the gradient is in LDS
  }
  FE_STAMP(4);
  if (tid >= 256) {
    // (waves 4-7 meet the stream waves' barrier first: neither the bookkeeping nor the carried rows below may hold the
    // product's partial sums back -- workgroup 0, which keeps the books, used to finish 1.6 us behind all the others)
    lds_barrier();  // #4: (the stream waves' partial sums)
    if (bx == 0 && tid < 256 + LOOKAHEAD_MAX_AGENTS) {
      // what the bookkeeping workgroup of the next k_eval_stats would do (advance_agent, accelerated): into the OTHER
      // state buffer -- every workgroup of this launch reads nest_src.  One lane per agent (one round trip for all of them)
      const int na = team->num_agents, k = tid - 256;
      const double Nr = (double)num_robots;
      if (k < na) {
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
    if (flags & FE_CARRY_W) {
      // ---- the row products of the NEXT agent (carried rows): W_p = sum_i X_i Q_ip at the point the next launch will
      // evaluate at -- B_CARRY_Y, left complete by the launch before this one -- for this workgroup's share of its poses,
      // one (pose, entry) per lane, three poses per wave: fe_block's expression slot after slot, so the sums are BITWISE the
      // ones the next launch would form itself.  Then, one lane per pose, their tangent projection at the point: the
      // gradient of a pose without shared edges as the next launch would form it (it adds G_j to W_j only where there are
      // edges).  Two short round trips behind the slab in this CU's queue, next to the stream waves' product.
      const int l = tid - 256, lw = l & 63, wq = l >> 6;
      const int npw = (agn.n + nblk - 1) / nblk;  // (<= 12, checked by the host)
      const int ls = lw / (4 * R), e = lw - ls * (4 * R);
      const int lp = 3 * wq + ls;
      const int pw = bx * npw + lp;
      const bool pv = ls < 3 && lp < npw && pw < agn.n;
      double *Ex = Es + (size_t)wq * 2 * 3 * 4 * R;  // (the edge operands are not needed any more: this wave's exchange space)
      if (pv) {
        const int c = e / R, a = e - c * R;
        const int tile = pw >> 6, pl = pw & 63, wdn = agn.soa_w;
        const double *__restrict__ Y2 = agn.buf[B_CARRY_Y];
        // (all indices, then all operands, then the sums: two round trips for the row instead of two per slot -- this wave
        // must not be the last one of its workgroup to leave)
        int ii[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ii[u] = gp(agn.soa_col)[((size_t)tile * wdn + min(u, wdn - 1)) * 64 + pl];
        double xv[8][4], bv[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double *xp = Y2 + (size_t)4 * R * ii[u] + a;
          const double *bp = agn.soa_val + ((size_t)tile * wdn + min(u, wdn - 1)) * 1024 + (2 * c) * 128 + 2 * pl;
          xv[u][0] = gp(xp)[0]; xv[u][1] = gp(xp)[R]; xv[u][2] = gp(xp)[2 * R]; xv[u][3] = gp(xp)[3 * R];
          bv[u][0] = gp(bp)[0]; bv[u][1] = gp(bp)[1]; bv[u][2] = gp(bp)[128]; bv[u][3] = gp(bp)[129];
        }
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double t = fma4(xv[u][0], bv[u][0], xv[u][1], bv[u][1], xv[u][2], bv[u][2], xv[u][3], bv[u][3], acc);
          acc = (u < wdn) ? t : acc;
        }
        const double xe_ = Y2[(size_t)4 * R * pw + e];
        const int qi = agn.pub_index[pw];
        if (qi >= 0) {  // (a public pose: the next launch finishes it -- row product and point, [entry][public pose])
          agn.buf[B_CARRY_W][(size_t)e * agn.npub + qi] = acc;
          agn.buf[B_CARRY_X][(size_t)e * agn.npub + qi] = xe_;
        }
        Ex[ls * 4 * R + e] = acc;
        Ex[3 * 4 * R + ls * 4 * R + e] = xe_;
      }
      WSYNC();
      if (pv && e == 0) {
        double w[4 * R], x[4 * R];
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) { w[i] = Ex[ls * 4 * R + i]; x[i] = Ex[3 * 4 * R + ls * 4 * R + i]; }
        tangent_inplace<R>(x, w);
        double *Gn = agn.buf[B_CARRY_G] + (size_t)4 * R * pw;
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) Gn[i] = w[i];
      }
    }
    FE_STAMP(14);
    FE_FLUSH();
    return;
  }

  const double Nr = (double)num_robots;
  const bool restart_now = ((ns.iter + 2) % restart_interval) == 0;
  const bool restart_next = ((ns.iter + 3) % restart_interval) == 0;
  const double nest_gamma = restart_now ? 0.0 : (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * ns.gamma * ns.gamma)) / (2.0 * Nr);
  const double g2 = (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * nest_gamma * nest_gamma)) / (2.0 * Nr);
  const double ahead_alpha = 1.0 / (g2 * Nr);
  // (two iterations ahead, carried rows: the scalars the NEXT launch will look ahead with)
  const bool restart_next2 = ((ns.iter + 4) % restart_interval) == 0;
  const double gamma_next = restart_next ? 0.0 : g2;
  const double g3 = (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * gamma_next * gamma_next)) / (2.0 * Nr);
  const double ahead2_alpha = 1.0 / (g3 * Nr);
  const bool ahead_opt = next_sel == sel;
  double acc[R];
#pragma unroll
  for (int a = 0; a < R; ++a) acc[a] = 0;
#pragma unroll
  for (int m = 0; m < MREG; ++m) {
    const int k = 2 * kl + 64 * (int)ag.fe_ord[m];  // (the agent's chunk order, as k_precond)
    double wv[2 * R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const double2 t2 = *reinterpret_cast<const double2 *>(&vs[k * R + 2 * q]);
      wv[2 * q] = t2.x; wv[2 * q + 1] = t2.y;
    }
#pragma unroll
    for (int a = 0; a < R; ++a) acc[a] = __builtin_fma(wv[R + a], mreg[m].y, __builtin_fma(wv[a], mreg[m].x, acc[a]));  // (two FMAs: the one-expression form is mul, fma, add)
  }
  FE_STAMP(5);
#pragma unroll
  for (int a = 0; a < R; ++a) red[kl * (8 * R + 1) + cg * R + a] = acc[a];
  if (tid < npose * 4 * R) { Ysh[tid] = pre_x; Esh[0][tid] = pre_v; Esh[1][tid] = pre_y; }
  lds_barrier();  // #4
  if (tid >= 128) { FE_FLUSH(); return; }
  if (tid < 8 * R) {
    double s = 0;
#pragma unroll
    for (int q = 0; q < 32; ++q) s += red[q * (8 * R + 1) + tid];
    zs[tid] = s;
  }
  FE_STAMP(6);
  if (tid < 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (tid >= 64) {
    // ---- look-ahead Nesterov step of iteration k+1 for this workgroup's share of the OTHER agents' poses (second wave;
    // k_precond<PM_RGD>, ahead bit 1, without the status outputs)
    const int self = sel;
    int pre[LOOKAHEAD_MAX_AGENTS + 1];
    double *px[LOOKAHEAD_MAX_AGENTS], *pv[LOOKAHEAD_MAX_AGENTS], *py[LOOKAHEAD_MAX_AGENTS];
    long long dalt[LOOKAHEAD_MAX_AGENTS];  // doubles from a primary array of agent k to its twin
#pragma unroll
    for (int k = 0; k <= LOOKAHEAD_MAX_AGENTS; ++k) pre[k] = team->pose_prefix[k];
    const int na = team->num_agents;
#pragma unroll
    for (int k = 0; k < LOOKAHEAD_MAX_AGENTS; ++k) {
      px[k] = (k < na) ? agents[k].buf[B_X] : nullptr;
      pv[k] = (k < na) ? agents[k].buf[B_V] : nullptr;
      py[k] = (k < na) ? agents[k].buf[B_Y] : nullptr;
      dalt[k] = (long long)B_ALT * 4 * R * (pre[k + 1] - pre[k]);
    }
    double *py2 = (flags & FE_CARRY_Y) ? agents[next2_sel].buf[B_CARRY_Y] : nullptr;
    const int total = pre[LOOKAHEAD_MAX_AGENTS] - n;
    const int per = (total + nblk - 1) / nblk;  // <= 64, checked by the host
    const int l1 = tid - 64;
    const int q = bx * per + l1;
    if (l1 < per && q < total) {
      int self_lo = 0;
#pragma unroll
      for (int k = 0; k < LOOKAHEAD_MAX_AGENTS; ++k) if (k == self) self_lo = pre[k];
      const int gq = q < self_lo ? q : q + n;
      int a = 0, lo = 0;
      double *xa = px[0], *va = pv[0], *ya = py[0];
      long long da = dalt[0];
#pragma unroll
      for (int k = 1; k < LOOKAHEAD_MAX_AGENTS; ++k)
        if (k < na && gq >= pre[k]) { a = k; lo = pre[k]; xa = px[k]; va = pv[k]; ya = py[k]; da = dalt[k]; }
      // this launch's copy of the agent's poses (read) and the other one (written)
      const double *xr = parity ? xa + da : xa, *yr = parity ? ya + da : ya;
      double *oX = parity ? xa : xa + da, *oY = parity ? ya : ya + da, *oV = va;
      const int la_pose = gq - lo;
      const bool la_opt = next_sel == a;
      const size_t o = (size_t)la_pose * 4 * R;
      double la_x[4 * R], la_v[4 * R];
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) { la_x[i] = xr[o + i]; la_v[i] = va[o + i]; }
      if (restart_next) {
        // (X stays; Y = V = X unless the agent optimizes next -- then Y stays too: both are carried into the other copy)
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) {
          oX[o + i] = la_x[i];
          if (!la_opt) { oY[o + i] = la_x[i]; oV[o + i] = la_x[i]; la_v[i] = la_x[i]; } else oY[o + i] = yr[o + i];
        }
      } else {
        double y[4 * R];
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) y[i] = (1.0 - ahead_alpha) * la_x[i] + ahead_alpha * la_v[i];
        polar_inplace<R>(y);
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) { oY[o + i] = y[i]; oX[o + i] = y[i]; la_x[i] = y[i]; }
      }
      if (py2 && a == next2_sel) {
        // carried rows, first half: the point the agent of iteration k+2 will be evaluated at -- what the look-ahead wave of
        // the NEXT launch will leave in its X array (the same expressions on the same operands: bitwise) -- is formed one
        // launch early, so that the next launch finds it complete and can form the row products on the side.  (The agent
        // moves in neither launch: it is neither this launch's nor the next one's.  la_x / la_v hold its X and V after
        // iteration k+1 here.)
        if (!restart_next2) {
          double y[4 * R];
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) y[i] = (1.0 - ahead2_alpha) * la_x[i] + ahead2_alpha * la_v[i];
          polar_inplace<R>(y);
#pragma unroll
          for (int i = 0; i < 4 * R; ++i) la_x[i] = y[i];
        }
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) py2[o + i] = la_x[i];
      }
    }
    FE_STAMP(15);
    FE_FLUSH();
    return;
  }

  // ---- the step of the workgroup's two poses (k_precond<PM_RGD>, advance = 2, ahead bit 0, no statistics): one pose on 16
  // lanes, the pose in LDS (device_math.h, lane-parallel forms: bitwise the serial routines) -- the six Gram sums of a QF /
  // polar factor side by side, the entries of a product on 3R lanes.  Serial, this tail was 2.2 us of dependent fp64
  // arithmetic on two lanes of the wave.  (The look-ahead wave next to it keeps one lane per pose: its single polar factor
  // is off the critical path -- spreading it over lanes as well was measured and changes nothing.)
  {
    const int lp = tid >> 4, s = tid & 15;
    if (lp < npose) {
      const size_t o = (size_t)(lp ? pj1 : pj0) * 4 * R;
      double *xs = tl_x + lp * 4 * R, *vsv = tl_v + lp * 4 * R, *ys = tl_y + lp * 4 * R, *Ss = tl_s + lp * 16;
      double *zz = zs + lp * 4 * R;
      const double *x0 = Ysh + lp * 4 * R, *v0 = Esh[0] + lp * 4 * R, *y0 = Esh[1] + lp * 4 * R;
      const int i0 = s, i1 = s + 16;  // the (at most two) entries of the pose this lane moves
      const bool h0 = i0 < 4 * R, h1 = i1 < 4 * R;
      tangent_lanes<R>(x0, zz, Ss, s);
      if (h0) xs[i0] = x0[i0] - step * zz[i0];
      if (h1) xs[i1] = x0[i1] - step * zz[i1];
      lanes_sync();
      qf_lanes<R>(xs, Ss, s);
      FE_STAMP(7);
      const bool reset = restart_now;  // restart iteration: V = Y = X
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
      FE_STAMP(8);
      if (restart_next) {
        if (h0) {
          Xw[o + i0] = xs[i0];
          if (!ahead_opt) { Yw[o + i0] = xs[i0]; vsv[i0] = xs[i0]; }
          else Yw[o + i0] = reset ? xs[i0] : y0[i0];  // (else Y stays: carried into the other copy)
        }
        if (h1) {
          Xw[o + i1] = xs[i1];
          if (!ahead_opt) { Yw[o + i1] = xs[i1]; vsv[i1] = xs[i1]; }
          else Yw[o + i1] = reset ? xs[i1] : y0[i1];
        }
      } else {
        if (h0) ys[i0] = (1.0 - ahead_alpha) * xs[i0] + ahead_alpha * vsv[i0];
        if (h1) ys[i1] = (1.0 - ahead_alpha) * xs[i1] + ahead_alpha * vsv[i1];
        lanes_sync();
        polar_lanes<R>(ys, Ss, s);
        FE_STAMP(9);
        if (h0) { Yw[o + i0] = ys[i0]; Xw[o + i0] = ys[i0]; }
        if (h1) { Yw[o + i1] = ys[i1]; Xw[o + i1] = ys[i1]; }
      }
      lanes_sync();
      if (h0) ag.buf[B_V][o + i0] = vsv[i0];
      if (h1) ag.buf[B_V][o + i1] = vsv[i1];
    }
  }
  FE_STAMP(15);
  FE_FLUSH();
}

// r <= 5: R * 2048 + 256 * 4R doubles of LDS next to the partial sums (133 KB at r = 5)
bool step_fe_supported(int r) { return r >= 3 && r <= 5; }
int step_fe_max_edges() { return FE_MAX_EDGES; }

int step_fe_carry_max_poses() { return 256 / 20; }  // poses of the next agent a workgroup can take (one (pose, entry) per lane of waves 4-7)

// carry: FE_CARRY_IN -- the row products of `sel` wait in its B_CARRY_W (the previous launch ran with FE_CARRY_W);
// FE_CARRY_W -- form the row products of next_sel from its B_CARRY_Y (left by the launch before this one, FE_CARRY_Y there);
// FE_CARRY_Y -- leave the evaluation point of next2_sel in its B_CARRY_Y
void launch_step_fe(const LaunchCtx &c, int sel, int next_sel, double step, int num_robots, int restart_interval,
                    const NestState *nest_src, NestState *nest_dst, int parity, int next2_sel, int carry) {
  const AgentDev &d = c.host_agents[sel];
  const AgentDev &dn = c.host_agents[(carry & FE_CARRY_W) ? next_sel : sel];
  const int grid = ((d.N4 + 7) / 8 + 7) / 8 * 8;
  const int flags = carry & (FE_CARRY_W | FE_CARRY_Y);
  FeBases fb = {};
  for (int k = 0; k < c.num_agents && k < LOOKAHEAD_MAX_AGENTS; ++k) { fb.ybase[k] = c.host_agents[k].buf[B_Y]; fb.npose[k] = c.host_agents[k].n; fb.part[k] = c.host_agents[k].part; }
#define FE_LAUNCH(RR, WW)                                                                                              \
  hipLaunchKernelGGL((k_step_fe<RR, WW>), dim3(grid), dim3(512), 0, c.stream, c.agents, c.team, sel, next_sel, step, num_robots, \
                     restart_interval, nest_src, nest_dst, parity, d, next2_sel, flags, dn, fb)
#define FE_LAUNCH_W(RR)                                                                                                \
  switch ((carry & FE_CARRY_IN) ? 0 : d.soa_w) {                                                                       \
    case 0: FE_LAUNCH(RR, 0); break;                                                                                   \
    case 5: FE_LAUNCH(RR, 5); break;                                                                                   \
    case 6: FE_LAUNCH(RR, 6); break;                                                                                   \
    case 7: FE_LAUNCH(RR, 7); break;                                                                                   \
    case 8: FE_LAUNCH(RR, 8); break;                                                                                   \
    default: break;                                                                                                    \
  }
  switch (c.r) {
    case 3: FE_LAUNCH_W(3); break;
    case 4: FE_LAUNCH_W(4); break;
    case 5: FE_LAUNCH_W(5); break;
    default: break;
  }
#undef FE_LAUNCH_W
#undef FE_LAUNCH
}

}  // namespace dpgo
