// rtr_fused.hip -- ONE launch per local RTR solve (QuadraticOptimizer::optimize with the
// trust-region Newton method, Steihaug truncated CG preconditioned by the dense inverse).  A persistent kernel, one
// workgroup per CU, that keeps the preconditioner ON CHIP:
//   * workgroup b owns poses 2b, 2b+1 of the agent, i.e. 8 columns of M = (Q + shift I)^-1.  Its 8 x N4 slab of M
//     (128 KB for a 500-pose agent; 32 MB over the grid) is read from HBM ONCE per solve into LDS and serves every
//     preconditioner apply of every tCG iteration of every outer iteration from there.  The launch-per-step sequence
//     streams the 32 MB once per tCG iteration (solve.hip: [init, (Hess-vec, step) x J, retract, evaluate, accept]
//     patterns, each kernel 9-15 us, a quarter of them phase-gated no-ops);
//   * all scalars of the solve (dots, alpha, beta, rho, radius, the phase) live in registers: every wave re-derives
//     them from the same per-workgroup partial sums in the same order, so the control flow is uniform over the grid
//     and the host reads ONE state record at the end;
//   * workgroups exchange the vectors a phase needs from the others (delta, z, H delta, eta, the candidate point and
//     its gradient) through write-through stores (agent-scope relaxed atomics = global_store sc1), a sharded-counter
//     grid hand-off (cdna_hip_programming.md Guideline 16, R1 form; see iter_fused.hip) and L1-bypassing loads.
// The arithmetic is that of the launch-per-step kernels (k_precond<TCG_INIT / TCG_STEP>, k_tcg_hv, k_retract,
// k_rtr_eval2, k_rtr_accept); only the order in which the 2000 products of a preconditioner row are added differs.
// Eligibility (rtr_fused_eligible): dense preconditioner, ceil(n / 2) <= CUs, slab + scratch <= 160 KB of LDS.
// Larger agents, block-Jacobi agents and the colour-parallel group update keep the launch-per-step path.
#include "kernel_common.h"
#include "twolevel_dev.h"

namespace dpgo {

constexpr int RB_LINE = 16;             // 64-bit words per 128-byte line
constexpr int RB_TOP = 8 * RB_LINE;     // shard counters at g * RB_LINE
constexpr int RB_GEN = 9 * RB_LINE;     // generation words at RB_GEN + g * RB_LINE
constexpr int RB_EPOCH = 17 * RB_LINE;  // epoch of the last completed hand-off (carried across launches)
constexpr int RB_ABORT = 17 * RB_LINE + 1;  // raised by a workgroup whose hand-off timed out: everybody leaves
constexpr long long RB_TIMEOUT_TICKS = 50000000;  // 0.5 s of the 100 MHz wall clock (a time, not a count of polls)
// End of the iteration for the agent that just solved, folded into the solve's launch (tail != 0; the team schedule's
// non-restart iterations): what k_nest_post, k_status and k_advance do in launches of their own --
//   bit 0: V <- proj(V + gamma (X - Y)) on the own poses (gamma as k_nest_pre published it in scal[6]);
//   always: |X - XPrev|^2 of the own poses -> PART_B[2] of this workgroup (the layout of the fused RGD step: rel_src 1);
//   workgroup 0: the Nesterov scalars of every agent and the team's iteration counter advance (not with bit 2).
// No hand-off: every workgroup touches its own two poses only, and nothing here reads what the advance writes.
template <int R>
__device__ __forceinline__ void solve_tail(const AgentDev *__restrict__ agents, const AgentDev &ag, TeamDev *team, int tail,
                                           int num_robots, int restart_interval, int bx, int npose, int tid, int pj0, int pj1, int pj2 = -1) {
  if (tid < 64) {
    double rel = 0;
    if (tid < npose) {
      const size_t o = (size_t)(tid == 0 ? pj0 : (tid == 1 ? pj1 : pj2)) * 4 * R;
      double x[4 * R], xp[4 * R];
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) { x[i] = ag.buf[B_X][o + i]; xp[i] = ag.buf[B_XPREV][o + i]; }
      if (tail & 1) {
        const double gamma = ag.scal[6];
        double v[4 * R], y[4 * R];
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) { v[i] = ag.buf[B_V][o + i]; y[i] = ag.buf[B_Y][o + i]; }
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) v[i] += gamma * (x[i] - y[i]);
        polar_inplace<R>(v);
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) ag.buf[B_V][o + i] = v[i];
      }
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) { const double d = x[i] - xp[i]; rel += d * d; }
    }
    rel = wave_sum(rel);
    if (tid == 0) ag.part[PART_B + (size_t)bx * PART_STRIDE + 2] = rel;
  }
  if (bx == 0 && tid == 0 && !(tail & 4)) {  // (bit 2: the per-agent API -- the caller's report advances this agent alone)
    for (int k = 0; k < team->num_agents; ++k) advance_agent(agents[k], tail & 1, num_robots, restart_interval);
    team->iter += 1;
  }
}

constexpr int RTR_WS_PITCH = 512;  // doubles per partial-sum array of the scratch (one entry per workgroup):
                                   // [0] <delta, H delta>  [1] <z, r>  [2] <r, r>  [3..6] f, |grad|^2, <g, eta>, <eta, H eta>

// optional per-phase timestamps of workgroup 0, wave 0 (build with -DDPGO_RTR_TRACE): bar[RB_TRACE + k]
constexpr int RB_TRACE = 17 * RB_LINE + 2;
#ifdef DPGO_RTR_TRACE
#define RTR_STAMP(k) do { if (tid == 0 && bx == 0 && (k) < 60) bar[RB_TRACE + (k)] = wall_clock64(); } while (0)
#define RTR_FINE(k) do { if (tid == 0 && bx == 0 && fine_on) bar[RB_TRACE + 64 + (k)] = wall_clock64(); } while (0)
#else
#define RTR_STAMP(k) do { } while (0)
#define RTR_FINE(k) do { } while (0)
#endif

constexpr int RB_TLX = 28 * RB_LINE;        // two-level exchange: one arrival counter per XCD at RB_TLX + g * RB_LINE
constexpr int RB_TLX_EPOCH = 36 * RB_LINE;  // exchanges completed so far (carried across launches)
struct GridBar {
  unsigned long long *bar;
  unsigned long long epoch;
  unsigned long long tlx_epoch;  // two-level solve: exchanges of u so far
  int g, size_g, ngroups, total;
  int *err;
  int *ok;  // LDS word: 0 once this workgroup's hand-off timed out
};

// DPGO_RTR_FLATBAR=1: every workgroup adds itself to its shard counter and leaves when the SUM of the eight shard
// counters (read as one batch of L1-bypassing loads) has reached total x epoch -- the last arrival is one atomic and one
// poll away from everybody, where the counter tree (shard -> top -> generation words) puts three dependent round trips
// between them.  Measured against each other (0.304 against 0.291 ms per iteration: the tree stays): profiles/r04_rtr_phases.md.
#ifndef DPGO_RTR_FLATBAR
#define DPGO_RTR_FLATBAR 0
#endif

// every workgroup of the launch arrives, every workgroup leaves after the last arrival.  Callers publish with st_c
// (write-through) and read the others' data with CVec loads afterwards.  Returns false when the hand-off timed out:
// the grid is not resident at once (another process holds CUs with a persistent kernel of its own).  That can only
// happen at the FIRST hand-off of a launch -- once it completes, every workgroup is resident for good -- and before
// it nothing but scratch has been written, so the caller leaves, the workgroups that start later find the abort word
// and leave too, and the host repeats the solve with the launch-per-step kernels.
__device__ __forceinline__ bool grid_sync(GridBar &gb, unsigned long long *tr = nullptr) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // every wave: its write-through stores have left the CU
  if (tr && threadIdx.x == 0) tr[0] = wall_clock64();
  __syncthreads();
  if (tr && threadIdx.x == 0) tr[1] = wall_clock64();
  gb.epoch += 1ull;
#if DPGO_RTR_FLATBAR
  if (threadIdx.x == 0) {
#if DPGO_RTR_PLAIN_ST
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    (void)__hip_atomic_fetch_add(&gb.bar[gb.g * RB_LINE], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tr) tr[2] = wall_clock64();
    const unsigned long long target = (unsigned long long)gb.total * gb.epoch;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gb.bar, 0, 8 * RB_LINE * 8, 0x00020000);
    const long long t_start = (long long)wall_clock64();
    while (true) {
      v2u_t w[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) w[q] = __builtin_amdgcn_raw_buffer_load_b64(rs, q * RB_LINE * 8, 0, 16);  // sc1
      unsigned long long sum = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) sum += ((unsigned long long)w[q].y << 32) | (unsigned long long)w[q].x;
      if (sum >= target) break;
      __builtin_amdgcn_s_sleep(1);
      if ((long long)wall_clock64() - t_start > RB_TIMEOUT_TICKS) {
        *gb.err = 2;
        *gb.ok = 0;
        __hip_atomic_store(&gb.bar[RB_ABORT], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    if (tr) tr[3] = wall_clock64();
#if DPGO_RTR_PLAIN_LD
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  }
  __syncthreads();
  return *gb.ok != 0;
#else
  if (threadIdx.x == 0) {
#if DPGO_RTR_PLAIN_ST
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    const unsigned long long old =
        __hip_atomic_fetch_add(&gb.bar[gb.g * RB_LINE], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1ull == (unsigned long long)gb.size_g * gb.epoch) {
      const unsigned long long old2 =
          __hip_atomic_fetch_add(&gb.bar[RB_TOP], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old2 + 1ull == (unsigned long long)gb.ngroups * gb.epoch) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
          __hip_atomic_store(&gb.bar[RB_GEN + q * RB_LINE], gb.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (tr) tr[2] = wall_clock64();
    const long long t_start = (long long)wall_clock64();
    while (__hip_atomic_load(&gb.bar[RB_GEN + gb.g * RB_LINE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gb.epoch) {
      __builtin_amdgcn_s_sleep(1);
      if ((long long)wall_clock64() - t_start > RB_TIMEOUT_TICKS) {
        *gb.err = 2;
        *gb.ok = 0;
        __hip_atomic_store(&gb.bar[RB_ABORT], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    if (tr) tr[3] = wall_clock64();
#if DPGO_RTR_PLAIN_LD
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  }
  __syncthreads();
  return *gb.ok != 0;
#endif
}

// The exchange inside a two-level apply is not a barrier: the nA producers publish u, everybody needs all of u, nobody
// needs anything from the consumers.  Producers count themselves in on eight counters (one per XCD, fire and forget);
// every workgroup polls its XCD's counter -- consumers arrive nowhere and wait for nobody but the producers (a full grid
// hand-off here until round 4: an atomic round trip, the counter tree and the generation words for 400 workgroups).
// u is rewritten by the next apply only behind at least one full hand-off of the solve, so no reader is overtaken.
__device__ __forceinline__ bool tl_exchange(GridBar &gb, bool producer, int nA) {
  gb.tlx_epoch += 1ull;
  if (producer) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this workgroup's write-through stores of u have left the CU
    __syncthreads();
    if (threadIdx.x < 8)
      (void)__hip_atomic_fetch_add(&gb.bar[RB_TLX + threadIdx.x * RB_LINE], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) {
    const unsigned long long target = (unsigned long long)nA * gb.tlx_epoch;
    const long long t_start = (long long)wall_clock64();
    while (__hip_atomic_load(&gb.bar[RB_TLX + gb.g * RB_LINE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((long long)wall_clock64() - t_start > RB_TIMEOUT_TICKS) {
        *gb.err = 3;
        *gb.ok = 0;
        __hip_atomic_store(&gb.bar[RB_ABORT], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
  return *gb.ok != 0;
}

// sum of `count` contiguous partials published by the workgroups of THIS launch, same order in every wave
// (CSUM_U partials per lane: 4 for the dense solve's <= 256 workgroups, 8 for the two-level solve's <= 512)
template <int NA, int CSUM_U>
__device__ __forceinline__ void csum_issue(const CVec &ws, int first, int count, int lane, double (*v)[CSUM_U]) {
  // straight-line (a predicated load is waited for on its own): entries beyond `count` re-read the last one, times 0
#pragma unroll
  for (int q = 0; q < NA; ++q)
#pragma unroll
    for (int u = 0; u < CSUM_U; ++u) v[q][u] = ws.ld((first + q) * RTR_WS_PITCH + min(lane + 64 * u, count - 1));
}
template <int NA, int CSUM_U>
__device__ __forceinline__ void csum_finish(double (*v)[CSUM_U], int count, int lane, double *out) {
#pragma unroll
  for (int q = 0; q < NA; ++q) {
#pragma unroll
    for (int u = 0; u < CSUM_U; ++u) v[q][u] *= (lane + 64 * u < count) ? 1.0 : 0.0;
    double s = (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
    if constexpr (CSUM_U == 8) s += (v[q][4] + v[q][5]) + (v[q][6] + v[q][7]);
    out[q] = wave_sum(s);
  }
}

// A vector that is written ONCE per launch at its address (the ring of H delta buffers below) needs no L1-bypassing
// loads: no stale copy of it can sit in this CU's L1 or this XCD's L2, so ordinary loads fetch it from the fabric once per
// XCD and serve the XCD's other 31 workgroups from L2 (sc1 loads bring all 80 KB to every workgroup: 20 MB per apply).
struct PVec {
  const double *p;
  __device__ __forceinline__ explicit PVec(const double *q) : p(q) {}
  __device__ __forceinline__ double ld(int i) const { return p[i]; }
  __device__ __forceinline__ double2 ld2(int i) const { return *reinterpret_cast<const double2 *>(p + i); }
};
constexpr int RTR_RING_MASK = RTR_RING - 1;
static_assert((RTR_RING & RTR_RING_MASK) == 0, "ring size is a power of two");

// The NW x 64 per-lane sums of the 8R outputs of a slab product -> zs (valid in every wave on return).
// Reduce-scatter across the four 16-lane rows of a wave with the gfx950 lane swaps: v_permlane32_swap exchanges the
// upper half of one register with the lower half of another, so ONE add then leaves the two-half sum of output i in
// lanes 0-31 and that of output i + 4R in lanes 32-63 (8R values -> 4R); v_permlane16_swap does the same for odd / even
// rows (4R -> 2R).  Row r of the wave then holds outputs [2R (r & 1) + 4R (r >> 1), + 2R): a DPP butterfly over its 16
// lanes, the first lane of each row stores 2R sums, and after ONE barrier one lane per output adds the NW wave totals.
// (Round 3 / early round 4: a quad butterfly on all 8R values, 16 LDS rows per wave and a 64-term sum per output --
// 4000 of the 5600 shader clocks of a product, profiles/experiments/slab_bench.hip.)
__device__ __forceinline__ double swap_add_32(double x, double y) {
  const auto l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ double swap_add_16(double x, double y) {
  const auto l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
template <int R, int NW, int NC = 8>
__device__ __forceinline__ void reduce_8R(double (*acc)[R], double *red, double *zs, int tid) {
  constexpr int N = NC * R, H = N / 2, Q = N / 4;  // NC columns: 8 (two poses per workgroup) or 12 (three)
  static_assert(N % 4 == 0, "two halvings");
  const int lane = tid & 63, wave = tid >> 6;
  double s[H], u[Q];
#pragma unroll
  for (int i = 0; i < H; ++i) s[i] = swap_add_32(acc[i / R][i % R], acc[(i + H) / R][(i + H) % R]);
#pragma unroll
  for (int i = 0; i < Q; ++i) u[i] = swap_add_16(s[i], s[i + Q]);
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    double x = u[i];
    x += dpp_move<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_move<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_move<0x141>(x);  // row_half_mirror
    x += dpp_move<0x140>(x);  // row_mirror
    u[i] = x;
  }
  if ((lane & 15) == 0) {
    const int row = lane >> 4;
    double *dst = red + wave * N + Q * (row & 1) + H * (row >> 1);
#pragma unroll
    for (int i = 0; i < Q; ++i) dst[i] = u[i];
  }
  __syncthreads();
  if (tid < N) {
    double t[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) t[w] = red[w * N + tid];
    if constexpr (NW == 4) zs[tid] = (t[0] + t[1]) + (t[2] + t[3]);
    else zs[tid] = t[0] + t[1];
  }
}

// zs[c * R + a] = sum_k V[k][a] * Ms[c][k]  for the 8 columns of the slab: lane t takes rows 2t, 2t+1 (+512 m) of V
// for ALL 8 columns, so every row of V is fetched once per workgroup (16-byte L1-bypassing loads straight from L2,
// no LDS staging) and the slab is read from LDS exactly once.  The 256 per-lane sums go through a quad reduction
// (DPP) and 64 LDS rows added in row order by one lane per output.  The result is valid in wave 0 on return.
constexpr int SLAB_MAXM = 4;
// straight-line code: rows beyond N4 read row N4 - 2 again and are multiplied by zero (one wave per SIMD: nothing
// hides a wait, so every load of a step is in flight before the first FMA)
template <int R, int MAXM, class Vec>
__device__ __forceinline__ void slab_issue(int N4, const Vec &V, int tid, double2 (*v)[R]) {
#pragma unroll
  for (int m = 0; m < MAXM; ++m) {
    const int k = 2 * tid + 512 * m, kk = min(k, N4 - 2);
#pragma unroll
    for (int q = 0; q < R; ++q) v[m][q] = V.ld2(kk * R + 2 * q);
  }
}
// NC columns (4 per own pose), the first CL of them in LDS, the others -- where the slab of a 3-pose workgroup does not
// fit LDS -- in the lane's own registers (sreg[column - CL][m]: the lane only ever meets ITS rows of a column, so what
// does not fit LDS needs no sharing at all)
template <int R, int NC = 8, int CL = 8, int MAXM = SLAB_MAXM>
__device__ __forceinline__ void slab_finish(const double *Ms, const double2 (*sreg)[MAXM], int N4, double2 (*v)[R], double *red,
                                            double *zs, int tid, unsigned long long *tr = nullptr) {
#ifdef DPGO_RTR_TRACE
  if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (tid == 0) { tr[0] = wall_clock64(); tr[3] = __builtin_amdgcn_s_memtime(); } }  // the vector has arrived
#endif
  double acc[NC][R];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int a = 0; a < R; ++a) acc[c][a] = 0.0;
#pragma unroll
  for (int m = 0; m < MAXM; ++m) {
    const int k = 2 * tid + 512 * m, kk = min(k, N4 - 2);
    const double live = (k < N4) ? 1.0 : 0.0;
    double2 mm[NC];
#pragma unroll
    for (int c = 0; c < CL; ++c) mm[c] = *reinterpret_cast<const double2 *>(&Ms[(size_t)c * N4 + kk]);
    if constexpr (NC > CL) {
#pragma unroll
      for (int c = CL; c < NC; ++c) mm[c] = sreg[c - CL][m];
    }
    double w[2 * R];
#pragma unroll
    for (int q = 0; q < R; ++q) { w[2 * q] = v[m][q].x * live; w[2 * q + 1] = v[m][q].y * live; }
    // two fused multiply-adds per accumulator, NC x R independent accumulators: `acc += w0 * m.x + w1 * m.y` compiles to
    // mul, fma, add through ONE temporary -- a dependent chain per accumulator that one wave per SIMD cannot hide
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int a = 0; a < R; ++a) acc[c][a] = __builtin_fma(w[a], mm[c].x, acc[c][a]);
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int a = 0; a < R; ++a) acc[c][a] = __builtin_fma(w[R + a], mm[c].y, acc[c][a]);
  }
#ifdef DPGO_RTR_TRACE
  if (tr && tid == 0) { tr[1] = wall_clock64(); tr[4] = __builtin_amdgcn_s_memtime(); }  // products done
#endif
  reduce_8R<R, 4, NC>(acc, red, zs, tid);
#ifdef DPGO_RTR_TRACE
  if (tr && tid == 0) { tr[2] = wall_clock64(); tr[5] = __builtin_amdgcn_s_memtime(); }  // sums in zs
#endif
  if (tid < 64) WSYNC();
}

// spmm_row (kernel_common.h) for a pose whose first <= 8 ELL slots (index + 4 x 4 block) sit in LDS: every gather of
// the row is requested before the first use -- one round trip for the whole row instead of index -> gather per group
// of four slots.  Rows longer than the cached slots continue in the CSR tail from global memory.
constexpr int RTR_SLOTS = 8;

// RtrState without the array member: every field is a scalar the compiler keeps in registers (the tcg_o[] array of
// RtrState, indexed by the outer iteration, forced the whole record into scratch memory: 210 scratch accesses on the
// scalar path of every phase)
struct SolveState {
  double f1, ngf, Delta;
  double z_r, d_Pd, e_Pd, e_Pe, norm_r0, alpha;
  double f_init, gn_init;
  int outer_it, outer_done, tcg_active, tcg_j, tcg_status, need_init;
  int hv_count, pc_count, tcg_total, accepted, outer_count;
  int o0, o1, o2, o3;
};
__device__ __forceinline__ RtrState to_record(const SolveState &S) {
  RtrState T = {};
  T.f1 = S.f1; T.ngf = S.ngf; T.Delta = S.Delta;
  T.z_r = S.z_r; T.d_Pd = S.d_Pd; T.e_Pd = S.e_Pd; T.e_Pe = S.e_Pe; T.norm_r0 = S.norm_r0; T.alpha = S.alpha;
  T.f_init = S.f_init; T.gn_init = S.gn_init;
  T.outer_it = S.outer_it; T.outer_done = S.outer_done; T.tcg_active = S.tcg_active; T.tcg_j = S.tcg_j;
  T.tcg_status = S.tcg_status; T.need_init = S.need_init;
  T.hv_count = S.hv_count; T.pc_count = S.pc_count; T.tcg_total = S.tcg_total; T.accepted = S.accepted;
  T.outer_count = S.outer_count;
  T.tcg_o[0] = S.o0; T.tcg_o[1] = S.o1; T.tcg_o[2] = S.o2; T.tcg_o[3] = S.o3;
  return T;
}
template <int NRAW, class Ld>
__device__ __forceinline__ void gather_issue(const int *idxL, Ld ld, double (*raw)[NRAW][4]) {
  // straight-line: slots the matrix does not have are cached as (own pose, zero block)
  int idx[RTR_SLOTS];
#pragma unroll
  for (int u = 0; u < RTR_SLOTS; ++u) idx[u] = idxL[u];
#pragma unroll
  for (int u = 0; u < RTR_SLOTS; ++u) ld(idx[u], raw[u]);
}
// ld: pose index -> NRAW x 4 raw operands (row a of the pose); comb: raw operands -> the NV input vectors
template <int R, int NV, int NRAW, class Ld, class Comb>
__device__ __forceinline__ void spmm_row_finish(const AgentDev &ag, int j, const double *BL, double (*raw)[NRAW][4], Ld ld,
                                                Comb comb, double (*acc)[4], int ell_w, int p0, int p1) {
  double x[RTR_SLOTS][NV][4];
#pragma unroll
  for (int u = 0; u < RTR_SLOTS; ++u) comb(raw[u], x[u]);
  auto src = [&](int i, double(*xo)[4]) {
    double rr[NRAW][4];
    ld(i, rr);
    comb(rr, xo);
  };
  // the 4 x 4 blocks come from LDS one slot ahead of their multiply-adds (one wave per SIMD: a ds_read that is waited
  // for where it is used costs its full latency, and there were 64 of them in a row)
  double2 Bn[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) Bn[q] = *reinterpret_cast<const double2 *>(BL + 2 * q);
#pragma unroll
  for (int u = 0; u < RTR_SLOTS; ++u) {
    double B[16];
#pragma unroll
    for (int q = 0; q < 8; ++q) { B[2 * q] = Bn[q].x; B[2 * q + 1] = Bn[q].y; }
    if (u + 1 < RTR_SLOTS) {
#pragma unroll
      for (int q = 0; q < 8; ++q) Bn[q] = *reinterpret_cast<const double2 *>(BL + 16 * (u + 1) + 2 * q);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc[v][c] = fma4(x[u][v][0], B[4 * c], x[u][v][1], B[4 * c + 1], x[u][v][2], B[4 * c + 2], x[u][v][3], B[4 * c + 3], acc[v][c]);
  }
  // (matrix width and the CSR-tail range of the row come from the caller: read once per launch, not once per product)
  for (int s0 = RTR_SLOTS; s0 < ell_w; s0 += 4) ell_group<R, NV>(ag, j, s0, src, acc);
  for (int p = p0; p < p1; ++p) {
    const int i = ag.tcol[p];
    const double *bp = ag.tval + (size_t)16 * p;
    double xt[NV][4];
    src(i, xt);
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const double2 b01 = ld2(bp + 4 * c), b23 = ld2(bp + 4 * c + 2);
        acc[v][c] = fma4(xt[v][0], b01.x, xt[v][1], b01.y, xt[v][2], b23.x, xt[v][3], b23.y, acc[v][c]);
      }
  }
}

// hess_tail (kernel_common.h) for lanes of ONE wave
// Hc[3 p + q] = 0.5 (S_pq + S_qp), S = Y^T E of the pose: fixed over a tCG solve, formed once per outer iteration by
// curvature_block (vrow[p] * 0.5 * (S_pq + S_qp) and vrow[p] * Hc are the same number: the halving is exact)
template <int R>
__device__ __forceinline__ void curvature_block(const double *Ysh, const double *Esh, double *Hc) {
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
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < 3; ++q) Hc[3 * p + q] = 0.5 * (S[3 * p + q] + S[3 * q + p]);
}
template <int R>
__device__ __forceinline__ void hess_tail_w(const double *Ysh, const double *Hc, double *Wsh, int a, const double wrow[4],
                                            const double vrow[4], double hrow[4], bool act) {
  if (act) {
    double h[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) h[i] = Hc[i];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      double s = wrow[q];
#pragma unroll
      for (int p = 0; p < 3; ++p) s -= vrow[p] * h[3 * p + q];
      Wsh[q * R + a] = s;
    }
  }
  WSYNC();
  if (act) {
    double o[3];
    tangent_row<R>(Ysh, Wsh, a, o);
    hrow[0] = o[0]; hrow[1] = o[1]; hrow[2] = o[2]; hrow[3] = wrow[3];
  }
  WSYNC();
}


// ---- the two-level form of the preconditioner (twolevel.h) inside the persistent solve: the workgroup's slab -- rows
// that meet the input vector (its subdomain's D_i, or -E_i of the adjacent subdomains for a workgroup that owns
// separator poses), then the separator rows (W_i or Sc^-1) -- sits in LDS for the whole solve: 37 KB instead of the
// 128 KB of a dense slab, which is what lets agents of 513 .. 1024 poses keep the one-launch solve (two workgroups per
// CU).  An apply is twolevel_dev.h's product with the exchange carried by one more grid hand-off: the producers publish
// u, everybody re-reads it.  Lane t takes row pairs t + 128 m; which rows of the vector they meet is fixed over the solve.
// The two-level solve runs 128-thread workgroups: two of them share a CU with ONE wave per SIMD each, i.e. with the whole
// register file the solve's gather phases need (two 256-thread workgroups per CU would halve it and spill).
constexpr int TLS_NT = 128;
constexpr int TLS_NPRE = 4, TLS_NPOST = 8;  // passes of 128 row pairs: subdomains of <= 256 poses, <= 512 separator poses
struct TLLane {
  int voff[TLS_NPRE];     // offset (doubles) of the pair's two rows in an r x 4n vector
  double vlive[TLS_NPRE]; // 0 for lanes past the workgroup's rows
};
// the 128 per-lane sums of the 8R outputs -> zs (valid in both waves): reduce_8R over two waves
template <int R>
__device__ __forceinline__ void tls_reduce(double (*acc)[R], double *red, double *zs, int tid) {
  reduce_8R<R, 2>(acc, red, zs, tid);
  __syncthreads();
}
template <int R, class Vec>
__device__ __forceinline__ bool tl_product_lds(const double *Ms, const TLDev &tl, const TLWg &w, int b, const TLLane &ln,
                                               const Vec &V, GridBar &gb, double *red, double *zs, int tid) {
  const int npre = 2 * w.pre_cnt, npost = 2 * tl.ns;
  const bool producer = b < tl.nA;
  double acc[8][R];
  tl_zero<R>(acc);
  double vown = 0;
  if (producer && tid < 8 * R) {
    const int lp = tid / (4 * R), own = w.own[lp];
    if (own >= 0) vown = V.ld(own * 4 * R + (tid - lp * 4 * R));
  }
  // rows that meet the input vector: two passes per batch (every request of a batch in flight before its first product)
#pragma unroll 1
  for (int m0 = 0; m0 < TLS_NPRE && m0 * TLS_NT < npre; m0 += 2) {
    double2 v[2][R];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int a = 0; a < R; ++a) v[u][a] = V.ld2(ln.voff[m0 + u] + 2 * a);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int qq = min(tid + TLS_NT * (m0 + u), max(npre - 1, 0));
      double2 mm[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) mm[c] = *reinterpret_cast<const double2 *>(&Ms[(size_t)qq * 16 + 2 * c]);
      tl_fma<R>(acc, v[u], mm, ln.vlive[m0 + u]);
    }
  }
  if (npost == 0) {
    tls_reduce<R>(acc, red, zs, tid);
    return true;
  }
  if (producer) {
    tls_reduce<R>(acc, red, zs, tid);
    if (tid < 8 * R) {
      const int lp = tid / (4 * R), e = tid - lp * 4 * R;
      if (w.own[lp] >= 0) st_c(tl.u + ((size_t)(w.sep0 + lp) * 4 * R + e), vown + zs[tid]);
    }
    tl_zero<R>(acc);
  }
  if (!tl_exchange(gb, producer, tl.nA)) return false;
  const CVec cu(tl.u, 4 * tl.ns * R);
  const double *post = Ms + (size_t)npre * 16;
#pragma unroll 1
  for (int m0 = 0; m0 < TLS_NPOST && m0 * TLS_NT < npost; m0 += 2) {
    double2 v[2][R];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int qq = min(tid + TLS_NT * (m0 + u), npost - 1);
#pragma unroll
      for (int a = 0; a < R; ++a) v[u][a] = cu.ld2(qq * 2 * R + 2 * a);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = tid + TLS_NT * (m0 + u), qq = min(q, npost - 1);
      double2 mm[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) mm[c] = *reinterpret_cast<const double2 *>(&post[(size_t)qq * 16 + 2 * c]);
      tl_fma<R>(acc, v[u], mm, q < npost ? 1.0 : 0.0);
    }
  }
  tls_reduce<R>(acc, red, zs, tid);
  return true;
}

// TL: the agent runs the two-level form of the preconditioner (slab layout and ownership of twolevel.h; up to two
// workgroups per CU) instead of the dense inverse
// NP: poses per workgroup of the dense solve.  2: 8 columns of M in LDS (agents of up to 512 poses).  3: 12 columns, 7 of
// them in LDS and 5 in the lanes' registers -- a lane of the slab product only ever meets its own rows of a column, so
// the part of the slab that LDS cannot hold needs no sharing (agents of 513 .. 640 poses: the torus3D GNC size; round 4.
// Until then those agents ran the two-level form below: one more hand-off and 17 instead of 12 us per tCG iteration)
template <int R, bool TL, int NP = 2>
__global__ __launch_bounds__(TL ? TLS_NT : 256) void k_rtr_solve(const AgentDev *__restrict__ agents, int ai, unsigned long long *bar, double *ws,
                                                   unsigned long long *cum, RtrState *host_rec,
                                                   unsigned long long *host_cum, int *err, double Delta0, double tol, int max_outer, int max_inner,
                                                   double max_radius, TeamDev *team, int tail, int num_robots,
                                                   int restart_interval, const AgentDev agv) {
  // (16-byte aligned: the static arrays in front of it end on an odd multiple of 8 bytes, and every 16-byte ds_read of a
  // slab that starts there is split by the hardware -- the slab product ran at a sixth of its speed until round 4)
  static_assert(NP == 2 || (NP == 3 && !TL), "two poses per workgroup, or three in the dense solve");
  constexpr int NC = 4 * NP, NO = NC * R;                 // columns of M / outputs of a slab product
  constexpr int MAXM = NP == 3 ? 5 : SLAB_MAXM;           // row pairs per lane: N4 <= 512 MAXM
  constexpr int CL = NP == 3 ? 7 : NC, CR = NC - CL;      // columns in LDS / in registers
  extern __shared__ __attribute__((aligned(16))) double Ms[];  // [CL][N4]: this workgroup's columns of M
  __shared__ double red[(TL ? 2 : 4) * NO];  // one row of wave totals per wave (reduce_8R)
  // own poses, [pose][component c][row a]: X, Euclidean / Riemannian gradient at X; tCG residual, z, delta, eta;
  // scratch; candidate point and its gradients
  __shared__ double zs[NO], Xs[NO], Es[NO], Gs[NO], Rs[NO], Zo[NO], Ds[NO], Et[NO], Ws[NO], X2s[NO], E2s[NO], G2s[NO];
  __shared__ double BL[NP * RTR_SLOTS * 16];  // the first ELL slots of the own poses: 4 x 4 blocks and indices
  __shared__ int idxL[NP * RTR_SLOTS];
  __shared__ double Hcs[NP * 9];  // curvature blocks of the own poses at the current X
  // (the agent's descriptor BY VALUE: pointers read from the agents array are generic pointers to the compiler, their
  // loads flat loads -- out of order, so every wait for one of them is a wait for all loads and LDS operations in flight;
  // 245 of them in this kernel until round 5)
#ifdef DPGO_RTR_DESC_MEM
  const AgentDev &ag = agents[ai];
#else
  const AgentDev &ag = agv;
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bx = (int)blockIdx.x, N4 = ag.N4, n = ag.n;
  const int nblk = TL ? ag.tl.nwg - ag.tl.nS2 : (n + NP - 1) / NP;  // == gridDim.x
  // the two poses this workgroup owns: consecutive ones, or the pair the two-level layout assigns.  Two-level: the
  // solve's workgroups are the layout's producers (one separator pose each: here they also take that pose's column of
  // Sc^-1, which their slabs carry, TLDev::prod_post) and its interior workgroups; `tb` is the workgroup's index in the
  // layout's tables (the nS2 workgroups that own the separator columns in a stand-alone apply are skipped)
  TLWg tlw = {};
  const int tb = (TL && bx >= ag.tl.nA) ? bx + ag.tl.nS2 : bx;
  int pj0 = NP * bx, pj1 = (NP * bx + 1 < n) ? NP * bx + 1 : -1;
  const int pj2 = (NP == 3 && NP * bx + 2 < n) ? NP * bx + 2 : -1;
  if constexpr (TL) { tlw = ag.tl.wg[tb]; pj0 = tlw.own[0]; pj1 = tlw.own[1]; }
  const int npose = (pj2 >= 0) ? 3 : ((pj1 >= 0) ? 2 : 1);
  const int lp = tid / R, a = tid - lp * R;
  const bool rl = tid < npose * R;  // row lanes: one per (own pose, row of the lifted pose), all in wave 0
  const int j = (lp == 2 && pj2 >= 0) ? pj2 : ((lp == 1 && pj1 >= 0) ? pj1 : pj0);
  constexpr int CSUM_U = TL ? 8 : 4;
  double *wsA = ws, *wsB0 = ws + RTR_WS_PITCH, *wsB1 = ws + 2 * RTR_WS_PITCH, *wsC = ws + 3 * RTR_WS_PITCH;

  // ---- trust-region set-up from the partial sums of the evaluation launched in front (k_rtr_begin)
  SolveState S = {};
  {
    const int nb = spmm_blocks<R>(n);
    const double f = sum_partials(ag.part + PART_A, nb, PART_STRIDE, lane);
    const double g = sum_partials(ag.part + PART_A + 1, nb, PART_STRIDE, lane);
    S.f1 = f; S.ngf = sqrt(g); S.Delta = Delta0;
    S.f_init = f; S.gn_init = S.ngf;
    S.outer_done = (S.ngf < tol) || (max_outer <= 0);
    S.need_init = 1;
  }
  if (S.outer_done) {
    if (bx == 0 && tid == 0) {
      const RtrState T = to_record(S);
      ag.st[0] = T; ag.st[1] = T;
      cum[0] += 1ull;
      *host_rec = T;
      for (int k = 0; k < 4; ++k) host_cum[k] = cum[k];
    }
    if (tail) solve_tail<R>(agents, ag, team, tail, num_robots, restart_interval, bx, npose, tid, pj0, pj1, pj2);
    return;
  }
  __shared__ int bar_ok;
  if (tid == 0) bar_ok = __hip_atomic_load(&bar[RB_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull;
  __syncthreads();
  if (!bar_ok) return;  // a workgroup of this launch gave up before this one started
  GridBar gb;
  gb.bar = bar; gb.err = err; gb.ok = &bar_ok;
  gb.epoch = bar[RB_EPOCH];
  gb.tlx_epoch = TL ? bar[RB_TLX_EPOCH] : 0ull;
  gb.g = bx & 7;
  gb.size_g = (nblk - gb.g + 7) / 8;
  gb.ngroups = min(8, nblk);
  gb.total = nblk;
#ifdef DPGO_RTR_TRACE
  if (tid == 0 && bx == 0) for (int k = 0; k < 60; ++k) bar[RB_TRACE + k] = 0ull;
#endif
  RTR_STAMP(0);

  // ---- the slab, HBM -> LDS, once: the workgroup's slab of the two-level form ...
  TLLane tln = {};
  if constexpr (TL) {
    const int cnt2 = (2 * tlw.pre_cnt + 2 * ag.tl.ns) * 8;  // double2 elements
    const double *src = ag.tl.slabs + tlw.slab_off;
    for (int base = 0; base < cnt2; base += 8 * TLS_NT) {
      double2 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = ld2_nt(src + 2 * (size_t)min(base + tid + TLS_NT * u, cnt2 - 1));
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + tid + TLS_NT * u;
        if (i < cnt2) *reinterpret_cast<double2 *>(&Ms[2 * (size_t)i]) = t[u];
      }
    }
    const int npre = 2 * tlw.pre_cnt;
    const int *rp = ag.tl.rowpose + (size_t)tb * ag.tl.rp_stride;
#pragma unroll
    for (int m = 0; m < TLS_NPRE; ++m) {
      const int q = tid + TLS_NT * m, qq = max(min(q, npre - 1), 0);
      tln.voff[m] = (4 * rp[qq >> 1] + 2 * (qq & 1)) * R;
      tln.vlive[m] = q < npre ? 1.0 : 0.0;
    }
  } else if constexpr (NP == 2) {
    // ... or 8 contiguous columns of the dense M
    const int tot2 = 2 * npose * N4;  // double2 elements of the valid columns
    const double *Msrc = ag.M + (size_t)8 * bx * N4;
    // straight-line: 32 x 16 bytes per lane (N4 <= 2048), every request in flight before the first LDS store; indices
    // past the valid columns re-read the last element and store zeros (predicated loads were waited for one by one:
    // 8.5 us for the 32 MB)
    double2 t[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) t[u] = ld2_nt(Msrc + 2 * (size_t)min(tid + 256 * u, tot2 - 1));
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int i = tid + 256 * u;
      if (i < 4 * N4) *reinterpret_cast<double2 *>(&Ms[2 * (size_t)i]) = (i < tot2) ? t[u] : make_double2(0.0, 0.0);
    }
  }
  // ... or 12: the first CL columns into LDS (three passes of 12 x 16 bytes per lane), the lane's rows of the other CR
  // into its registers for the whole solve
  double2 sreg[CR > 0 ? CR : 1][MAXM];
  if constexpr (NP == 3) {
    const int valid2 = 2 * npose * N4;               // double2 elements of the columns that exist (the last workgroup)
    const int tot2 = CL * (N4 / 2);                  // double2 elements of the LDS part
    const double *Msrc = ag.M + (size_t)NC * bx * N4;
    for (int base = 0; base < tot2; base += 256 * 12) {
      double2 t[12];
#pragma unroll
      for (int u = 0; u < 12; ++u) t[u] = ld2_nt(Msrc + 2 * (size_t)min(base + tid + 256 * u, min(tot2, valid2) - 1));
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        const int i = base + tid + 256 * u;
        if (i < tot2) *reinterpret_cast<double2 *>(&Ms[2 * (size_t)i]) = (i < valid2) ? t[u] : make_double2(0.0, 0.0);
      }
    }
#pragma unroll
    for (int u = 0; u < CR; ++u)
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        const int c = CL + u, cc = min(c, 4 * npose - 1), kk = min(2 * tid + 512 * m, N4 - 2);
        const double2 t = ld2_nt(Msrc + (size_t)cc * N4 + kk);
        sreg[u][m] = (c < 4 * npose) ? t : make_double2(0.0, 0.0);
      }
  }
  if (rl) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const size_t o = ((size_t)4 * j + c) * R + a;
      const int idx = lp * 4 * R + c * R + a;
      Xs[idx] = ag.buf[B_X][o];
      Es[idx] = ag.buf[B_EGRAD][o];
      Gs[idx] = ag.buf[B_GF][o];
    }
  }
  const int ellw = ag.ell_w, Wc = min(ellw, RTR_SLOTS);
  const int tp0 = rl ? ag.trowptr[j] : 0, tp1 = rl ? ag.trowptr[j + 1] : 0;
#pragma unroll
  for (int q = tid; q < NP * RTR_SLOTS * 16; q += (TL ? TLS_NT : 256)) {
    const int pl = q / (RTR_SLOTS * 16), u = (q / 16) % RTR_SLOTS, e = q % 16;
    const int jj = (pl == 2 && pj2 >= 0) ? pj2 : ((pl == 1 && pj1 >= 0) ? pj1 : pj0);
    const bool have = pl < npose && u < Wc;
    BL[q] = have ? ag.ell_val[((size_t)u * n + jj) * 16 + e] : 0.0;
    if (e == 0) idxL[pl * RTR_SLOTS + u] = have ? ag.ell_col[(size_t)u * n + jj] : jj;
  }
  __syncthreads();
  RTR_STAMP(1);
  int stamp = 2;
  (void)stamp;
  bool fine_on = false;
  (void)fine_on;

  double *Zg = ag.buf[B_Z], *ETAg = ag.buf[B_ETA], *X2g = ag.buf[B_X2], *GFg = ag.buf[B_GF];
  const int NV8 = N4 * R;
  const CVec cGF(GFg, NV8), cX2(X2g, NV8), cETA(ETAg, NV8), cWS(ws, RTR_WS_PITCH * 7);
  // H delta of tCG iteration k of this launch lives in slot k mod RTR_RING of a ring behind the partial sums: an address
  // is written once and only then read (ordinary loads, PVec); when the ring wraps every workgroup drops what its L1 / L2
  // hold of the previous pass (one agent-scope acquire per RTR_RING iterations)
  // (slots are padded to whole 256-byte blocks: a cache line shared by two slots would be fetched with the first and
  // serve stale bytes of the second)
  double *ring = ws + RTR_WS_DOUBLES;
  const size_t ring_pitch = rtr_ring_pitch(NV8);
  int hd_slot = 0;
  const double kappa = 0.1;  // tCG stop: |r| <= |r0| min(|r0|^theta, kappa) with theta = 1

  int gf_fresh = 0;  // the gradient at X was published as the candidate's gradient (GF2) by the evaluation just accepted
  while (true) {
    if (wave == 0) WSYNC();  // (the acceptance step's LDS updates of X / egrad, written by the row lanes)
    if (tid < npose) curvature_block<R>(Xs + tid * 4 * R, Es + tid * 4 * R, Hcs + tid * 9);
    // (visible to the row lanes through the barriers inside the set-up's slab product)
    // ================= tCG set-up (k_precond<PM_TCG_INIT>): z0 = P(gf M), r0 = gf, eta = 0, delta0 = -z0
    {
      // after an accepted step the whole new gradient is already visible as GF2 (published before the hand-off in
      // front of the acceptance test), so no hand-off stands between the acceptance and this product; GF itself
      // (own rows written through at the acceptance) serves the set-ups that follow a rejection, hand-offs later
      const CVec cSrc(__builtin_amdgcn_readfirstlane(gf_fresh) ? ag.buf[B_GF2] : GFg, NV8);
      if constexpr (TL) {
        if (!tl_product_lds<R>(Ms, ag.tl, tlw, tb, tln, cSrc, gb, red, zs, tid)) return;
      } else {
        double2 vv[MAXM][R];
        slab_issue<R, MAXM>(N4, cSrc, tid, vv);
        slab_finish<R, NC, CL, MAXM>(Ms, sreg, N4, vv, red, zs, tid);
      }
    }
    {
      double zr = 0, rr = 0;
      if (rl) {
        double z[4];
        tangent_row<R>(Xs + lp * 4 * R, zs + lp * 4 * R, a, z);
        z[3] = zs[lp * 4 * R + 3 * R + a];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const size_t o = ((size_t)4 * j + c) * R + a;
          const int idx = lp * 4 * R + c * R + a;
          const double v = Gs[idx];
          Rs[idx] = v; Et[idx] = 0.0; Ds[idx] = -z[c]; Zo[idx] = z[c];
          st_c(ag.buf[B_D0] + o, -z[c]);
          st_c(Zg + o, z[c]);
          zr += z[c] * v;
          rr += v * v;
        }
      }
      if (wave == 0) {
        zr = wave_sum(zr); rr = wave_sum(rr);
        if (lane == 0) { st_c(wsB0 + bx, zr); st_c(wsB1 + bx, rr); }
      }
    }
    S.tcg_active = 1; S.tcg_j = 0; S.tcg_status = 0; S.need_init = 0;
    S.e_Pd = 0; S.e_Pe = 0; S.alpha = 0;
    S.pc_count += 1; S.outer_count += 1;
    if (!grid_sync(gb)) return;
    RTR_STAMP(stamp++);

    // ================= tCG iterations
    while (true) {
      // ---- part 1 (k_tcg_hv): delta <- -z + beta delta, H delta on the own poses, partial <delta, H delta>
#ifdef DPGO_RTR_TRACE
      fine_on = (S.outer_it == 0 && S.tcg_j == 1);
#endif
      RTR_FINE(0);
      // the partial sums and (speculatively: the stop test needs the sums) the neighbour rows of z and delta are
      // requested together -- one round trip to what the other workgroups just published instead of two
      const bool fresh = __builtin_amdgcn_readfirstlane(S.tcg_j) == 0;
      const int jp = __builtin_amdgcn_readfirstlane(S.tcg_j) & 1;
      const double *Dold = ag.buf[jp ? B_D0 : B_D1];  // delta of iteration j-1
      double *Dnew = ag.buf[jp ? B_D1 : B_D0];        // delta of iteration j (the set-up wrote D0 for j = 0)
      // fresh: delta_0 as the set-up published it; else -z + beta delta_{j-1} (the owner's own expression)
      const CVec cA(fresh ? Dnew : Zg, NV8), cB(fresh ? Dnew : Dold, NV8);
      auto ldA = [&](int i, double(*rw)[4]) {
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
          const int o = (4 * i + cp) * R + a;
          rw[0][cp] = cA.ld(o);
          rw[1][cp] = cB.ld(o);
        }
      };
      double sv[2][CSUM_U], sb[2];
      csum_issue<2, CSUM_U>(cWS, 1, nblk, lane, sv);
      double rawA[RTR_SLOTS][2][4];
      if (wave == 0) gather_issue<2>(idxL + min(lp, NP - 1) * RTR_SLOTS, ldA, rawA);
      csum_finish<2, CSUM_U>(sv, nblk, lane, sb);
      const double zr_new = sb[0], rr_new = sb[1];
      RTR_FINE(1);
      double beta = 0;
      if (fresh) {
        S.z_r = zr_new; S.d_Pd = zr_new; S.norm_r0 = sqrt(rr_new);
      } else {
        const double nr = sqrt(rr_new), thr = S.norm_r0;
        bool stop = false;
        if (nr <= S.norm_r0 * (thr < kappa ? thr : kappa)) { S.tcg_status = (kappa < thr) ? 3 : 4; stop = true; }
        else if (S.tcg_j >= max_inner) { S.tcg_status = 0; stop = true; }
        if (stop) { S.tcg_active = 0; break; }
        beta = zr_new / S.z_r;
        S.e_Pd = beta * (S.e_Pd + S.alpha * S.d_Pd);
        S.d_Pd = zr_new + beta * beta * S.d_Pd;
        S.z_r = zr_new;
      }
      S.hv_count += 1; S.tcg_total += 1;
      double *HDj = ring + (size_t)(hd_slot & RTR_RING_MASK) * ring_pitch;
      if (wave == 0) {
        double w[1][4] = {{0, 0, 0, 0}}, vrow[4] = {0, 0, 0, 0}, hrow[4] = {0, 0, 0, 0};
        if (rl) {
          spmm_row_finish<R, 1, 2>(ag, j, BL + lp * RTR_SLOTS * 16, rawA, ldA,
            [&](double(*rw)[4], double(*x)[4]) {
#pragma unroll
              for (int cp = 0; cp < 4; ++cp) x[0][cp] = fresh ? rw[1][cp] : (-rw[0][cp] + beta * rw[1][cp]);
            }, w, ellw, tp0, tp1);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const size_t o = ((size_t)4 * j + c) * R + a;
            const int idx = lp * 4 * R + c * R + a;
            vrow[c] = fresh ? Ds[idx] : (-Zo[idx] + beta * Ds[idx]);
            if (!fresh) { Ds[idx] = vrow[c]; st_c(Dnew + o, vrow[c]); }
          }
        }
        RTR_FINE(2);
        hess_tail_w<R>(Xs + lp * 4 * R, Hcs + min(lp, NP - 1) * 9, Ws + lp * 4 * R, a, w[0], vrow, hrow, rl);
        RTR_FINE(3);
        double d = 0;
        if (rl) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const size_t o = ((size_t)4 * j + c) * R + a;
            st_c(HDj + o, hrow[c]);
            Ws[lp * 4 * R + c * R + a] = hrow[c];  // own rows of H delta for the residual update below
            d += vrow[c] * hrow[c];
          }
        }
        d = wave_sum(d);
        if (lane == 0) st_c(wsA + bx, d);
      }
      RTR_FINE(4);
#ifdef DPGO_RTR_TRACE
      if (!grid_sync(gb, (bx == 0 && fine_on) ? bar + RB_TRACE + 64 + 5 : nullptr)) return;
#else
      if (!grid_sync(gb)) return;
#endif
      RTR_FINE(9);
      RTR_STAMP(stamp++);
      if (hd_slot >= RTR_RING && (hd_slot & RTR_RING_MASK) == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the ring wrapped
      hd_slot += 1;
      const PVec cHD(HDj);

      // ---- part 2 (k_precond<PM_TCG_STEP>): alpha, boundary test, eta += alpha delta, r += alpha H delta,
      //      z += alpha P(H delta M)
      double sv1[1][CSUM_U], d_Hd;
      csum_issue<1, CSUM_U>(cWS, 0, nblk, lane, sv1);
      double2 vv[TL ? 1 : MAXM][R];
      if constexpr (!TL) slab_issue<R, MAXM>(N4, cHD, tid, vv);  // (speculative: the boundary test below needs the sum)
      csum_finish<1, CSUM_U>(sv1, nblk, lane, &d_Hd);
      RTR_FINE(10);
      const double alpha = S.z_r / d_Hd;
      const double e_Pe_new = S.e_Pe + 2.0 * alpha * S.e_Pd + alpha * alpha * S.d_Pd;
      if (d_Hd <= 0 || e_Pe_new >= S.Delta * S.Delta) {
        const double tau = (-S.e_Pd + sqrt(S.e_Pd * S.e_Pd + S.d_Pd * (S.Delta * S.Delta - S.e_Pe))) / S.d_Pd;
        S.tcg_active = 0;
        S.tcg_status = (d_Hd <= 0) ? 1 : 2;
        if (rl) {
#pragma unroll
          for (int c = 0; c < 4; ++c) { const int idx = lp * 4 * R + c * R + a; Et[idx] += tau * Ds[idx]; }
        }
        break;
      }
      S.e_Pe = e_Pe_new; S.alpha = alpha; S.tcg_j += 1; S.pc_count += 1;
      if constexpr (TL) {
        if (!tl_product_lds<R>(Ms, ag.tl, tlw, tb, tln, cHD, gb, red, zs, tid)) return;
      } else {
#ifdef DPGO_RTR_TRACE
        slab_finish<R, NC, CL, MAXM>(Ms, sreg, N4, vv, red, zs, tid, (bx == 0 && fine_on) ? bar + RB_TRACE + 64 + 20 : nullptr);
#else
        slab_finish<R, NC, CL, MAXM>(Ms, sreg, N4, vv, red, zs, tid);
#endif
      }
      RTR_FINE(11);
      {
        double zr = 0, rr = 0;
        if (rl) {
          double z[4];
          tangent_row<R>(Xs + lp * 4 * R, zs + lp * 4 * R, a, z);
          z[3] = zs[lp * 4 * R + 3 * R + a];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const size_t o = ((size_t)4 * j + c) * R + a;
            const int idx = lp * 4 * R + c * R + a;
            Et[idx] += alpha * Ds[idx];
            const double v = Rs[idx] + alpha * Ws[idx];
            Rs[idx] = v;
            const double zn = Zo[idx] + alpha * z[c];
            Zo[idx] = zn;
            st_c(Zg + o, zn);
            zr += zn * v;
            rr += v * v;
          }
        }
        if (wave == 0) {
          zr = wave_sum(zr); rr = wave_sum(rr);
          if (lane == 0) { st_c(wsB0 + bx, zr); st_c(wsB1 + bx, rr); }
        }
      }
      RTR_FINE(12);
#ifdef DPGO_RTR_TRACE
      if (!grid_sync(gb, (bx == 0 && fine_on) ? bar + RB_TRACE + 64 + 13 : nullptr)) return;
#else
      if (!grid_sync(gb)) return;
#endif
      RTR_FINE(17);
      RTR_STAMP(stamp++);
    }

    // ================= candidate point (k_retract): X2 = Retr_X(eta) on the own poses, eta published
    if (wave == 0) {
      WSYNC();
      if (tid < npose) {
        double x[4 * R];
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) x[i] = Xs[tid * 4 * R + i] + Et[tid * 4 * R + i];
        qf_inplace<R>(x);
#pragma unroll
        for (int i = 0; i < 4 * R; ++i) X2s[tid * 4 * R + i] = x[i];
      }
      WSYNC();
      if (rl) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const size_t o = ((size_t)4 * j + c) * R + a;
          const int idx = lp * 4 * R + c * R + a;
          st_c(X2g + o, X2s[idx]);
          st_c(ETAg + o, Et[idx]);
        }
      }
    }
    if (!grid_sync(gb)) return;
    RTR_STAMP(stamp++);

    // ================= cost and gradient at X2, model decrease (k_rtr_eval2)
    if (wave == 0) {
      double fpart = 0, gpart = 0, ge = 0, eh = 0;
      double acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, vrow[4] = {0, 0, 0, 0}, hrow[4] = {0, 0, 0, 0}, eg[4] = {0, 0, 0, 0};
      if (rl) {
        auto ldE = [&](int i, double(*rw)[4]) {
#pragma unroll
          for (int cp = 0; cp < 4; ++cp) {
            const int o = (4 * i + cp) * R + a;
            rw[0][cp] = cX2.ld(o);
            rw[1][cp] = cETA.ld(o);
          }
        };
        double rawE[RTR_SLOTS][2][4];
        gather_issue<2>(idxL + lp * RTR_SLOTS, ldE, rawE);
        spmm_row_finish<R, 2, 2>(ag, j, BL + lp * RTR_SLOTS * 16, rawE, ldE,
          [&](double(*rw)[4], double(*x)[4]) {
#pragma unroll
            for (int cp = 0; cp < 4; ++cp) { x[0][cp] = rw[0][cp]; x[1][cp] = rw[1][cp]; }
          }, acc, ellw, tp0, tp1);
        const bool pub = ag.pub_index[j] >= 0;
        const double *G = ag.buf[B_G] + (size_t)j * 4 * R;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int idx = lp * 4 * R + c * R + a;
          const double xr = X2s[idx], g = pub ? G[c * R + a] : 0.0;
          fpart += (0.5 * acc[0][c] + g) * xr;
          eg[c] = acc[0][c] + g;
          E2s[idx] = eg[c];
          Ws[idx] = eg[c];
        }
      }
      WSYNC();
      if (rl) {
        double o3[3];
        tangent_row<R>(X2s + lp * 4 * R, Ws + lp * 4 * R, a, o3);
#pragma unroll
        for (int c = 0; c < 3; ++c) { G2s[lp * 4 * R + c * R + a] = o3[c]; gpart += o3[c] * o3[c]; }
        G2s[lp * 4 * R + 3 * R + a] = eg[3];
        gpart += eg[3] * eg[3];
        {
          double *GF2g = ag.buf[B_GF2] + (size_t)j * 4 * R;
#pragma unroll
          for (int c = 0; c < 3; ++c) st_c(GF2g + c * R + a, o3[c]);
          st_c(GF2g + 3 * R + a, eg[3]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) vrow[c] = Et[lp * 4 * R + c * R + a];
      }
      WSYNC();
      hess_tail_w<R>(Xs + lp * 4 * R, Hcs + min(lp, NP - 1) * 9, Ws + lp * 4 * R, a, acc[1], vrow, hrow, rl);
      if (rl) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          ge += Gs[lp * 4 * R + c * R + a] * vrow[c];
          eh += vrow[c] * hrow[c];
        }
      }
      fpart = wave_sum(fpart); gpart = wave_sum(gpart); ge = wave_sum(ge); eh = wave_sum(eh);
      if (lane == 0) {
        st_c(wsC + bx, fpart); st_c(wsC + RTR_WS_PITCH + bx, gpart);
        st_c(wsC + 2 * RTR_WS_PITCH + bx, ge); st_c(wsC + 3 * RTR_WS_PITCH + bx, eh);
      }
    }
    if (!grid_sync(gb)) return;
    RTR_STAMP(stamp++);

    // ================= acceptance test and radius update (k_rtr_accept; ROPTLIB SolversTR constants)
    {
      double sv4[4][CSUM_U], sc[4];
      csum_issue<4, CSUM_U>(cWS, 3, nblk, lane, sv4);
      csum_finish<4, CSUM_U>(sv4, nblk, lane, sc);
      const double f2 = sc[0], g2 = sc[1], ge = sc[2], eh = sc[3];
      const double rho = (S.f1 - f2) / (-ge - 0.5 * eh);
      const bool accept = rho > 0.1;
      if (rho > 0.75) {
        if (S.tcg_status == 1 || S.tcg_status == 2) S.Delta = fmin(2.0 * S.Delta, max_radius);
      } else if (rho < 0.25) {
        S.Delta = 0.25 * S.Delta;
      }
      if (accept) { S.f1 = f2; S.ngf = sqrt(g2); S.accepted += 1; }
      gf_fresh = accept ? 1 : 0;
      S.hv_count += 1;
      {
        const int took = S.tcg_j + 1;
        if (S.outer_it == 0) S.o0 = took; else if (S.outer_it == 1) S.o1 = took;
        else if (S.outer_it == 2) S.o2 = took; else if (S.outer_it == 3) S.o3 = took;
      }
      S.outer_it += 1;
      S.outer_done = (S.outer_it >= max_outer) || (S.ngf < tol);
      S.tcg_active = 0;
      S.need_init = 1;
      if (accept && rl) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const size_t o = ((size_t)4 * j + c) * R + a;
          const int idx = lp * 4 * R + c * R + a;
          Xs[idx] = X2s[idx]; Es[idx] = E2s[idx]; Gs[idx] = G2s[idx];
          ag.buf[B_X][o] = X2s[idx];
          ag.buf[B_EGRAD][o] = E2s[idx];
          st_c(GFg + o, G2s[idx]);
        }
      }
    }
    if (S.outer_done) break;
    RTR_STAMP(stamp++);
  }
  if (bx == 0 && tid == 0) {
    const RtrState T = to_record(S);
    ag.st[0] = T; ag.st[1] = T;
    bar[RB_EPOCH] = gb.epoch;
    if constexpr (TL) bar[RB_TLX_EPOCH] = gb.tlx_epoch;
    // running totals of this agent's solves: the host reads them (and the record) whenever it next synchronises
    cum[0] += 1ull; cum[1] += (unsigned long long)S.hv_count; cum[2] += (unsigned long long)S.pc_count;
    cum[3] += (unsigned long long)S.outer_count;
    // ... straight into pinned host memory (no copy kernels behind the solve): valid once the stream has drained
    *host_rec = T;
    for (int k = 0; k < 4; ++k) host_cum[k] = cum[k];
  }
  if (tail) {
    // (the accepted point's own rows went to B_X with plain stores from this workgroup's row lanes)
    __syncthreads();
    solve_tail<R>(agents, ag, team, tail, num_robots, restart_interval, bx, npose, tid, pj0, pj1, pj2);
  }
}

static size_t rtr_static_lds(int r, bool tl = false, int np = 2) {
  const size_t no = (size_t)4 * np * r;  // outputs of a slab product
  return sizeof(double) * ((tl ? 2 : 4) * no + 12 * no + (size_t)np * RTR_SLOTS * 16 + (size_t)np * 9) + sizeof(int) * np * RTR_SLOTS + 256;
}

// poses per workgroup of the dense one-launch solve: 2 (slab in LDS: up to 512 poses), 3 (7 of the 12 columns in LDS, 5 in
// registers: up to 640 poses, r <= 5), 0: not eligible
int rtr_fused_np(int r, int n, int num_cus) {
  if (n < 1) return 0;
  const int cus = std::min(num_cus, 256);
  if ((n + 1) / 2 <= cus && 4 * n <= 2048 && (size_t)64 * 4 * n + rtr_static_lds(r, false, 2) <= (size_t)160 * 1024) return 2;
  if (r <= 5 && (n + 2) / 3 <= cus && 4 * n <= 2560 && (size_t)7 * 32 * n + rtr_static_lds(r, false, 3) <= (size_t)160 * 1024) return 3;
  return 0;
}

size_t rtr_fused_lds_bytes(int r, int n) {
  return 4 * n <= 2048 ? (size_t)64 * 4 * n + rtr_static_lds(r, false, 2) : (size_t)7 * 32 * n + rtr_static_lds(r, false, 3);
}

bool rtr_fused_eligible(int r, int n, int num_cus) { return rtr_fused_np(r, n, num_cus) != 0; }

// the same for an agent with the two-level preconditioner: nwg workgroups (separator / subdomain parts of the ownership
// order, each padded to even), max_rows = the most rows (poses) any workgroup's slab meets before the exchange.  Two
// workgroups share a CU where their LDS allows it, so agents of up to ~1000 poses stay in one launch.
size_t rtr_fused_tl_lds_bytes(int r, int max_pre_poses, int ns) { return (size_t)128 * (2 * max_pre_poses + 2 * ns) + rtr_static_lds(r, true); }

// row pairs (rows before the exchange + separator rows) a workgroup's slab may hold for two workgroups to share a CU
int rtr_fused_tl_fit_pairs(int r) { return (int)(((size_t)80 * 1024 - rtr_static_lds(r, true)) / 128); }

bool rtr_fused_tl_eligible(int r, int nwg, int max_pre_poses, int ns, int num_cus, int max_lds) {
  if (nwg < 1 || nwg > RTR_WS_PITCH) return false;
  if (2 * max_pre_poses > TLS_NT * TLS_NPRE || 2 * ns > TLS_NT * TLS_NPOST) return false;
  const size_t lds = rtr_fused_tl_lds_bytes(r, max_pre_poses, ns);
  if (lds > (size_t)max_lds) return false;
  const int per_cu = (2 * lds <= (size_t)160 * 1024) ? 2 : 1;  // (128-thread workgroups: two per CU still run one wave per SIMD)
  return nwg <= num_cus * per_cu;
}

int launch_rtr_solve(const LaunchCtx &c, int ai, int n, unsigned long long *bar, double *ws, unsigned long long *cum, RtrState *host_rec, unsigned long long *host_cum,
                     int *err, double Delta0, double tol, int max_outer, int max_inner, double max_radius, int tail,
                     int num_robots, int restart_interval, int tl_nwg, size_t tl_dyn, int np) {
  // the two-level slab, 8 columns x N4 doubles, or the 7 columns of a 3-pose workgroup that live in LDS
  const size_t dyn = tl_nwg > 0 ? tl_dyn : (np == 3 ? (size_t)7 * 32 * n : (size_t)64 * 4 * n);
  hipError_t e = hipSuccess;
  DPGO_DISPATCH_R(c.r, {
    static bool configured = false;
    if (!configured) {
      e = hipFuncSetAttribute((const void *)k_rtr_solve<R, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)rtr_static_lds(R));
      if (e == hipSuccess)
        e = hipFuncSetAttribute((const void *)k_rtr_solve<R, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)rtr_static_lds(R, true));
      if constexpr (R <= 5) {
        if (e == hipSuccess)
          e = hipFuncSetAttribute((const void *)k_rtr_solve<R, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)rtr_static_lds(R, false, 3));
      }
      configured = (e == hipSuccess);
    }
    if (e == hipSuccess) {
      if (tl_nwg > 0)
        hipLaunchKernelGGL((k_rtr_solve<R, true>), dim3(tl_nwg), dim3(TLS_NT), dyn, c.stream, c.agents, ai, bar, ws, cum, host_rec, host_cum, err,
                           Delta0, tol, max_outer, max_inner, max_radius, c.team, tail, num_robots, restart_interval, c.host_agents[ai]);
      else if (np == 3) {
        if constexpr (R <= 5)
          hipLaunchKernelGGL((k_rtr_solve<R, false, 3>), dim3((n + 2) / 3), dim3(256), dyn, c.stream, c.agents, ai, bar, ws, cum, host_rec, host_cum, err,
                             Delta0, tol, max_outer, max_inner, max_radius, c.team, tail, num_robots, restart_interval, c.host_agents[ai]);
        else e = hipErrorInvalidValue;
      } else
        hipLaunchKernelGGL((k_rtr_solve<R, false>), dim3((n + 1) / 2), dim3(256), dyn, c.stream, c.agents, ai, bar, ws, cum, host_rec, host_cum, err,
                           Delta0, tol, max_outer, max_inner, max_radius, c.team, tail, num_robots, restart_interval, c.host_agents[ai]);
    }
  });
  return e == hipSuccess ? 0 : -1;
}

}  // namespace dpgo
