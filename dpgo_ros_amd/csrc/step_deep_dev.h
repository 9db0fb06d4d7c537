// step_deep_dev.h -- device helpers shared by the deep-carried one-launch iteration (step_deep.hip, k_step_fd) and its
// persistent form (step_persist.hip, k_step_pd): hand-offs between the waves of a workgroup through counters in LDS, the
// chunk-ordered layout, the staging of the shared edges' operands, the product over the last chunks, the Nesterov scalars.
#pragma once
#include "kernel_common.h"

namespace dpgo {

constexpr int FD_KC = 2048;
// build switches of the experiments recorded in profiles/r06_deep_carry.md (every one measured SLOWER than the defaults but
// the last): DPGO_FD_HEAD -- 16-byte loads per lane of the next agent's slab requested in front of the request hand-off;
// DPGO_FD_GC_LATE -- the carried rows of the current agent requested behind it; DPGO_FD_PACC_WT -- partial sums stored
// write-through; DPGO_FD_E_EARLY -- wave 7 hands the coefficients over before it requests its look-ahead operands;
// DPGO_FD_KA_PREFETCH (default on) -- one word of every line of the kernel arguments touched up front
#ifndef DPGO_FD_HEAD
#define DPGO_FD_HEAD 0
#endif
#ifndef DPGO_FD_GC_LATE
#define DPGO_FD_GC_LATE 0
#endif
#ifndef DPGO_FD_PACC_WT
#define DPGO_FD_PACC_WT 0
#endif
#ifndef DPGO_FD_E_EARLY
#define DPGO_FD_E_EARLY 0
#endif
#ifndef DPGO_FD_KA_PREFETCH
#define DPGO_FD_KA_PREFETCH 1
#endif
constexpr int FD_HEAD = DPGO_FD_HEAD;  // 16-byte loads per lane of the NEXT agent's private chunks requested in front of the request hand-off

// hand-offs between the waves of the workgroup: counters in LDS (see the head of the file)
enum { FD_SY_C = 0, FD_SY_E, FD_SY_D, FD_SY_F, FD_SY_N, FD_SY_RQ, FD_SY_COUNT = 8 };

__device__ __forceinline__ void fd_signal(int *cnt) {
  // (the LDS operations of one wave execute in order: the count follows the wave's writes)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// "everything this wave needs is in the CU's memory queue": counted behind the wave's last request (the compiler may move
// neither the requests below it nor the count above them)
__device__ __forceinline__ void fd_signal_requested(int *cnt) {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::: "memory");
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void fd_wait(int *cnt, int target) {
  // (bounded: the waves of a workgroup are resident together, so the count always arrives -- the bound only keeps a logic
  // error from hanging the device; 4M polls are a fraction of a second)
  for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target; ++spin)
    __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// where pose p of an agent sits in the chunk-ordered ("position") layout of its vector: chunk p / 16 moves to the place
// it has in the agent's chunk order, [place][16 poses][4r]
template <int R>
__device__ __forceinline__ int fd_pos_off(const unsigned char *ord, int pose) {
  const int ch = pose >> 4;
  int p = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) p = ((int)ord[i] == ch) ? i : p;
  return p * 64 * R + (pose & 15) * 4 * R;
}

// The operands of the shared edges go to LDS, [edge][neighbour pose 4r | 16 coefficients].
//   * wave 6, neighbour poses: one lane per edge (edges l, 64 + l, 128 + l), 2r loads of 16 bytes each.  Where the neighbour
//     lives comes from the edge's 16-bit code (frame | agent << 12) in the descriptor -- scalar registers and a select
//     chain, no descriptor round trip (k_step_fe).  (Measured and dropped, round 6: 64 consecutive 16-byte parts per load --
//     a quarter of the cache-line requests -- with the address handed from the edge's lane by ds_bpermute: the wave reached
//     the request hand-off 1.5 us LATER; and with the codes looked up per part: scalar loads inside every trip, 4 us later.)
//   * wave 7, coefficients: the packed copy [edge][16] (AgentDev::fe_coef) read straight through, 1 KB per load -- 100
//     cache-line requests where one lane per edge asked for 800 (the kernel's first microseconds are a count of such
//     requests: ~4400 per CU in front of the stream, and a CU's texture path takes about one a cycle).
template <int R>
struct FdXn {
  double2 v[3][2 * R];
};

template <int R>
__device__ __forceinline__ void fd_xn_request(const AgentDev &ag, const FeBases &fb, int parity, int ln, FdXn<R> &xr) {
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    {  // (every slot, edges or not: loads under a wave-uniform `if` leave the compiler without a count of what is in flight
       // behind the join, and the wait in front of the LDS writes becomes a wait for EVERYTHING the wave has requested)
      unsigned wsel = 0;
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const unsigned wk = ag.fe_code[(32 * q + k < FE_MAX_EDGES / 2) ? 32 * q + k : FE_MAX_EDGES / 2 - 1];
        wsel = ((ln >> 1) == k) ? wk : wsel;
      }
      // (lanes beyond the last edge read the pose of an edge that exists: words beyond the last edge hold zeros = agent 0, frame 0)
      const unsigned code = (ln & 1) ? (wsel >> 16) : (wsel & 0xffffu);
      const int sa = (int)(code >> 12), sf = (int)(code & 0xfffu);
      const double *yb = fb.ybase[0];
      int yn = fb.npose[0];
#pragma unroll
      for (int k = 1; k < LOOKAHEAD_MAX_AGENTS; ++k) { yb = (sa == k) ? fb.ybase[k] : yb; yn = (sa == k) ? fb.npose[k] : yn; }
      const double *xp = yb + (parity ? (size_t)B_ALT * 4 * R * yn : (size_t)0) + (size_t)sf * 4 * R;
#pragma unroll
      for (int k = 0; k < 2 * R; ++k) xr.v[q][k] = ld2(xp + 2 * k);
    }
  }
}

template <int R>
__device__ __forceinline__ void fd_xn_to_lds(const AgentDev &ag, int ln, const FdXn<R> &xr, double *Es) {
  constexpr int EPE = 4 * R + 16;
  int nsh = ag.nshared;
  asm volatile("" : "+v"(nsh));  // (see fd_cf_to_lds)
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (64 * q < nsh && 64 * q + ln < nsh) {
      double *E = Es + (size_t)(64 * q + ln) * EPE;
#pragma unroll
      for (int k = 0; k < 2 * R; ++k) *reinterpret_cast<double2 *>(E + 2 * k) = xr.v[q][k];
    }
  }
}

struct FdCf {
  static constexpr int TRIPS = FE_MAX_EDGES * 8 / 64;
  double2 v[TRIPS];
};

__device__ __forceinline__ void fd_cf_request(const AgentDev &ag, int ln, FdCf &cr) {
  const int nit = ag.nshared * 8;
#pragma unroll
  for (int j = 0; j < FdCf::TRIPS; ++j) cr.v[j] = ld2(ag.fe_coef + 2 * min(64 * j + ln, nit - 1));  // (every trip: see fd_xn_request)
}

template <int R>
__device__ __forceinline__ void fd_cf_to_lds(const AgentDev &ag, int ln, const FdCf &cr, double *Es) {
  constexpr int EPE = 4 * R + 16;
  // (held in a register: left as an expression of the descriptor the count was RE-READ from the kernel arguments in front of
  // every one of the 20 stores below, each time behind a wait for the scalar load and the LDS store before it -- 0.9 us
  // between this wave's last request and the hand-off the chain waits for; read off the ISA, round 6)
  int nit = ag.nshared * 8;
  asm volatile("" : "+v"(nit));
#pragma unroll
  for (int j = 0; j < FdCf::TRIPS; ++j) {
    const int t = 64 * j + ln;
    if (64 * j < nit && t < nit) *reinterpret_cast<double2 *>(Es + (size_t)(t >> 3) * EPE + 4 * R + 2 * (t & 7)) = cr.v[j];
  }
}

// Nesterov scalars of the iterations this launch looks at, from the state BEFORE iteration k (the recurrences of
// k_step_fe, one step further): every wave that needs them derives them from the same NestState
struct FdNest {
  bool restart_now, restart_next, restart_next2, restart_next3;
  double nest_gamma, ahead_alpha, ahead2_alpha, ahead3_alpha;
};

__device__ __forceinline__ FdNest fd_nest(const NestState &ns, int num_robots, int restart_interval) {
  FdNest o;
  const double Nr = (double)num_robots;
  o.restart_now = ((ns.iter + 2) % restart_interval) == 0;
  o.restart_next = ((ns.iter + 3) % restart_interval) == 0;
  o.restart_next2 = ((ns.iter + 4) % restart_interval) == 0;
  o.restart_next3 = ((ns.iter + 5) % restart_interval) == 0;
  o.nest_gamma = o.restart_now ? 0.0 : (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * ns.gamma * ns.gamma)) / (2.0 * Nr);
  const double g2 = (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * o.nest_gamma * o.nest_gamma)) / (2.0 * Nr);
  o.ahead_alpha = 1.0 / (g2 * Nr);
  const double gamma_next = o.restart_next ? 0.0 : g2;
  const double g3 = (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * gamma_next * gamma_next)) / (2.0 * Nr);
  o.ahead2_alpha = 1.0 / (g3 * Nr);
  const double gamma_next2 = o.restart_next2 ? 0.0 : g3;
  const double g4 = (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * gamma_next2 * gamma_next2)) / (2.0 * Nr);
  o.ahead3_alpha = 1.0 / (g4 * Nr);
  return o;
}

// the last NC chunks of one column per lane of waves 4-7 (lane l of the four: column group (l >> 5) & 7, k-lane l & 31 -- the
// mapping of k_step_fe's stream waves) and the partial sums the previous launch left for it: requested in front of barrier
// A; behind hand-off D the SAME chain of fused multiply-adds continues over the rows that are in LDS by then
template <int R, int NC>
struct FdCur {
  double2 mc[NC];
  double pa[R];
};

template <int R, int M0, int NC>
__device__ __forceinline__ void fd_cur_request(const AgentDev &ag, const double *pacc_in, int bx, int nblk, int l, FdCur<R, NC> &cu) {
  const int cg = (l >> 5) & 7, kl = l & 31;
  const int N4 = ag.N4, col = 8 * bx + cg;
  const double *Mc = ag.M + (size_t)((col < N4) ? col : 0) * N4;
#pragma unroll
  for (int i = 0; i < NC; ++i) cu.mc[i] = ld2_nt(Mc + min(2 * kl + 64 * (int)ag.fe_ord[M0 + i], N4 - 2));
#pragma unroll
  for (int a = 0; a < R; ++a) cu.pa[a] = gp(pacc_in)[((size_t)min(bx, nblk - 1) * R + a) * 256 + l];
}

template <int R, int M0, int NC>
__device__ __forceinline__ void fd_cur_product(const FdCur<R, NC> &cu, const double *vs, double *red, int l) {
  const int cg = (l >> 5) & 7, kl = l & 31;
  double acc[R];
#pragma unroll
  for (int a = 0; a < R; ++a) acc[a] = cu.pa[a];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int k = 2 * kl + 64 * (M0 + i);
    double wv[2 * R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const double2 t2 = *reinterpret_cast<const double2 *>(&vs[k * R + 2 * q]);
      wv[2 * q] = t2.x; wv[2 * q + 1] = t2.y;
    }
#pragma unroll
    for (int a = 0; a < R; ++a) acc[a] = __builtin_fma(wv[R + a], cu.mc[i].y, __builtin_fma(wv[a], cu.mc[i].x, acc[a]));
  }
#pragma unroll
  for (int a = 0; a < R; ++a) red[kl * (8 * R + 1) + cg * R + a] = acc[a];
}

}  // namespace dpgo
