// twolevel_dev.h -- device side of the two-level preconditioner apply (twolevel.h):
//   z[:, own columns] = v_pre * slab_pre + u * slab_post,      u = v_S - sum_i v_i E_i
// for the 8 columns (2 poses) a workgroup owns.  Lane t takes row pair t (+256 m) of its workgroup's slab for ALL 8
// columns -- 128 contiguous bytes per lane, stored in exactly that order by the set-up -- and the 2R matching entries of
// the input vector (gathered through the workgroup's row-pose list) or of u.  The first nA workgroups of the launch own
// one separator pose each: they form its entry of u from the adjacent subdomains' rows, publish it (write-through stores,
// one counter increment per workgroup) and leave.  Everybody else multiplies its subdomain's rows by D_i while it waits
// for that counter, then adds the u rows (the separator poses' own columns of Sc^-1 belong to the workgroups right
// behind the producers).  Producers are the FIRST workgroups of the grid and wait for NOBODY, so the exchange cannot
// deadlock however many workgroups are resident at once (a grid larger than the device simply finds the counter
// complete); the last workgroup past the exchange clears the counters for the next launch.
#pragma once
#include "kernel_common.h"

namespace dpgo {

constexpr int TL_FLAG_PUB = 0 /* 8 copies, 16 words apart */, TL_FLAG_DONE = 128, TL_FLAG_EPOCH = 144;
constexpr int TL_FLAG_WORDS = 192;
constexpr int TL_SPIN_LIMIT = 1 << 22;

template <int R>
__device__ __forceinline__ void tl_zero(double (*acc)[R]) {
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int a = 0; a < R; ++a) acc[c][a] = 0.0;
}

// acc[c][a] += w[a] * mm[c].x + w[R + a] * mm[c].y   (w: the two vector rows of the pair, scaled by `live`)
template <int R>
__device__ __forceinline__ void tl_fma(double (*acc)[R], const double2 *v, const double2 *mm, double live) {
  double w[2 * R];
#pragma unroll
  for (int q = 0; q < R; ++q) { w[2 * q] = v[q].x * live; w[2 * q + 1] = v[q].y * live; }
  // (two fused multiply-adds per accumulator: the one-expression form is mul, fma, add through one temporary, a
  // dependent chain per accumulator)
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int a = 0; a < R; ++a) acc[c][a] = __builtin_fma(w[a], mm[c].x, acc[c][a]);
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int a = 0; a < R; ++a) acc[c][a] = __builtin_fma(w[R + a], mm[c].y, acc[c][a]);
}

// the 256 per-lane sums of the 8R outputs -> zs[c * R + a], valid in every wave on return.  red: TL_RED_DOUBLES(R)
// doubles of LDS.  Quad reduction (DPP), 64 LDS rows of odd pitch, one lane per output adds them in row order.  (256
// rows without the DPP stage measured the same 2 us and cost 84 KB of LDS, i.e. one workgroup per CU.)
#define TL_RED_DOUBLES(R) (64 * (8 * (R) + 1))
template <int R>
__device__ __forceinline__ void tl_reduce(double (*acc)[R], double *red, double *zs, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int a = 0; a < R; ++a) {
      double x = acc[c][a];
      x += dpp_move<0xB1>(x);  // lanes ^ 1
      x += dpp_move<0x4E>(x);  // lanes ^ 2
      acc[c][a] = x;
    }
  if ((lane & 3) == 0) {
    double *row = red + (size_t)(wave * 16 + (lane >> 2)) * (8 * R + 1);
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int a = 0; a < R; ++a) row[c * R + a] = acc[c][a];
  }
  __syncthreads();
  if (tid < 8 * R) {
    double t[64];
#pragma unroll
    for (int q = 0; q < 64; ++q) t[q] = red[q * (8 * R + 1) + tid];
    double s = 0;
#pragma unroll
    for (int q = 0; q < 64; ++q) s += t[q];
    zs[tid] = s;
  }
  __syncthreads();
}

// first pass over the rows that meet the input vector: requested as early as the workgroup knows its slab, consumed by
// tl_apply (the step kernel's own prologue runs under these loads)
template <int R>
struct TLPre {
  double2 v[R], mm[8];
  double live;
};
template <int R>
__device__ __forceinline__ void tl_issue(const TLDev &tl, const TLWg &w, int b, const double *__restrict__ V, int tid, TLPre<R> &pre) {
  const int *rp = tl.rowpose + (size_t)b * tl.rp_stride;
  const double *slab = tl.slabs + w.slab_off;
  const int npre = 2 * w.pre_cnt;
  const int qq = max(min(tid, npre - 1), 0);
  const int pose = rp[qq >> 1];
  const double *vp = V + ((size_t)4 * pose + 2 * (qq & 1)) * R;
#pragma unroll
  for (int a = 0; a < R; ++a) pre.v[a] = ld2(vp + 2 * a);
#pragma unroll
  for (int c = 0; c < 8; ++c) pre.mm[c] = ld2_nt(slab + (size_t)qq * 16 + 2 * c);
  pre.live = tid < npre ? 1.0 : 0.0;
}

// One apply for workgroup b of the launch (256 threads).  V: the input vector (r x 4n, written by an EARLIER launch).
// Returns false for a producer (it has published its entry of u and is done: no columns of its own); otherwise
// zs[(4 lp + c) * R + a] holds column c of own pose lp on return (all waves may read it).
// trace (TRACE builds of the step kernel): per-wave wall-clock stamps [wave][16], slots 1..5 written here
template <int R, bool TRACE = false>
__device__ __forceinline__ bool tl_apply(const TLDev &tl, const TLWg &w, int b, const double *__restrict__ V, const TLPre<R> &pre,
                                         double *red, double *zs, int tid, double *trace = nullptr) {
#define TL_STAMP(k) do { if (TRACE && trace && (tid & 63) == 0) trace[(tid >> 6) * 16 + (k)] = (double)wall_clock64(); } while (0)
  const int *rp = tl.rowpose + (size_t)b * tl.rp_stride;
  const double *slab = tl.slabs + w.slab_off;
  const int npre = 2 * w.pre_cnt, npost = 2 * tl.ns;
  const bool producer = b < tl.nA;
  // a producer adds its own separator rows of the input vector to what the subdomains contribute: requested now
  double vown = 0;
  if (producer && tid < 4 * R && w.own[0] >= 0) vown = V[(size_t)w.own[0] * 4 * R + tid];
  const double *post = slab + (size_t)npre * 16;
  // the first pass of the separator rows is requested before the wait as well
  double2 mmP[8];
  if (!producer) {
    const int qq = max(min(tid, npost - 1), 0);
#pragma unroll
    for (int c = 0; c < 8; ++c) mmP[c] = ld2_nt(post + (size_t)qq * 16 + 2 * c);
  }
  double acc[8][R];
  tl_zero<R>(acc);
  // ---- rows that meet the input vector itself
  tl_fma<R>(acc, pre.v, pre.mm, pre.live);
  for (int q0 = 256; q0 < npre; q0 += 256) {
    const int q = q0 + tid, qq = min(q, npre - 1);
    const int pose = rp[qq >> 1];
    const double *vp = V + ((size_t)4 * pose + 2 * (qq & 1)) * R;
    double2 v[R], mm[8];
#pragma unroll
    for (int a = 0; a < R; ++a) v[a] = ld2(vp + 2 * a);
#pragma unroll
    for (int c = 0; c < 8; ++c) mm[c] = ld2_nt(slab + (size_t)qq * 16 + 2 * c);
    tl_fma<R>(acc, v, mm, q < npre ? 1.0 : 0.0);
  }
  TL_STAMP(1);
  if (npost == 0) {  // no separator: the operator is block diagonal
    tl_reduce<R>(acc, red, zs, tid);
    return true;
  }
  unsigned long long departed = 0;
  if (producer) {
    // phase A: u of the own separator pose = v + (what the adjacent subdomains contribute), published for everybody:
    // write-through stores, then one increment of every XCD's copy of the counter.  Producers wait for nobody.
    tl_reduce<R>(acc, red, zs, tid);
    if (tid < 4 * R && w.own[0] >= 0) st_c(tl.u + ((size_t)w.sep0 * 4 * R + tid), vown + zs[tid]);
    if (tid < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores have left the CU
    __syncthreads();
    if (tid < 8) __hip_atomic_fetch_add(&tl.flag[TL_FLAG_PUB + 16 * tid], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    TL_STAMP(2);
    if (tid == 255) departed = __hip_atomic_fetch_add(&tl.flag[TL_FLAG_DONE], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    // ---- the exchange: every producer has published (the workgroups of an XCD poll that XCD's copy of the counter)
    if (tid == 0) {
      const unsigned long long *f = &tl.flag[TL_FLAG_PUB + 16 * ((int)blockIdx.x & 7)];
      int spins = 0;
      while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)tl.nA) {
        if (++spins > TL_SPIN_LIMIT) { *tl.err = 3; break; }
      }
    }
    __syncthreads();
    TL_STAMP(3);
    if (tid == 255) departed = __hip_atomic_fetch_add(&tl.flag[TL_FLAG_DONE], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ---- separator rows: u.  Written once per launch, by write-through stores, before the counter moved, and never
    // read by anybody in this launch before that: ordinary loads (the L2 of this XCD fetches each line once for all its
    // workgroups; caches start a launch invalidated)
    {
      const int qq = min(tid, npost - 1);
      double2 v[R];
#pragma unroll
      for (int a = 0; a < R; ++a) v[a] = ld2(tl.u + (size_t)qq * 2 * R + 2 * a);
      tl_fma<R>(acc, v, mmP, tid < npost ? 1.0 : 0.0);
    }
    for (int q0 = 256; q0 < npost; q0 += 256) {
      const int q = q0 + tid, qq = min(q, npost - 1);
      double2 v[R], mm[8];
#pragma unroll
      for (int a = 0; a < R; ++a) v[a] = ld2(tl.u + (size_t)qq * 2 * R + 2 * a);
#pragma unroll
      for (int c = 0; c < 8; ++c) mm[c] = ld2_nt(post + (size_t)qq * 16 + 2 * c);
      tl_fma<R>(acc, v, mm, q < npost ? 1.0 : 0.0);
    }
    TL_STAMP(4);
    tl_reduce<R>(acc, red, zs, tid);
    TL_STAMP(5);
  }
  // the last workgroup past the exchange clears the counters for the next launch on this agent
  if (tid == 255 && departed + 1ull == (unsigned long long)tl.nwg) {
#pragma unroll
    for (int x = 0; x < 8; ++x) __hip_atomic_store(&tl.flag[TL_FLAG_PUB + 16 * x], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&tl.flag[TL_FLAG_DONE], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#undef TL_STAMP
  return !producer;
}

}  // namespace dpgo
