// twolevel_dev.h -- device side of the two-level preconditioner apply (twolevel.h):
//   z[:, own columns] = v_pre * slab_pre + u * slab_post,      u = v_S - sum_i v_i E_i
// for the 8 columns (2 poses) a workgroup owns.  Lane t takes row pair t (+256 m) of its workgroup's slab for ALL 8
// columns -- 128 contiguous bytes per lane, stored in exactly that order by the set-up -- and the 2R matching entries of
// the input vector (gathered through the workgroup's row-pose list) or of u.  The first nA workgroups of the launch own
// the separator poses: they form u for their poses from the adjacent subdomains' rows, publish it (write-through stores,
// one counter increment per workgroup) and only then turn to their own columns of Sc^-1.  Everybody else multiplies its
// subdomain's rows by D_i while it waits for that counter, then adds the u rows.  Producers are the FIRST workgroups of
// the grid and never wait for anybody, so the exchange cannot deadlock however many workgroups are resident at once
// (a grid larger than the device simply finds the counter complete); the last workgroup past the exchange clears the
// counters for the next launch.
#pragma once
#include "kernel_common.h"

namespace dpgo {

constexpr int TL_FLAG_PUB = 0, TL_FLAG_DONE = 16, TL_FLAG_EPOCH = 32;
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
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int a = 0; a < R; ++a) acc[c][a] += w[a] * mm[c].x + w[R + a] * mm[c].y;
}

// the 256 per-lane sums of the 8R outputs -> zs[c * R + a], valid in wave 0 on return.  red: 64 x (8R + 1) doubles.
// Quad reduction (DPP), 64 LDS rows of odd pitch, one lane per output adds them in row order.
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
  if (tid < 64) WSYNC();
}

// One apply for workgroup b of the launch (256 threads).  V: the input vector (r x 4n, written by an EARLIER launch).
// On return zs[(4 lp + c) * R + a] holds column c of own pose lp, valid in wave 0.
template <int R>
__device__ __forceinline__ void tl_apply(const TLDev &tl, const TLWg &w, int b, const double *__restrict__ V, double *red,
                                         double *zs, int tid) {
  const int *rp = tl.rowpose + (size_t)b * tl.rp_stride;
  const double *slab = tl.slabs + w.slab_off;
  const int npre = 2 * w.pre_cnt, npost = 2 * tl.ns;
  double acc[8][R];
  tl_zero<R>(acc);
  // ---- rows that meet the input vector itself
  for (int q0 = 0; q0 < npre; q0 += 256) {
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
  if (npost == 0) {  // no separator: the operator is block diagonal
    tl_reduce<R>(acc, red, zs, tid);
    return;
  }
  const double *post = slab + (size_t)npre * 16;
  // the first pass of the separator rows is requested before the wait
  double2 mmP[8];
  {
    const int qq = min(tid, npost - 1);
#pragma unroll
    for (int c = 0; c < 8; ++c) mmP[c] = ld2_nt(post + (size_t)qq * 16 + 2 * c);
  }
  if (b < tl.nA) {
    // phase A: u of the own separator poses = v + (what the adjacent subdomains contribute), published for everybody
    tl_reduce<R>(acc, red, zs, tid);
    if (tid < 8 * R) {
      const int lp = tid / (4 * R), e = tid - lp * 4 * R, own = w.own[lp];
      if (own >= 0) st_c(tl.u + ((size_t)(2 * b + lp) * 4 * R + e), V[(size_t)own * 4 * R + e] + zs[tid]);
    }
    if (tid < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores have left the CU
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(&tl.flag[TL_FLAG_PUB], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tl_zero<R>(acc);
  }
  // ---- the exchange: every producer has published
  if (tid == 0) {
    int spins = 0;
    while (__hip_atomic_load(&tl.flag[TL_FLAG_PUB], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)tl.nA) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > TL_SPIN_LIMIT) { *tl.err = 3; break; }
    }
  }
  __syncthreads();
  unsigned long long departed = 0;
  if (tid == 255) departed = __hip_atomic_fetch_add(&tl.flag[TL_FLAG_DONE], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // ---- separator rows: u (published by other workgroups of this launch: L1-bypassing loads)
  const CVec cu(tl.u, 4 * tl.ns * R);
  {
    const int qq = min(tid, npost - 1);
    double2 v[R];
#pragma unroll
    for (int a = 0; a < R; ++a) v[a] = cu.ld2(qq * 2 * R + 2 * a);
    tl_fma<R>(acc, v, mmP, tid < npost ? 1.0 : 0.0);
  }
  for (int q0 = 256; q0 < npost; q0 += 256) {
    const int q = q0 + tid, qq = min(q, npost - 1);
    double2 v[R], mm[8];
#pragma unroll
    for (int a = 0; a < R; ++a) v[a] = cu.ld2(qq * 2 * R + 2 * a);
#pragma unroll
    for (int c = 0; c < 8; ++c) mm[c] = ld2_nt(post + (size_t)qq * 16 + 2 * c);
    tl_fma<R>(acc, v, mm, q < npost ? 1.0 : 0.0);
  }
  tl_reduce<R>(acc, red, zs, tid);
  // the last workgroup past the exchange clears the counters for the next launch on this agent
  if (tid == 255 && departed + 1ull == (unsigned long long)tl.nwg) {
    __hip_atomic_store(&tl.flag[TL_FLAG_PUB], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&tl.flag[TL_FLAG_DONE], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace dpgo
