// standalone microbenchmark of the dense preconditioner apply decomposition (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef double v2d_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld2_nt(const double *p) { const v2d_t v = __builtin_nontemporal_load((const v2d_t*)p); return make_double2(v.x, v.y); }
__device__ __forceinline__ double2 ld2(const double *p) { return *(const double2*)p; }
constexpr int R = 5;

// V0: stream only. block = 8 columns x 32 k-lanes, 32 double2 per thread, sum and write one value
template <int NT>
__global__ __launch_bounds__(256) void k_stream(const double* M, double* out, int N4) {
  const int tid = threadIdx.x, cg = tid >> 5, kl = tid & 31;
  const int col = 8 * blockIdx.x + cg;
  const double* Mc = M + (size_t)col * N4;
  double2 m[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) { const int k = 2 * kl + 64 * q; m[q] = (k < N4) ? (NT ? ld2_nt(Mc + k) : ld2(Mc + k)) : make_double2(0, 0); }
  double s = 0;
#pragma unroll
  for (int q = 0; q < 32; ++q) s += m[q].x + m[q].y;
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (kl == 0) out[col] = s;
}

// V1: stream + LDS-staged vector (the product's structure, PLAIN mode without epilogue)
constexpr int KC = 2048, KCP = KC + 4;
__global__ __launch_bounds__(256) void k_gemv(const double* M, const double* V, double* out, int N4) {
  __shared__ double vs[R * KCP];
  const int tid = threadIdx.x, cg = tid >> 5, kl = tid & 31;
  const int col = 8 * blockIdx.x + cg;
  const double* Mc = M + (size_t)col * N4;
  double2 m[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) { const int k = 2 * kl + 64 * q; m[q] = (k < N4) ? ld2_nt(Mc + k) : make_double2(0, 0); }
  for (int t0 = tid; t0 < KC * R; t0 += 256 * 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int tt = t0 + 256 * u; v[u] = (tt < N4 * R) ? V[tt] : 0.0; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int tt = t0 + 256 * u; if (tt < KC * R) { const int k = tt / R, a = tt - k * R; vs[a * KCP + k] = v[u]; } }
  }
  __syncthreads();
  double acc[R];
#pragma unroll
  for (int a = 0; a < R; ++a) acc[a] = 0;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int k = 2 * kl + 64 * q;
#pragma unroll
    for (int a = 0; a < R; ++a) { const double2 v = *(const double2*)&vs[a * KCP + k]; acc[a] += v.x * m[q].x + v.y * m[q].y; }
  }
#pragma unroll
  for (int a = 0; a < R; ++a) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc[a] += __shfl_xor(acc[a], off, 64);
  }
  if (kl == 0) {
#pragma unroll
    for (int a = 0; a < R; ++a) out[(size_t)col * R + a] = acc[a];
  }
}

// V2: lanes along k, every thread handles all 8 columns of the block for its own k's; v straight from global
// (no LDS staging, no redundancy inside the block); 40 accumulators reduced through LDS at the end.
__global__ __launch_bounds__(256) void k_gemv2(const double* M, const double* V, double* out, int N4) {
  __shared__ double red[4][8 * R];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int col0 = 8 * blockIdx.x;
  double acc[8][R];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int a = 0; a < R; ++a) acc[c][a] = 0;
  // k pairs: thread handles k = 2*tid + 512*q, q < 4  (N4 <= 2048)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = 2 * tid + 512 * q;
    double2 m[8];
    double v0[R], v1[R];
    const bool ok = k < N4;
#pragma unroll
    for (int c = 0; c < 8; ++c) m[c] = ok ? ld2_nt(M + (size_t)(col0 + c) * N4 + k) : make_double2(0, 0);
#pragma unroll
    for (int a = 0; a < R; ++a) { v0[a] = ok ? V[(size_t)k * R + a] : 0.0; v1[a] = ok ? V[(size_t)(k + 1) * R + a] : 0.0; }
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int a = 0; a < R; ++a) acc[c][a] += v0[a] * m[c].x + v1[a] * m[c].y;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int a = 0; a < R; ++a) {
      double s = acc[c][a];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      if (lane == 0) red[w][c * R + a] = s;
    }
  __syncthreads();
  if (tid < 8 * R) out[(size_t)col0 * R + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// V3: v first (one batch of 16-byte loads, transposed into [a][k]), then the M slab, then compute
__global__ __launch_bounds__(256) void k_gemv3(const double* M, const double* V, double* out, int N4) {
  __shared__ double vs[R * KCP];
  const int tid = threadIdx.x, cg = tid >> 5, kl = tid & 31;
  const int col = 8 * blockIdx.x + cg;
  const double* Mc = M + (size_t)col * N4;
  double2 v[20];
#pragma unroll
  for (int u = 0; u < 20; ++u) { const int tt = 2 * (tid + 256 * u); v[u] = (tt < N4 * R) ? ld2(V + tt) : make_double2(0, 0); }
  double2 m[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) { const int k = 2 * kl + 64 * q; m[q] = (k < N4) ? ld2_nt(Mc + k) : make_double2(0, 0); }
#pragma unroll
  for (int u = 0; u < 20; ++u) {
    const int tt = 2 * (tid + 256 * u);
    if (tt < KC * R) { int k = tt / R, a = tt - k * R; vs[a * KCP + k] = v[u].x; k = (tt + 1) / R; a = tt + 1 - k * R; vs[a * KCP + k] = v[u].y; }
  }
  __syncthreads();
  double acc[R];
#pragma unroll
  for (int a = 0; a < R; ++a) acc[a] = 0;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int k = 2 * kl + 64 * q;
#pragma unroll
    for (int a = 0; a < R; ++a) { const double2 w = *(const double2*)&vs[a * KCP + k]; acc[a] += w.x * m[q].x + w.y * m[q].y; }
  }
#pragma unroll
  for (int a = 0; a < R; ++a) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc[a] += __shfl_xor(acc[a], off, 64);
  }
  if (kl == 0) {
#pragma unroll
    for (int a = 0; a < R; ++a) out[(size_t)col * R + a] = acc[a];
  }
}

// V7: v first, straight copy into LDS in its native [k][a] layout (no transposition), then M, then compute:
// per q a thread reads the 10 contiguous doubles v[k][0..4], v[k+1][0..4] as 5 x 16 B (80-byte lane stride:
// conflict-free for ds_read_b128)
__global__ __launch_bounds__(256) void k_gemv7(const double* M, const double* V, double* out, int N4) {
  __shared__ double vs[R * KC];
  const int tid = threadIdx.x, cg = tid >> 5, kl = tid & 31;
  const int col = 8 * blockIdx.x + cg;
  const double* Mc = M + (size_t)col * N4;
  double2 v[20];
#pragma unroll
  for (int u = 0; u < 20; ++u) { const int tt = 2 * (tid + 256 * u); v[u] = (tt < N4 * R) ? ld2(V + tt) : make_double2(0, 0); }
  double2 m[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) { const int k = 2 * kl + 64 * q; m[q] = (k < N4) ? ld2_nt(Mc + k) : make_double2(0, 0); }
#pragma unroll
  for (int u = 0; u < 20; ++u) { const int tt = 2 * (tid + 256 * u); *(double2*)&vs[tt] = v[u]; }
  __syncthreads();
  double acc[R];
#pragma unroll
  for (int a = 0; a < R; ++a) acc[a] = 0;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int k = 2 * kl + 64 * q;
    double w[2 * R];
#pragma unroll
    for (int j = 0; j < R; ++j) { const double2 t = *(const double2*)&vs[k * R + 2 * j]; w[2 * j] = t.x; w[2 * j + 1] = t.y; }
#pragma unroll
    for (int a = 0; a < R; ++a) acc[a] += w[a] * m[q].x + w[R + a] * m[q].y;
  }
#pragma unroll
  for (int a = 0; a < R; ++a) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc[a] += __shfl_xor(acc[a], off, 64);
  }
  if (kl == 0) {
#pragma unroll
    for (int a = 0; a < R; ++a) out[(size_t)col * R + a] = acc[a];
  }
}

// V8: as V7 but M first then v (all 52 loads issued back to back, no intermediate wait)
__global__ __launch_bounds__(256) void k_gemv8(const double* M, const double* V, double* out, int N4) {
  __shared__ double vs[R * KC];
  const int tid = threadIdx.x, cg = tid >> 5, kl = tid & 31;
  const int col = 8 * blockIdx.x + cg;
  const double* Mc = M + (size_t)col * N4;
  double2 m[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) { const int k = 2 * kl + 64 * q; m[q] = (k < N4) ? ld2_nt(Mc + k) : make_double2(0, 0); }
  double2 v[20];
#pragma unroll
  for (int u = 0; u < 20; ++u) { const int tt = 2 * (tid + 256 * u); v[u] = (tt < N4 * R) ? ld2(V + tt) : make_double2(0, 0); }
#pragma unroll
  for (int u = 0; u < 20; ++u) { const int tt = 2 * (tid + 256 * u); *(double2*)&vs[tt] = v[u]; }
  __syncthreads();
  double acc[R];
#pragma unroll
  for (int a = 0; a < R; ++a) acc[a] = 0;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int k = 2 * kl + 64 * q;
    double w[2 * R];
#pragma unroll
    for (int j = 0; j < R; ++j) { const double2 t = *(const double2*)&vs[k * R + 2 * j]; w[2 * j] = t.x; w[2 * j + 1] = t.y; }
#pragma unroll
    for (int a = 0; a < R; ++a) acc[a] += w[a] * m[q].x + w[R + a] * m[q].y;
  }
#pragma unroll
  for (int a = 0; a < R; ++a) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc[a] += __shfl_xor(acc[a], off, 64);
  }
  if (kl == 0) {
#pragma unroll
    for (int a = 0; a < R; ++a) out[(size_t)col * R + a] = acc[a];
  }
}

// V9: stage v completely (loads, LDS write, __syncthreads) BEFORE issuing the M stream; the FMA loop then
// drains m[] in issue order with no barrier in between, so compute overlaps the stream
__global__ __launch_bounds__(256) void k_gemv9(const double* M, const double* V, double* out, int N4) {
  __shared__ double vs[R * KC];
  const int tid = threadIdx.x, cg = tid >> 5, kl = tid & 31;
  const int col = 8 * blockIdx.x + cg;
  const double* Mc = M + (size_t)col * N4;
  {
    double2 v[20];
#pragma unroll
    for (int u = 0; u < 20; ++u) { const int tt = 2 * (tid + 256 * u); v[u] = (tt < N4 * R) ? ld2(V + tt) : make_double2(0, 0); }
#pragma unroll
    for (int u = 0; u < 20; ++u) { const int tt = 2 * (tid + 256 * u); *(double2*)&vs[tt] = v[u]; }
  }
  __syncthreads();
  double2 m[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) { const int k = 2 * kl + 64 * q; m[q] = (k < N4) ? ld2_nt(Mc + k) : make_double2(0, 0); }
  double acc[R];
#pragma unroll
  for (int a = 0; a < R; ++a) acc[a] = 0;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int k = 2 * kl + 64 * q;
    double w[2 * R];
#pragma unroll
    for (int j = 0; j < R; ++j) { const double2 t = *(const double2*)&vs[k * R + 2 * j]; w[2 * j] = t.x; w[2 * j + 1] = t.y; }
#pragma unroll
    for (int a = 0; a < R; ++a) acc[a] += w[a] * m[q].x + w[R + a] * m[q].y;
  }
#pragma unroll
  for (int a = 0; a < R; ++a) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc[a] += __shfl_xor(acc[a], off, 64);
  }
  if (kl == 0) {
#pragma unroll
    for (int a = 0; a < R; ++a) out[(size_t)col * R + a] = acc[a];
  }
}

// V10: everything issued up front (v then M); the barrier is a raw s_barrier that does not drain VMEM
__global__ __launch_bounds__(256) void k_gemv10(const double* M, const double* V, double* out, int N4) {
  __shared__ double vs[R * KC];
  const int tid = threadIdx.x, cg = tid >> 5, kl = tid & 31;
  const int col = 8 * blockIdx.x + cg;
  const double* Mc = M + (size_t)col * N4;
  double2 v[20];
#pragma unroll
  for (int u = 0; u < 20; ++u) { const int tt = 2 * (tid + 256 * u); v[u] = (tt < N4 * R) ? ld2(V + tt) : make_double2(0, 0); }
  double2 m[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) { const int k = 2 * kl + 64 * q; m[q] = (k < N4) ? ld2_nt(Mc + k) : make_double2(0, 0); }
#pragma unroll
  for (int u = 0; u < 20; ++u) { const int tt = 2 * (tid + 256 * u); *(double2*)&vs[tt] = v[u]; }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  double acc[R];
#pragma unroll
  for (int a = 0; a < R; ++a) acc[a] = 0;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int k = 2 * kl + 64 * q;
    double w[2 * R];
#pragma unroll
    for (int j = 0; j < R; ++j) { const double2 t = *(const double2*)&vs[k * R + 2 * j]; w[2 * j] = t.x; w[2 * j + 1] = t.y; }
#pragma unroll
    for (int a = 0; a < R; ++a) acc[a] += w[a] * m[q].x + w[R + a] * m[q].y;
  }
#pragma unroll
  for (int a = 0; a < R; ++a) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc[a] += __shfl_xor(acc[a], off, 64);
  }
  if (kl == 0) {
#pragma unroll
    for (int a = 0; a < R; ++a) out[(size_t)col * R + a] = acc[a];
  }
}

// V11: 2-D tiling (16 column slabs of 128 x 16 k-splits of 125 rows = 256 workgroups).  Lanes run along the
// COLUMN index using M = M^T (row k of the lower/upper triangle is contiguous over columns), the vector
// entries v[k][:] are wave-uniform scalar loads, no LDS staging.  Writes split-k partials [ks][c][a].
__global__ __launch_bounds__(256) void k_gemv11(const double* M, const double* V, double* part, int N4) {
  __shared__ double red[4][128 * R];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slab = blockIdx.x & 15, ks = blockIdx.x >> 4;
  const int c = 128 * slab + 2 * lane;
  const int KS = (N4 + 15) / 16;
  const int kbeg = ks * KS, kend = min(N4, kbeg + KS);
  const int per = (KS + 3) / 4;
  const int k0 = kbeg + w * per, k1 = min(kend, k0 + per);
  const bool cact = c < N4;
  double acc0[R], acc1[R];
#pragma unroll
  for (int a = 0; a < R; ++a) { acc0[a] = 0; acc1[a] = 0; }
  double2 m[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) { const int k = k0 + q; m[q] = (cact && k < k1) ? ld2_nt(M + (size_t)k * N4 + c) : make_double2(0, 0); }
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int k = min(k0 + q, N4 - 1);
#pragma unroll
    for (int a = 0; a < R; ++a) { const double v = V[(size_t)k * R + a]; acc0[a] += v * m[q].x; acc1[a] += v * m[q].y; }
  }
#pragma unroll
  for (int a = 0; a < R; ++a) { red[w][(2 * lane) * R + a] = acc0[a]; red[w][(2 * lane + 1) * R + a] = acc1[a]; }
  __syncthreads();
  for (int t = tid; t < 128 * R; t += 256) {
    const int cc = 128 * slab + t / R;
    if (cc < N4) part[((size_t)ks * N4 + cc) * R + t % R] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
  }
}

// V9a: staging kept, FMA loop replaced by a plain sum of m  (isolates the staging cost)
__global__ __launch_bounds__(256) void k_gemv9a(const double* M, const double* V, double* out, int N4) {
  __shared__ double vs[R * KC];
  const int tid = threadIdx.x, cg = tid >> 5, kl = tid & 31;
  const int col = 8 * blockIdx.x + cg;
  const double* Mc = M + (size_t)col * N4;
  {
    double2 v[20];
#pragma unroll
    for (int u = 0; u < 20; ++u) { const int tt = 2 * (tid + 256 * u); v[u] = (tt < N4 * R) ? ld2(V + tt) : make_double2(0, 0); }
#pragma unroll
    for (int u = 0; u < 20; ++u) { const int tt = 2 * (tid + 256 * u); *(double2*)&vs[tt] = v[u]; }
  }
  __syncthreads();
  double2 m[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) { const int k = 2 * kl + 64 * q; m[q] = (k < N4) ? ld2_nt(Mc + k) : make_double2(0, 0); }
  double s = vs[tid];
#pragma unroll
  for (int q = 0; q < 32; ++q) s += m[q].x + m[q].y;
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (kl == 0) out[col] = s;
}
// V9b: no staging (LDS left uninitialised), FMA loop kept  (isolates the arithmetic + LDS reads)
__global__ __launch_bounds__(256) void k_gemv9b(const double* M, const double* V, double* out, int N4) {
  __shared__ double vs[R * KC];
  const int tid = threadIdx.x, cg = tid >> 5, kl = tid & 31;
  const int col = 8 * blockIdx.x + cg;
  const double* Mc = M + (size_t)col * N4;
  double2 m[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) { const int k = 2 * kl + 64 * q; m[q] = (k < N4) ? ld2_nt(Mc + k) : make_double2(0, 0); }
  double acc[R];
#pragma unroll
  for (int a = 0; a < R; ++a) acc[a] = 0;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int k = 2 * kl + 64 * q;
    double w[2 * R];
#pragma unroll
    for (int j = 0; j < R; ++j) { const double2 t = *(const double2*)&vs[k * R + 2 * j]; w[2 * j] = t.x; w[2 * j + 1] = t.y; }
#pragma unroll
    for (int a = 0; a < R; ++a) acc[a] += w[a] * m[q].x + w[R + a] * m[q].y;
  }
#pragma unroll
  for (int a = 0; a < R; ++a) {
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) acc[a] += __shfl_xor(acc[a], off, 64);
  }
  if (kl == 0) {
#pragma unroll
    for (int a = 0; a < R; ++a) out[(size_t)col * R + a] = acc[a];
  }
}
// V0b: pure stream but with 80 KB of static LDS declared (occupancy / dispatch effect of the LDS footprint)
__global__ __launch_bounds__(256) void k_stream_lds(const double* M, double* out, int N4) {
  __shared__ double vs[R * KC];
  const int tid = threadIdx.x, cg = tid >> 5, kl = tid & 31;
  const int col = 8 * blockIdx.x + cg;
  const double* Mc = M + (size_t)col * N4;
  double2 m[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) { const int k = 2 * kl + 64 * q; m[q] = (k < N4) ? ld2_nt(Mc + k) : make_double2(0, 0); }
  vs[tid] = m[0].x;
  double s = 0;
#pragma unroll
  for (int q = 0; q < 32; ++q) s += m[q].x + m[q].y;
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (kl == 0) out[col] = s + vs[(tid + 7) & 255];
}

// V12<NT>: same as V9 with NT threads per workgroup (8 columns x NT/8 k-lanes), more waves per SIMD
template <int NT>
__global__ __launch_bounds__(NT) void k_gemv12(const double* M, const double* V, double* out, int N4) {
  __shared__ double vs[R * KC];
  __shared__ double red[NT / 64][8 * R];
  constexpr int KL = NT / 8, NQ = KC / (2 * KL), NS = (KC * R / 2 + NT - 1) / NT;
  const int tid = threadIdx.x, cg = tid / KL, kl = tid % KL;
  const int col = 8 * blockIdx.x + cg;
  const double* Mc = M + (size_t)col * N4;
  {
    double2 v[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) { const int tt = 2 * (tid + NT * u); v[u] = (tt < N4 * R) ? ld2(V + tt) : make_double2(0, 0); }
#pragma unroll
    for (int u = 0; u < NS; ++u) { const int tt = 2 * (tid + NT * u); if (tt < KC * R) *(double2*)&vs[tt] = v[u]; }
  }
  __syncthreads();
  double2 m[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { const int k = 2 * kl + 2 * KL * q; m[q] = (k < N4) ? ld2_nt(Mc + k) : make_double2(0, 0); }
  double acc[R];
#pragma unroll
  for (int a = 0; a < R; ++a) acc[a] = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int k = 2 * kl + 2 * KL * q;
    double w[2 * R];
#pragma unroll
    for (int j = 0; j < R; ++j) { const double2 t = *(const double2*)&vs[k * R + 2 * j]; w[2 * j] = t.x; w[2 * j + 1] = t.y; }
#pragma unroll
    for (int a = 0; a < R; ++a) acc[a] += w[a] * m[q].x + w[R + a] * m[q].y;
  }
  // reduce over the KL lanes of a column (KL = 64: one wave per column; KL = 128: two waves)
#pragma unroll
  for (int a = 0; a < R; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[a] += __shfl_xor(acc[a], off, 64);
  }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int a = 0; a < R; ++a) red[tid >> 6][a] = acc[a];
  }
  __syncthreads();
  if (tid < 8 * R) {
    const int c = tid / R, a = tid % R;
    double s2 = 0;
    for (int w = 0; w < KL / 64; ++w) s2 += red[c * (KL / 64) + w][a];
    out[(size_t)(8 * blockIdx.x + c) * R + a] = s2;
  }
}

__global__ void k_noop() {}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <class F> float timeit(F f, int reps, hipStream_t s) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) f();
  CK(hipEventRecord(a, s)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps * 1e3f;
}
int main() {
  const int N4 = 2000, NAG = 5;
  hipStream_t s; CK(hipStreamCreate(&s));
  std::vector<double*> Ms(NAG);
  for (auto& p : Ms) { CK(hipMalloc(&p, sizeof(double) * N4 * N4)); CK(hipMemset(p, 0, sizeof(double) * N4 * N4)); }
  double *V, *out; CK(hipMalloc(&V, sizeof(double) * N4 * R)); CK(hipMalloc(&out, sizeof(double) * N4 * R)); CK(hipMemset(V, 0, sizeof(double) * N4 * R));
  int it = 0;
  printf("noop              %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, s); }, 500, s));
  printf("stream nt  same M %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_stream<1>, dim3(250), dim3(256), 0, s, Ms[0], out, N4); }, 500, s));
  printf("stream nt  5 Ms   %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_stream<1>, dim3(250), dim3(256), 0, s, Ms[(it++) % NAG], out, N4); }, 500, s));
  printf("stream tmp same M %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_stream<0>, dim3(250), dim3(256), 0, s, Ms[0], out, N4); }, 500, s));
  printf("stream tmp 5 Ms   %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_stream<0>, dim3(250), dim3(256), 0, s, Ms[(it++) % NAG], out, N4); }, 500, s));
  printf("gemv lds   5 Ms   %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_gemv, dim3(250), dim3(256), 0, s, Ms[(it++) % NAG], V, out, N4); }, 500, s));
  printf("gemv2 regs 5 Ms   %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_gemv2, dim3(250), dim3(256), 0, s, Ms[(it++) % NAG], V, out, N4); }, 500, s));
  printf("gemv3 v-first T    %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_gemv3, dim3(250), dim3(256), 0, s, Ms[(it++) % NAG], V, out, N4); }, 500, s));
  printf("gemv7 v-first str  %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_gemv7, dim3(250), dim3(256), 0, s, Ms[(it++) % NAG], V, out, N4); }, 500, s));
  printf("gemv8 M-first str  %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_gemv8, dim3(250), dim3(256), 0, s, Ms[(it++) % NAG], V, out, N4); }, 500, s));
  printf("gemv9 stage->strm  %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_gemv9, dim3(250), dim3(256), 0, s, Ms[(it++) % NAG], V, out, N4); }, 500, s));
  printf("gemv10 raw barrier %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_gemv10, dim3(250), dim3(256), 0, s, Ms[(it++) % NAG], V, out, N4); }, 500, s));
  double* part; CK(hipMalloc(&part, sizeof(double) * 16 * N4 * R));
  printf("gemv11 2D split-k  %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_gemv11, dim3(256), dim3(256), 0, s, Ms[(it++) % NAG], V, part, N4); }, 500, s));
  printf("gemv9a stage only  %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_gemv9a, dim3(250), dim3(256), 0, s, Ms[(it++) % NAG], V, out, N4); }, 500, s));
  printf("gemv9b fma only    %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_gemv9b, dim3(250), dim3(256), 0, s, Ms[(it++) % NAG], V, out, N4); }, 500, s));
  printf("stream + 80KB LDS  %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_stream_lds, dim3(250), dim3(256), 0, s, Ms[(it++) % NAG], out, N4); }, 500, s));
  printf("gemv12 512 thr     %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_gemv12<512>, dim3(250), dim3(512), 0, s, Ms[(it++) % NAG], V, out, N4); }, 500, s));
  printf("gemv12 1024 thr    %.2f us\n", timeit([&] { hipLaunchKernelGGL(k_gemv12<1024>, dim3(250), dim3(1024), 0, s, Ms[(it++) % NAG], V, out, N4); }, 500, s));
  // graph of 20 launches
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_gemv2, dim3(250), dim3(256), 0, s, Ms[i % NAG], V, out, N4);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  printf("gemv2 in graph    %.2f us per kernel\n", timeit([&] { CK(hipGraphLaunch(ge, s)); }, 50, s) / 20);
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_gemv, dim3(250), dim3(256), 0, s, Ms[i % NAG], V, out, N4);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  printf("gemv  in graph    %.2f us per kernel\n", timeit([&] { CK(hipGraphLaunch(ge, s)); }, 50, s) / 20);
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_stream<1>, dim3(250), dim3(256), 0, s, Ms[i % NAG], out, N4);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  printf("stream in graph   %.2f us per kernel\n", timeit([&] { CK(hipGraphLaunch(ge, s)); }, 50, s) / 20);
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_noop, dim3(1), dim3(64), 0, s);
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  printf("noop in graph     %.2f us per kernel\n", timeit([&] { CK(hipGraphLaunch(ge, s)); }, 50, s) / 20);
  return 0;
}
