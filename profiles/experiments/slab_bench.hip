// slab_bench.hip -- the LDS slab product of the one-launch RTR solve (rtr_fused.hip: slab_issue / slab_finish) in
// isolation: 250 workgroups x 256 threads, 128 KB of M per workgroup in LDS, `iters` products per launch.  Reports the
// time per product by the 100 MHz wall clock and the shader clock (s_memtime), i.e. the effective clock as well.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -I dpgo_ros_amd/csrc profiles/experiments/slab_bench.hip -o /tmp/slab_bench
#include "rtr_fused.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <algorithm>
using namespace dpgo;

#ifndef SB_VARIANT
#define SB_VARIANT 0
#endif

// VAR 0: slab_issue + slab_finish as the solve runs them; 1: the vector loaded once in front of the loop (LDS + FMA +
// reduction only); 2: variant 1 without the cross-lane reduction (products only)
template <int R, int VAR>
__global__ __launch_bounds__(256) void k_slab(const double *M, const double *V, double *out, unsigned long long *tm, int N4, int iters) {
  extern __shared__ double Ms[];
  __shared__ double red[64 * (8 * R + 1)];
  __shared__ double zs[8 * R];
  const int tid = threadIdx.x, bx = blockIdx.x;
  for (int i = tid; i < 4 * N4; i += 256) *reinterpret_cast<double2 *>(&Ms[2 * i]) = *reinterpret_cast<const double2 *>(&M[(size_t)8 * bx * N4 + 2 * i]);
  if (tid < 8 * R) zs[tid] = 0;
  __syncthreads();
  double accum = 0;
  double2 vv[SLAB_MAXM][R];
  if (VAR >= 1) { const PVec cV(V); slab_issue<R>(N4, cV, tid, vv); }
  const unsigned long long t0 = wall_clock64(), c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (VAR == 0) { const PVec cV(V + (size_t)(it & 7) * N4 * R); slab_issue<R>(N4, cV, tid, vv); }
    if (VAR <= 1) {
      slab_finish<R>(Ms, N4, vv, red, zs, tid);
    } else {
      double acc[8][R];
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int a = 0; a < R; ++a) acc[c][a] = 0.0;
#pragma unroll
      for (int m = 0; m < SLAB_MAXM; ++m) {
        const int k = 2 * tid + 512 * m, kk = min(k, N4 - 2);
        const double live = (k < N4) ? 1.0 : 0.0;
        double2 mm[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) mm[c] = *reinterpret_cast<const double2 *>(&Ms[(size_t)c * N4 + kk]);
        double w[2 * R];
#pragma unroll
        for (int q = 0; q < R; ++q) { w[2 * q] = vv[m][q].x * live; w[2 * q + 1] = vv[m][q].y * live; }
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int a = 0; a < R; ++a) acc[c][a] = __builtin_fma(w[a], mm[c].x, acc[c][a]);
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int a = 0; a < R; ++a) acc[c][a] = __builtin_fma(w[R + a], mm[c].y, acc[c][a]);
      }
      double s = 0;
      if (VAR == 2) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int a = 0; a < R; ++a) s += acc[c][a];
      } else {  // 3: + the lane swaps (8R -> 2R values per lane); 4: + the row butterflies
        constexpr int N = 8 * R, H = N / 2, Q = N / 4;
        double s1[H], u[Q];
#pragma unroll
        for (int i = 0; i < H; ++i) s1[i] = swap_add_32(acc[i / R][i % R], acc[(i + H) / R][(i + H) % R]);
#pragma unroll
        for (int i = 0; i < Q; ++i) u[i] = swap_add_16(s1[i], s1[i + Q]);
        if (VAR >= 4) {
#pragma unroll
          for (int i = 0; i < Q; ++i) {
            double x = u[i];
            x += dpp_move<0xB1>(x); x += dpp_move<0x4E>(x); x += dpp_move<0x141>(x); x += dpp_move<0x140>(x);
            u[i] = x;
          }
        }
#pragma unroll
        for (int i = 0; i < Q; ++i) s += u[i];
      }
      accum += s;
      // (keep the loop body from being hoisted: the vector changes a little every pass)
#pragma unroll
      for (int q = 0; q < R; ++q) vv[0][q].x += 1e-300 * s;
    }
    if (VAR <= 1) {
      if (tid < 8 * R) accum += zs[tid];
      if (VAR == 1) {
#pragma unroll
        for (int q = 0; q < R; ++q) vv[0][q].x += 1e-300 * accum;
      }
      __syncthreads();
    }
  }
  const unsigned long long t1 = wall_clock64(), c1 = __builtin_amdgcn_s_memtime();
  out[(size_t)bx * 256 + tid] = accum;
  if (tid == 0) { tm[4 * bx] = t0; tm[4 * bx + 1] = t1; tm[4 * bx + 2] = c0; tm[4 * bx + 3] = c1; }
}

static const double *hostM = nullptr;
template <int VAR>
static void run(const char *name, const double *dM, const double *dV, double *dout, unsigned long long *dtm, int N4, int grid, int iters) {
  constexpr int R = 5;
  const size_t dyn = (size_t)64 * N4;
  hipFuncSetAttribute((const void *)k_slab<R, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((k_slab<R, VAR>), dim3(grid), dim3(256), dyn, 0, dM, dV, dout, dtm, N4, iters);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> tm(4 * grid);
  hipMemcpy(tm.data(), dtm, sizeof(unsigned long long) * 4 * grid, hipMemcpyDeviceToHost);
  double us = 0, clk = 0;
  for (int b = 0; b < grid; ++b) { us += (tm[4 * b + 1] - tm[4 * b]) / 100.0; clk += (double)(tm[4 * b + 3] - tm[4 * b + 2]); }
  us /= grid; clk /= grid;
  printf("%-44s %7.3f us per product, %8.0f shader clocks per product, %.2f GHz\n", name, us / iters, clk / iters, clk / us / 1e3);
  if (VAR == 0 && hostM) {  // the sums against a host evaluation (workgroups 0, 7, grid - 1)
    std::vector<double> o(256 * (size_t)grid);
    hipMemcpy(o.data(), dout, sizeof(double) * o.size(), hipMemcpyDeviceToHost);
    double worst = 0;
    for (int b : {0, 7, grid - 1})
      for (int t = 0; t < 8 * R; ++t) {
        const int c = t / R, a = t % R;
        double ref = 0;
        for (int it = 0; it < iters; ++it) {
          const double *V = hostM + (size_t)(it & 7) * N4 * R;
          double z = 0;
          for (int k = 0; k < N4; ++k) z += V[(size_t)k * R + a] * hostM[(size_t)(8 * b + c) * N4 + k];
          ref += z;
        }
        worst = std::max(worst, std::fabs(o[(size_t)b * 256 + t] - ref) / std::fabs(ref));
      }
    printf("   max relative deviation from the host sums: %.2e\n", worst);
  }
}

int main() {
  const int n = 500, N4 = 4 * n, grid = n / 2, iters = 200;
  double *dM, *dV, *dout;
  unsigned long long *dtm;
  hipMalloc(&dM, sizeof(double) * (size_t)N4 * N4);
  hipMalloc(&dV, sizeof(double) * (size_t)N4 * 5 * 8);
  hipMalloc(&dout, sizeof(double) * 256 * grid);
  hipMalloc(&dtm, sizeof(unsigned long long) * 4 * grid);
  std::vector<double> h((size_t)N4 * N4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (double)((i * 2654435761u) % 1000);
  hipMemcpy(dM, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  hipMemcpy(dV, h.data(), sizeof(double) * (size_t)N4 * 5 * 8, hipMemcpyHostToDevice);
  hostM = h.data();
  run<0>("vector from L2 + product + reduction", dM, dV, dout, dtm, N4, grid, iters);
  run<1>("product + reduction (vector in registers)", dM, dV, dout, dtm, N4, grid, iters);
  run<2>("product only", dM, dV, dout, dtm, N4, grid, iters);
  run<3>("product + lane swaps", dM, dV, dout, dtm, N4, grid, iters);
  run<4>("product + lane swaps + row butterflies", dM, dV, dout, dtm, N4, grid, iters);
  return 0;
}
