// launch_latency.hip -- floor of one host -> device -> host round trip on this box: K chained tiny kernels, the last of
// which stores a sequence word into coherent pinned host memory that the host polls; against hipStreamSynchronize;
// and the cost of a 32 KB payload written by that kernel into coherent (fine-grained) or ordinary pinned host memory.
// hipcc --offload-arch=gfx950 -O3 launch_latency.hip -o launch_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define HC(x) do { if ((x) != hipSuccess) { printf("%s failed\n", #x); return 1; } } while (0)
__global__ void k_tiny(double *x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.0; }
__global__ void k_flag(double *x, unsigned long long *flag, unsigned long long v, double *payload, int n, const double *src) {
  for (int t = threadIdx.x; t < n; t += blockDim.x) payload[t] = src[t] + (double)v;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    x[0] += 1.0;
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
int main() {
  double *x, *src, *pay_c, *pay_n;
  unsigned long long *flag;
  HC(hipMalloc(&x, 8)); HC(hipMemset(x, 0, 8));
  HC(hipMalloc(&src, 8 * 4096)); HC(hipMemset(src, 0, 8 * 4096));
  HC(hipHostMalloc((void **)&flag, 64, hipHostMallocCoherent)); *flag = 0;
  HC(hipHostMalloc((void **)&pay_c, 8 * 4096, hipHostMallocCoherent));
  HC(hipHostMalloc((void **)&pay_n, 8 * 4096, hipHostMallocNonCoherent));
  hipStream_t s; HC(hipStreamCreate(&s));
  unsigned long long seq = 0;
  for (int K = 0; K <= 2; ++K) {
    for (int mode = 0; mode < 2; ++mode) {
      double tot = 0; const int reps = 2000;
      for (int r = 0; r < reps + 100; ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < K; ++k) hipLaunchKernelGGL(k_tiny, dim3(8), dim3(64), 0, s, x);
        ++seq;
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, x, flag, seq, pay_c, 0, src);
        if (mode == 0) { while (*(volatile unsigned long long *)flag != seq) __builtin_ia32_pause(); }
        else HC(hipStreamSynchronize(s));
        if (r >= 100) tot += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      }
      printf("%d tiny kernels + flag kernel, %s: %.2f us per round trip\n", K, mode ? "hipStreamSynchronize" : "polled flag", tot / reps);
    }
  }
  for (int which = 0; which < 2; ++which)
    for (int n : {512, 4000}) {
      double tot = 0; const int reps = 2000; int bad = 0;
      double *pay = which ? pay_n : pay_c;
      for (int r = 0; r < reps + 100; ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        ++seq;
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(1024), 0, s, x, flag, seq, pay, n, src);
        while (*(volatile unsigned long long *)flag != seq) __builtin_ia32_pause();
        if (r >= 100) tot += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (((volatile double *)pay)[n - 1] != (double)seq || ((volatile double *)pay)[0] != (double)seq) ++bad;
      }
      printf("flag kernel + %d doubles into %s pinned memory, polled: %.2f us per round trip, %d stale payloads\n", n,
             which ? "ordinary (non-coherent)" : "coherent", tot / reps, bad);
    }
  return 0;
}
