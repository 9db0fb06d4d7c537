// grid-wide barrier cost on MI355X: 256 workgroups x 256 threads (one per CU), sense-reversing barrier with
// agent-scope atomics; kernel time with 0 / 1 / 2 barriers.  hipcc --offload-arch=gfx950 -O3 gridbar.hip -o gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
struct Bar { unsigned count; unsigned gen; unsigned fail; };
__device__ __forceinline__ void grid_barrier(Bar *b, unsigned nblk) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned g = __hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned old = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nblk - 1) {
      __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&b->gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned polls = 0;
      while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) {
        __builtin_amdgcn_s_sleep(1);
        if (++polls > 4000000u) { b->fail = 1; break; }
      }
    }
    __threadfence();
  }
  __syncthreads();
}
__global__ __launch_bounds__(256) void k(Bar *b, double *buf, int nbar) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double v = buf[i];
  for (int r = 0; r < nbar; ++r) {
    buf[i] = v + 1.0;
    grid_barrier(b, gridDim.x);
    v = buf[(i + 256 * 17) % (gridDim.x * 256)];   // written by another workgroup before the barrier
  }
  buf[i] = v;
}
int main() {
  Bar *b; double *buf; const int nb = 256;
  hipMalloc(&b, sizeof(Bar)); hipMemset(b, 0, sizeof(Bar));
  hipMalloc(&buf, sizeof(double) * nb * 256); hipMemset(buf, 0, sizeof(double) * nb * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nbar = 0; nbar <= 3; ++nbar) {
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, b, buf, nbar);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 500; ++r) hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, b, buf, nbar);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    Bar h; hipMemcpy(&h, b, sizeof h, hipMemcpyDeviceToHost);
    double s; hipMemcpy(&s, buf + 5, sizeof s, hipMemcpyDeviceToHost);
    printf("barriers %d: %.2f us per launch  (fail %u, buf %.0f)\n", nbar, ms / 500 * 1e3, h.fail, s);
  }
  return 0;
}
