// dense_inverse.hip -- M = A^{-1} for a dense SPD matrix on the GPU (fp64), used once per data
// matrix to turn the reference's sparse-Cholesky preconditioner solve (SURVEY 8a-a2/a3:
// "P = chol(Q + eps I)", rebuilt only on clearDataMatrices(), src/PGOAgentROS.cpp:1351) into a
// bandwidth-bound dense apply that the whole chip can share.  MI355X-first trade: N4^2 doubles of
// HBM (32 MB at n = 500, 288 GB available) buy a solve with no sequential dependency chain.
//
// Algorithm: blocked right-looking Cholesky A = L L^T (NB = 32), blocked triangular inverse
// W = L^{-1}, then M = W^T W.  All column-major, lower triangle referenced.
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace dpgo {

constexpr int NB = 32;

// factor the nb x nb diagonal block at (k0,k0) in place and write its inverse (lower, dense NB x NB
// column-major, zero padded) to Linv.  One workgroup of NB x NB threads.
__global__ __launch_bounds__(1024) void k_potrf_diag(double *A, int N, int k0, int nb, double *Linv, int *fail) {
  __shared__ double L[NB][NB + 1];
  __shared__ double Wi[NB][NB + 1];
  const int i = threadIdx.x % NB, j = threadIdx.x / NB;
  L[i][j] = (i < nb && j < nb && i >= j) ? A[(size_t)(k0 + j) * N + k0 + i] : (i == j ? 1.0 : 0.0);
  Wi[i][j] = (i == j) ? 1.0 : 0.0;
  __syncthreads();
  for (int k = 0; k < NB; ++k) {
    if (i == k && j == k) {
      const double d = L[k][k];
      if (!(d > 0.0)) { *fail = k0 + k + 1; L[k][k] = 1.0; } else L[k][k] = sqrt(d);
    }
    __syncthreads();
    if (j == k && i > k) L[i][k] /= L[k][k];
    __syncthreads();
    if (j > k && i >= j) L[i][j] -= L[i][k] * L[j][k];
    __syncthreads();
  }
  // forward substitution on the identity: column j of Wi solves L w = e_j
  for (int k = 0; k < NB; ++k) {
    if (i == k) Wi[k][j] /= L[k][k];
    __syncthreads();
    if (i > k) Wi[i][j] -= L[i][k] * Wi[k][j];
    __syncthreads();
  }
  if (i < nb && j < nb && i >= j) A[(size_t)(k0 + j) * N + k0 + i] = L[i][j];
  Linv[j * NB + i] = (i < nb && j < nb && i >= j) ? Wi[i][j] : 0.0;
}

// panel: A[i, k0:k0+nb] <- A[i, k0:k0+nb] Linv^T  for i >= k0 + nb.  One thread per row.
__global__ __launch_bounds__(256) void k_trsm_panel(double *A, int N, int k0, int nb, const double *Linv) {
  __shared__ double Ls[NB * NB];
  for (int t = threadIdx.x; t < NB * NB; t += 256) Ls[t] = Linv[t];
  __syncthreads();
  const int i = k0 + nb + blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  double row[NB], out[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) row[k] = (k < nb) ? A[(size_t)(k0 + k) * N + i] : 0.0;
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    double s = 0;
#pragma unroll
    for (int k = 0; k < NB; ++k)
      if (k <= c) s += row[k] * Ls[k * NB + c];  // Linv[c,k]
    out[c] = s;
  }
#pragma unroll
  for (int c = 0; c < NB; ++c)
    if (c < nb) A[(size_t)(k0 + c) * N + i] = out[c];
}

// 64x64 output tile, K = NB slab staged in LDS; thread (tx,ty) owns rows tx+16u, cols ty+16v
__device__ __forceinline__ void tile_mac(const double (*As)[65], const double (*Bs)[65], int kn, int tx, int ty,
                                         double acc[4][4]) {
  for (int k = 0; k < kn; ++k) {
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = As[k][tx + 16 * u]; b[u] = Bs[k][ty + 16 * u]; }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * b[v];
  }
}

// trailing update: A[i,j] -= sum_k P[i,k] P[j,k], i >= j >= s0 (= k0 + nb), P = columns k0..k0+nb
__global__ __launch_bounds__(256) void k_syrk(double *A, int N, int k0, int nb, int s0) {
  const int bi = blockIdx.x, bj = blockIdx.y;
  if (bj > bi) return;
  __shared__ double As[NB][65], Bs[NB][65];
  const int i0 = s0 + 64 * bi, j0 = s0 + 64 * bj;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  for (int t = tid; t < NB * 64; t += 256) {
    const int k = t >> 6, ii = t & 63;
    As[k][ii] = (k < nb && i0 + ii < N) ? A[(size_t)(k0 + k) * N + i0 + ii] : 0.0;
    Bs[k][ii] = (k < nb && j0 + ii < N) ? A[(size_t)(k0 + k) * N + j0 + ii] : 0.0;
  }
  __syncthreads();
  double acc[4][4] = {};
  tile_mac(As, Bs, nb, tx, ty, acc);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = i0 + tx + 16 * u, j = j0 + ty + 16 * v;
      if (i < N && j < N && i >= j) A[(size_t)j * N + i] -= acc[u][v];
    }
}

// block row ib of W = L^{-1}:  W[ib,jb] = -Linv_ib (sum_{jb<=kb<ib} L[ib,kb] W[kb,jb]),  W[ib,ib] = Linv_ib
// grid.x = jb in [0, ib]; 1024 threads = one NB x NB block.
__global__ __launch_bounds__(1024) void k_trtri_row(const double *L, double *W, int N, int ib, const double *LinvAll) {
  const int jb = blockIdx.x;
  __shared__ double Ls[NB][NB + 1], Ws[NB][NB + 1], Ts[NB][NB + 1];
  const int i = threadIdx.x % NB, j = threadIdx.x / NB;
  const int r0 = ib * NB, c0 = jb * NB;
  const double *Linv = LinvAll + (size_t)ib * NB * NB;
  if (jb == ib) {
    if (r0 + i < N && c0 + j < N) W[(size_t)(c0 + j) * N + r0 + i] = Linv[j * NB + i];
    return;
  }
  double acc = 0;
  for (int kb = jb; kb < ib; ++kb) {
    const int k0 = kb * NB;
    __syncthreads();
    Ls[i][j] = (r0 + i < N) ? L[(size_t)(k0 + j) * N + r0 + i] : 0.0;  // L[ib,kb](i,j)
    Ws[i][j] = W[(size_t)(c0 + j) * N + k0 + i];                        // W[kb,jb](i,j)
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < NB; ++k) acc += Ls[i][k] * Ws[k][j];
  }
  __syncthreads();
  Ts[i][j] = acc;
  __syncthreads();
  double s = 0;
#pragma unroll 8
  for (int k = 0; k < NB; ++k) s += Linv[k * NB + i] * Ts[k][j];  // Linv(i,k)
  if (r0 + i < N) W[(size_t)(c0 + j) * N + r0 + i] = -s;
}

// M = W^T W for lower-triangular W (upper part of W must be zero); tiles with bi >= bj, mirrored.
__global__ __launch_bounds__(256) void k_wtw(const double *W, double *M, int N) {
  const int bi = blockIdx.x, bj = blockIdx.y;
  if (bj > bi) return;
  __shared__ double As[NB][65], Bs[NB][65];
  const int i0 = 64 * bi, j0 = 64 * bj;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  double acc[4][4] = {};
  for (int kk = (i0 / NB) * NB; kk < N; kk += NB) {
    __syncthreads();
    for (int t = tid; t < NB * 64; t += 256) {
      const int k = t & 31, ii = t >> 5;
      As[k][ii] = (kk + k < N && i0 + ii < N) ? W[(size_t)(i0 + ii) * N + kk + k] : 0.0;
      Bs[k][ii] = (kk + k < N && j0 + ii < N) ? W[(size_t)(j0 + ii) * N + kk + k] : 0.0;
    }
    __syncthreads();
    tile_mac(As, Bs, NB, tx, ty, acc);
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = i0 + tx + 16 * u, j = j0 + ty + 16 * v;
      if (i < N && j < N) {
        M[(size_t)j * N + i] = acc[u][v];
        M[(size_t)i * N + j] = acc[u][v];
      }
    }
}

int dense_spd_inverse(hipStream_t stream, double *A, double *work, double *M, int N) {
  const int nblk = (N + NB - 1) / NB;
  double *LinvAll = nullptr;
  int *fail_d = nullptr;
  if (hipMalloc(&LinvAll, sizeof(double) * NB * NB * (size_t)nblk) != hipSuccess) return -1;
  if (hipMalloc(&fail_d, sizeof(int)) != hipSuccess) { (void)hipFree(LinvAll); return -1; }
  (void)hipMemsetAsync(fail_d, 0, sizeof(int), stream);
  (void)hipMemsetAsync(work, 0, sizeof(double) * (size_t)N * N, stream);
  for (int kb = 0; kb < nblk; ++kb) {
    const int k0 = kb * NB, nb = (N - k0 < NB) ? N - k0 : NB, s0 = k0 + nb;
    hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(NB * NB), 0, stream, A, N, k0, nb, LinvAll + (size_t)kb * NB * NB,
                       fail_d);
    if (s0 < N) {
      hipLaunchKernelGGL(k_trsm_panel, dim3((N - s0 + 255) / 256), dim3(256), 0, stream, A, N, k0, nb,
                         LinvAll + (size_t)kb * NB * NB);
      const int nt = (N - s0 + 63) / 64;
      hipLaunchKernelGGL(k_syrk, dim3(nt, nt), dim3(256), 0, stream, A, N, k0, nb, s0);
    }
  }
  for (int ib = 0; ib < nblk; ++ib)
    hipLaunchKernelGGL(k_trtri_row, dim3(ib + 1), dim3(NB * NB), 0, stream, A, work, N, ib, LinvAll);
  const int nt = (N + 63) / 64;
  hipLaunchKernelGGL(k_wtw, dim3(nt, nt), dim3(256), 0, stream, work, M, N);
  int fail = 0;
  (void)hipMemcpyAsync(&fail, fail_d, sizeof(int), hipMemcpyDeviceToHost, stream);
  (void)hipStreamSynchronize(stream);
  (void)hipFree(LinvAll);
  (void)hipFree(fail_d);
  return fail;
}

}  // namespace dpgo
