// dense_inverse.hip -- M = A^{-1} for a dense SPD matrix on the GPU (fp64), used once per data
// matrix to turn the reference's sparse-Cholesky preconditioner solve (SURVEY 8a-a2/a3:
// "P = chol(Q + eps I)", rebuilt only on clearDataMatrices(), src/PGOAgentROS.cpp:1351) into a
// bandwidth-bound dense apply that the whole chip can share.  MI355X-first trade: N4^2 doubles of
// HBM (32 MB at n = 500, 288 GB available) buy a solve with no sequential dependency chain.
//
// Algorithm: blocked right-looking Cholesky A = L L^T (NB = 32), blocked triangular inverse
// W = L^{-1}, then M = W^T W.  All column-major, lower triangle referenced.
#include <hip/hip_runtime.h>
#include <vector>

#include "kernels.h"

namespace dpgo {

constexpr int NB = 32;

// one matrix of a batch: the kernels take the batch index from blockIdx.z, so that the many small, dependent steps
// of several inversions (one per agent) share their launches
struct InvJob {
  double *A, *W, *M, *Linv;  // matrix (destroyed), work / triangular inverse, result, per-block inverse diagonal factors
  int N, nblk;
};

// factor the nb x nb diagonal block at (k0,k0) in place and write its inverse (lower, dense NB x NB
// column-major, zero padded) to Linv.  One wave: lane i < NB keeps row i of the block in registers; the column that
// every step produces is exchanged through LDS (broadcast reads), so there is no workgroup barrier in the 2 x NB
// dependent steps (1024 threads with three barriers per step took 26 us, this takes about 5).
__global__ __launch_bounds__(64) void k_potrf_diag(const InvJob *jobs, int kb, int *fail) {
  const InvJob jb_ = jobs[blockIdx.z];
  if (kb >= jb_.nblk) return;
  double *A = jb_.A;
  const int N = jb_.N, k0 = kb * NB, nb = min(NB, N - k0);
  double *Linv = jb_.Linv + (size_t)kb * NB * NB;
  __shared__ double Ls[NB][NB + 1];  // the factor, row-major, for the broadcast reads of the inverse
  __shared__ double col[NB], idiag[NB];
  const int lane = threadIdx.x;
  const bool own = lane < NB;
  double row[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
    row[j] = (own && lane < nb && j < nb && j <= lane) ? A[(size_t)(k0 + j) * N + k0 + lane] : ((own && j == lane) ? 1.0 : 0.0);
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    // pivot: lane k holds L[k][k] after the updates of the previous steps
    double d = __shfl(row[k], k, 64);
    if (!(d > 0.0)) { if (lane == 0) fail[blockIdx.z] = k0 + k + 1; d = 1.0; }
    const double inv = 1.0 / sqrt(d);  // one division per step; the column and the inverse below only multiply
    double lik = 0.0;
    if (own && lane >= k) { lik = (lane == k) ? d * inv : row[k] * inv; row[k] = lik; }
    if (own) col[lane] = (lane >= k) ? lik : 0.0;
    if (lane == k) idiag[k] = inv;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (own && lane > k) {
#pragma unroll
      for (int j = k + 1; j < NB; ++j)
        if (j <= lane) row[j] -= lik * col[j];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (own) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      Ls[lane][j] = (j <= lane) ? row[j] : 0.0;
      if (lane < nb && j < nb && j <= lane) A[(size_t)(k0 + j) * N + k0 + lane] = row[j];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // inverse: lane j solves L w = e_j by forward substitution (w[i] = 0 for i < j)
  if (own) {
    double w[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      double sres = (i == lane) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < NB; ++k)
        if (k < i) sres -= Ls[i][k] * w[k];
      w[i] = (i >= lane) ? sres * idiag[i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) Linv[lane * NB + i] = (i < nb && lane < nb && i >= lane) ? w[i] : 0.0;
  }
}

// panel: A[i, k0:k0+nb] <- A[i, k0:k0+nb] Linv^T  for i >= k0 + nb.  Workgroup = 64 rows x 4 column octets:
// the 64 x NB row tile goes through LDS once, every thread forms 8 of the NB outputs of its row.
__global__ __launch_bounds__(256) void k_trsm_panel(const InvJob *jobs, int kb) {
  const InvJob jb_ = jobs[blockIdx.z];
  if (kb >= jb_.nblk) return;
  double *A = jb_.A;
  const int N = jb_.N, k0 = kb * NB, nb = min(NB, N - k0);
  const double *Linv = jb_.Linv + (size_t)kb * NB * NB;
  if (k0 + nb + (int)blockIdx.x * 64 >= N) return;
  __shared__ double Ls[NB * NB];
  __shared__ double Rs[64][NB + 1];
  const int tid = threadIdx.x, r = tid & 63, cq = tid >> 6;
  const int i0 = k0 + nb + blockIdx.x * 64;
  for (int t = tid; t < NB * NB; t += 256) Ls[t] = Linv[t];
  for (int t = tid; t < 64 * NB; t += 256) {
    const int k = t >> 6, ii = t & 63;  // consecutive threads = consecutive rows of one column: coalesced
    Rs[ii][k] = (k < nb && i0 + ii < N) ? A[(size_t)(k0 + k) * N + i0 + ii] : 0.0;
  }
  __syncthreads();
  const int i = i0 + r;
  if (i >= N) return;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = cq * 8 + u;
    double s = 0;
    for (int k = 0; k <= c; ++k) s += Rs[r][k] * Ls[k * NB + c];  // Linv[c,k]
    if (c < nb) A[(size_t)(k0 + c) * N + i] = s;
  }
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
__global__ __launch_bounds__(256) void k_syrk(const InvJob *jobs, int kb) {
  const InvJob jb_ = jobs[blockIdx.z];
  if (kb >= jb_.nblk) return;
  double *A = jb_.A;
  const int N = jb_.N, k0 = kb * NB, nb = min(NB, N - k0), s0 = k0 + nb;
  const int bi = blockIdx.x, bj = blockIdx.y;
  if (bj > bi || s0 + 64 * bi >= N) return;
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

// W = L^{-1}, right-looking by block rows.  Block (ib, jb), jb < ib, of the work matrix accumulates
// S[ib,jb] = sum_{jb<=kb<ib} L[ib,kb] W[kb,jb]; once every block row above ib is final,
//   W[ib,ib] = Linv_ib,   W[ib,jb] = -Linv_ib S[ib,jb]                                  (k_trtri_fin, ib + 1 workgroups)
// and the new row is pushed into all rows below:  S[i,jb] += L[i,ib] W[ib,jb], i > ib, jb <= ib   (k_trtri_upd, 64x64 tiles).
__global__ __launch_bounds__(1024) void k_trtri_fin(const InvJob *jobs, int ib) {
  const InvJob jb_ = jobs[blockIdx.z];
  if (ib >= jb_.nblk) return;
  double *W = jb_.W;
  const int N = jb_.N;
  const double *LinvAll = jb_.Linv;
  const int jb = blockIdx.x;
  __shared__ double Ts[NB][NB + 1];
  const int i = threadIdx.x % NB, j = threadIdx.x / NB;
  const int r0 = ib * NB, c0 = jb * NB;
  const double *Linv = LinvAll + (size_t)ib * NB * NB;
  const bool in = r0 + i < N && c0 + j < N;
  if (jb == ib) {
    if (in) W[(size_t)(c0 + j) * N + r0 + i] = Linv[j * NB + i];
    return;
  }
  Ts[i][j] = in ? W[(size_t)(c0 + j) * N + r0 + i] : 0.0;
  __syncthreads();
  double s = 0;
#pragma unroll 8
  for (int k = 0; k < NB; ++k) s += Linv[k * NB + i] * Ts[k][j];  // Linv(i,k)
  if (in) W[(size_t)(c0 + j) * N + r0 + i] = -s;
}

__global__ __launch_bounds__(256) void k_trtri_upd(const InvJob *jobs, int ib) {
  const InvJob jb_ = jobs[blockIdx.z];
  if (ib >= jb_.nblk) return;
  const double *L = jb_.A;
  double *W = jb_.W;
  const int N = jb_.N;
  __shared__ double As[NB][65], Bs[NB][65];
  const int k0 = ib * NB, kn = min(NB, N - k0);
  if (k0 + NB + 64 * (int)blockIdx.x >= N || 64 * (int)blockIdx.y >= k0 + NB) return;
  const int i0 = k0 + NB + 64 * blockIdx.x, j0 = 64 * blockIdx.y;  // rows below block row ib, columns up to it
  const int jend = min(N, k0 + NB);
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  for (int t = tid; t < NB * 64; t += 256) {
    const int k = t >> 6, ii = t & 63;
    As[k][ii] = (k < kn && i0 + ii < N) ? L[(size_t)(k0 + k) * N + i0 + ii] : 0.0;  // L[i, k0 + k]
  }
  for (int t = tid; t < NB * 64; t += 256) {
    const int k = t & 31, jj = t >> 5;
    Bs[k][jj] = (k < kn && j0 + jj < jend) ? W[(size_t)(j0 + jj) * N + k0 + k] : 0.0;  // W[k0 + k, j]
  }
  __syncthreads();
  double acc[4][4] = {};
  tile_mac(As, Bs, kn, tx, ty, acc);
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int i = i0 + tx + 16 * u, j = j0 + ty + 16 * v;
      if (i < N && j < jend) W[(size_t)j * N + i] += acc[u][v];
    }
}

// M = W^T W for lower-triangular W (upper part of W must be zero); tiles with bi >= bj, mirrored.
__global__ __launch_bounds__(256) void k_wtw(const InvJob *jobs) {
  const InvJob jb_ = jobs[blockIdx.z];
  const double *W = jb_.W;
  double *M = jb_.M;
  const int N = jb_.N;
  const int bi = blockIdx.x, bj = blockIdx.y;
  if (bj > bi || 64 * bi >= N) return;
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

int dense_spd_inverse_batched(hipStream_t stream, int count, double *const *A, double *const *work, double *const *M,
                              const int *N) {
  if (count <= 0) return 0;
  std::vector<InvJob> jobs(count);
  size_t linv_total = 0;
  int max_n = 0, max_blk = 0;
  for (int b = 0; b < count; ++b) {
    jobs[b].A = A[b]; jobs[b].W = work[b]; jobs[b].M = M[b]; jobs[b].N = N[b];
    jobs[b].nblk = (N[b] + NB - 1) / NB;
    linv_total += (size_t)jobs[b].nblk * NB * NB;
    max_n = N[b] > max_n ? N[b] : max_n;
    max_blk = jobs[b].nblk > max_blk ? jobs[b].nblk : max_blk;
  }
  double *LinvAll = nullptr;
  int *fail_d = nullptr;
  InvJob *jobs_d = nullptr;
  if (hipMalloc(&LinvAll, sizeof(double) * linv_total) != hipSuccess) return -1;
  if (hipMalloc(&fail_d, sizeof(int) * count) != hipSuccess) { (void)hipFree(LinvAll); return -1; }
  if (hipMalloc(&jobs_d, sizeof(InvJob) * count) != hipSuccess) { (void)hipFree(LinvAll); (void)hipFree(fail_d); return -1; }
  {
    size_t off = 0;
    for (int b = 0; b < count; ++b) { jobs[b].Linv = LinvAll + off; off += (size_t)jobs[b].nblk * NB * NB; }
  }
  (void)hipMemcpyAsync(jobs_d, jobs.data(), sizeof(InvJob) * count, hipMemcpyHostToDevice, stream);
  (void)hipMemsetAsync(fail_d, 0, sizeof(int) * count, stream);
  for (int b = 0; b < count; ++b) (void)hipMemsetAsync(work[b], 0, sizeof(double) * (size_t)N[b] * N[b], stream);
  const unsigned nz = (unsigned)count;
  for (int kb = 0; kb < max_blk; ++kb) {
    const int s0 = (kb + 1) * NB;
    hipLaunchKernelGGL(k_potrf_diag, dim3(1, 1, nz), dim3(64), 0, stream, jobs_d, kb, fail_d);
    if (s0 < max_n) {
      hipLaunchKernelGGL(k_trsm_panel, dim3((max_n - s0 + 63) / 64, 1, nz), dim3(256), 0, stream, jobs_d, kb);
      const int nt = (max_n - s0 + 63) / 64;
      hipLaunchKernelGGL(k_syrk, dim3(nt, nt, nz), dim3(256), 0, stream, jobs_d, kb);
    }
  }
  for (int ib = 0; ib < max_blk; ++ib) {
    hipLaunchKernelGGL(k_trtri_fin, dim3(ib + 1, 1, nz), dim3(NB * NB), 0, stream, jobs_d, ib);
    const int below = max_n - (ib + 1) * NB;
    if (below > 0)
      hipLaunchKernelGGL(k_trtri_upd, dim3((below + 63) / 64, ((ib + 1) * NB + 63) / 64, nz), dim3(256), 0, stream, jobs_d, ib);
  }
  const int nt = (max_n + 63) / 64;
  hipLaunchKernelGGL(k_wtw, dim3(nt, nt, nz), dim3(256), 0, stream, jobs_d);
  std::vector<int> fail(count, 0);
  (void)hipMemcpyAsync(fail.data(), fail_d, sizeof(int) * count, hipMemcpyDeviceToHost, stream);
  (void)hipStreamSynchronize(stream);
  (void)hipFree(LinvAll);
  (void)hipFree(fail_d);
  (void)hipFree(jobs_d);
  for (int b = 0; b < count; ++b) if (fail[b]) return fail[b] + (b << 24);
  return 0;
}

int dense_spd_inverse(hipStream_t stream, double *A, double *work, double *M, int N) {
  return dense_spd_inverse_batched(stream, 1, &A, &work, &M, &N) & 0xffffff;
}

}  // namespace dpgo
