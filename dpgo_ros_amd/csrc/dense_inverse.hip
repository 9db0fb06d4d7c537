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
#include "team_internal.h"

namespace dpgo {

constexpr int NB = 32;
#ifndef DPGO_SOLVE_SBK
#define DPGO_SOLVE_SBK 8
#endif

// one matrix of a batch: the kernels take the batch index from blockIdx.z, so that the many small, dependent steps
// of several inversions (one per agent) share their launches
struct InvJob {
  double *A, *W, *M, *Linv;  // matrix (destroyed), work / triangular inverse, result, per-block inverse diagonal factors
  int N, nblk;
};

// broadcast of lane `l`'s double to the whole wave through two v_readlane_b32 (l is a compile-time constant after
// unrolling): the value arrives in scalar registers, no LDS round trip
__device__ __forceinline__ double lane_bcast(double x, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), l);
  return __hiloint2double(hi, lo);
}

// 1 / sqrt(d): hardware estimate (v_rsq_f64) refined by two Newton steps -- no division, no sqrt expansion on the
// 2 x NB long dependency chain of the diagonal block
__device__ __forceinline__ double rsqrt_nr(double d) {
  double y = __builtin_amdgcn_rsq(d);
  y = y * (1.5 - 0.5 * d * y * y);
  y = y * (1.5 - 0.5 * d * y * y);
  return y;
}

// factor the nb x nb diagonal block at (k0,k0) in place and write its inverse (lower, dense NB x NB
// column-major, zero padded) to Linv.  ONE wave (lane = threadIdx.x & 63), lane i < NB keeps row i of the block in
// registers; `cols` = NB x NB doubles of LDS.
// RIGHT-looking, columns exchanged through LDS: once column k is final (one v_readlane broadcast of the pivot, rsqrt from
// v_rsq_f64 + two Newton steps, one scale) the lanes leave it in LDS and every trailing column j > k takes
//   row[j] -= L[i][k] L[j][k]        (L[j][k]: a broadcast ds_read, the same address in every lane)
// -- NB - 1 - k multiply-adds that do not depend on each other.  The inverse W = L^-1 (lane j owns column j) runs the
// same way from the columns left in LDS: once w[k] is final, w[i] -= L[i][k] w[k] for all i > k.
// History: the left-looking form (each column's sum accumulated serially, operands through v_readlane: ~8000
// instructions) took 30 us, 11 of them in 32 predicated loads that were each waited for on their own.
// Two waves since round 6 (callers: the threads with tid < 128; `idg`: NB doubles, `prog`: one word of LDS the caller zeroed in
// front of its last barrier): wave 0 factors, wave 1 forms the inverse ONE COLUMN BEHIND it -- column k of L and 1 / L[k][k]
// are in LDS when wave 0 raises *prog to k + 1, and that is all step k of the substitution reads.  The two loops ran one after
// the other on one wave (15 us per diagonal block: every block step of every factorisation waits for it); the arithmetic of
// each is unchanged, so L and W are bitwise what they were.
__device__ __forceinline__ void potrf_diag_body(const InvJob &jb_, int kb, int *fail, int z, double *cols, double *idg, int *prog) {
  double *A = jb_.A;
  const int N = jb_.N, k0 = kb * NB, nb = min(NB, N - k0);
  double *Linv = jb_.Linv + (size_t)kb * NB * NB;
  const int lane = threadIdx.x & 63;
  const bool own = lane < NB;
  const int ls = min(lane, NB - 1);  // LDS slot of this lane (lanes >= NB shadow the last one and never store)
  if (((threadIdx.x >> 6) & 1) == 0) {
  // straight-line loads (a predicated load is waited for on its own): every lane reads NB values from clamped, valid
  // addresses, the padding is selected afterwards
  double row[NB];
  {
    const int ll = min(lane, nb - 1);
    double raw[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) raw[j] = A[(size_t)(k0 + min(j, nb - 1)) * N + k0 + ll];
#pragma unroll
    for (int j = 0; j < NB; ++j)
      row[j] = (own && lane < nb && j < nb && j <= lane) ? raw[j] : ((own && j == lane) ? 1.0 : 0.0);
  }
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    double d = lane_bcast(row[k], k);
    if (!(d > 0.0)) { if (lane == 0) fail[z] = k0 + k + 1; d = 1.0; }
    const double inv = rsqrt_nr(d);
    const double lik = (lane == k) ? d * inv : ((lane > k) ? row[k] * inv : 0.0);  // L[i][k]
    row[k] = lik;
    if (own) cols[k * NB + ls] = lik;
    if (lane == 0) idg[k] = inv;
    // (DS operations of one wave execute in order; the fences keep the compiler from moving the reads above the write --
    // and the raise of *prog below behind the column it announces)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_store(prog, k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double lj[NB];
#pragma unroll
    for (int j = k + 1; j < NB; ++j) lj[j] = cols[k * NB + j];
#pragma unroll
    for (int j = k + 1; j < NB; ++j) row[j] -= lik * lj[j];  // (lanes i < j hold unused values there)
  }
  if (own && lane < nb) {
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (j < nb && j <= lane) A[(size_t)(k0 + j) * N + k0 + lane] = row[j];
  }
  return;
  }
  // ---- wave 1: W = L^-1, lane j owns column j; step k waits for column k
  double w[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) w[i] = (i == lane) ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    {
      int spins = 0;
      while (__hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < k + 1) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { if (lane == 0) fail[z] = k0 + k + 1; break; }  // (never seen: a bound, not a path)
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    const double wk = (k >= lane) ? w[k] * idg[k] : 0.0;
    w[k] = wk;
    double li[NB];
#pragma unroll
    for (int i = k + 1; i < NB; ++i) li[i] = cols[k * NB + i];
#pragma unroll
    for (int i = k + 1; i < NB; ++i) w[i] -= li[i] * wk;
  }
  if (own) {
#pragma unroll
    for (int i = 0; i < NB; ++i) Linv[lane * NB + i] = (i < nb && lane < nb && i >= lane) ? w[i] : 0.0;
  }
}

__global__ __launch_bounds__(128) void k_potrf_diag(const InvJob *jobs, int kb, int *fail) {
  const InvJob jb_ = jobs[blockIdx.z];
  if (kb >= jb_.nblk) return;
  __shared__ double cols[NB * NB], idg[NB];
  __shared__ int prog;
  if (threadIdx.x == 0) prog = 0;
  __syncthreads();
  potrf_diag_body(jb_, kb, fail, (int)blockIdx.z, cols, idg, &prog);
}

// Staging of one NB x 64 operand tile (8 elements per thread): straight-line.  `at(t)` returns a CLAMPED, always valid
// address for element t, `ok(t)` whether the element exists, `put(t, v)` stores it (zero where it does not).  A load
// under a predicate is waited for on its own -- eight dependent round trips per tile, which was most of these
// kernels' time; this way all loads of a stage are in flight together.
template <class At, class Ok, class Put>
__device__ __forceinline__ void stage_tile(int tid, At at, Ok ok, Put put) {
  double v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = *at(tid + 256 * q);
#pragma unroll
  for (int q = 0; q < 8; ++q) put(tid + 256 * q, ok(tid + 256 * q) ? v[q] : 0.0);
}
template <class At1, class At2, class Ok1, class Ok2, class Put1, class Put2>
__device__ __forceinline__ void stage_tiles2(int tid, At1 at1, Ok1 ok1, Put1 put1, At2 at2, Ok2 ok2, Put2 put2) {
  double v[8], w[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { v[q] = *at1(tid + 256 * q); w[q] = *at2(tid + 256 * q); }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    put1(tid + 256 * q, ok1(tid + 256 * q) ? v[q] : 0.0);
    put2(tid + 256 * q, ok2(tid + 256 * q) ? w[q] : 0.0);
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
  static_assert(NB * 64 == 8 * 256, "stage_tile: 8 elements per thread");
  stage_tile(tid,  // consecutive threads = consecutive rows of one column: coalesced
             [&](int t) { return A + (size_t)(k0 + min(t >> 6, nb - 1)) * N + min(i0 + (t & 63), N - 1); },
             [&](int t) { return (t >> 6) < nb && i0 + (t & 63) < N; },
             [&](int t, double v) { Rs[t & 63][t >> 6] = v; });
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

// 64x64 output tile C[i][j] += sum_k As[k][i] Bs[k][j], K = NB slab staged in LDS (zero padded to NB), on the matrix
// cores: v_mfma_f64_16x16x4_f64, the one fp64 MFMA shape of gfx950.  256 threads = 4 waves; wave w owns the 32 x 32
// quadrant rows 32 (w & 1), columns 32 (w >> 1) as 2 x 2 blocks of 16 x 16.  Operand / result layout of the
// instruction (cdna_hip_programming.md, "f64 MFMA does NOT use these maps"): A[row = lane & 15][k = lane >> 4],
// B[k = lane >> 4][col = lane & 15], D[row = (lane >> 4) + 4 reg][col = lane & 15].  The instruction's ROW index is
// mapped to the matrix COLUMN j and its column index to the matrix row i, so that the 16 lanes that share a result
// register hold 16 consecutive rows i of one column: 128 contiguous bytes of the column-major matrices per store.
typedef double v4f64_t __attribute__((ext_vector_type(4)));

struct TileAcc {
  v4f64_t c[2][2];  // [u: i-block][v: j-block]
};

__device__ __forceinline__ void tile_zero(TileAcc &t) {
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) t.c[u][v] = v4f64_t{0.0, 0.0, 0.0, 0.0};
}

__device__ __forceinline__ void tile_mac(const double (*As)[65], const double (*Bs)[65], int tid, TileAcc &t) {
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
  const int ib = 32 * (wave & 1), jb = 32 * (wave >> 1);
#pragma unroll
  for (int k = 0; k < NB; k += 4) {
    double a[2], b[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { a[u] = Bs[k + lk][jb + 16 * u + li]; b[u] = As[k + lk][ib + 16 * u + li]; }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) t.c[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[v], b[u], t.c[u][v], 0, 0, 0);
  }
}

// element `reg` of block (u, v) held by this lane: tile row i, tile column j
__device__ __forceinline__ int tile_row(int tid, int u) { return 32 * ((tid >> 6) & 1) + 16 * u + (tid & 15); }
__device__ __forceinline__ int tile_col(int tid, int v, int reg) { return 32 * (tid >> 7) + 16 * v + ((tid & 63) >> 4) + 4 * reg; }

// trailing update: A[i,j] -= sum_k P[i,k] P[j,k], i >= j >= s0 (= k0 + nb), P = columns k0..k0+nb
__global__ __launch_bounds__(256) void k_syrk(const InvJob *jobs, int kb, int *fail) {
  const InvJob jb_ = jobs[blockIdx.z];
  if (kb >= jb_.nblk) return;
  double *A = jb_.A;
  const int N = jb_.N, k0 = kb * NB, nb = min(NB, N - k0), s0 = k0 + nb;
  const int bi = blockIdx.x, bj = blockIdx.y;
  if (bj > bi || s0 + 64 * bi >= N) return;
  __shared__ double As[NB][65], Bs[NB][65];
  __shared__ double potrf_idg[NB];  // the folded factorisation of the next diagonal block (potrf_diag_body)
  __shared__ int potrf_prog;
  const int i0 = s0 + 64 * bi, j0 = s0 + 64 * bj;
  const int tid = threadIdx.x;
  stage_tiles2(tid,
               [&](int t) { return A + (size_t)(k0 + min(t >> 6, nb - 1)) * N + min(i0 + (t & 63), N - 1); },
               [&](int t) { return (t >> 6) < nb && i0 + (t & 63) < N; },
               [&](int t, double v) { As[t >> 6][t & 63] = v; },
               [&](int t) { return A + (size_t)(k0 + min(t >> 6, nb - 1)) * N + min(j0 + (t & 63), N - 1); },
               [&](int t) { return (t >> 6) < nb && j0 + (t & 63) < N; },
               [&](int t, double v) { Bs[t >> 6][t & 63] = v; });
  __syncthreads();
  TileAcc acc;
  tile_zero(acc);
  tile_mac(As, Bs, tid, acc);
  {
    // read-modify-write of the tile, straight-line as well: all 16 reads in flight, then the stores
    double old[2][2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = min(i0 + tile_row(tid, u), N - 1), j = min(j0 + tile_col(tid, v, q), N - 1);
          old[u][v][q] = A[(size_t)j * N + i];
        }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = i0 + tile_row(tid, u), j = j0 + tile_col(tid, v, q);
          if (i < N && j < N && i >= j) A[(size_t)j * N + i] = old[u][v][q] - acc.c[u][v][q];
        }
  }
  if (fail && bi == 0 && bj == 0 && kb + 1 < jb_.nblk) {
    // this workgroup has just finished the tile that holds the next diagonal block (s0, s0): factor it here, under the
    // rest of the trailing update, instead of in a launch of its own on the critical path.  (The tile was written by this
    // workgroup's own threads: a WORKGROUP-scope fence orders it.  The device-scope fence that stood here until round 6
    // wrote the whole L2 back, once per matrix of the batch, while the other workgroups were filling it: trailing updates
    // of 169 and 145 us among ones of 35 in the UPDATE_WEIGHT round's first chain, profiles/experiments/jobs/r06_update_trace.sh)
    if (tid == 0) potrf_prog = 0;
    __threadfence_block();
    __syncthreads();
    if (tid < 128) potrf_diag_body(jb_, kb + 1, fail, (int)blockIdx.z, &As[0][0], potrf_idg, &potrf_prog);  // (the operand tiles are done with)
  }
}

// Trailing update by SUPER-BLOCKS (dense_spd_solve: one large system, the chordal relaxation's 7500^2): a rank-32 update
// of the whole trailing matrix reads and writes it once per block column -- n^3 / (3 NB) x 16 bytes, 14 ms of HBM time
// at n = 7500 -- so the columns beyond the current super-block (SBK block columns) are left alone until the super-block
// is factored and then receive all of its block columns in ONE pass (K = SBK x NB, accumulated in registers); inside
// the super-block every block column still updates the (few) columns up to its end at once.
//   A[i, j] -= sum_{kb0 <= kb < kb0 + nk} P_kb[i, :] P_kb[j, :]^T   for t0 <= j < jhi, i >= j;  tile (0, 0) factors the
//   diagonal block `next_kb` (>= 0) when it is done, as k_syrk does
__global__ __launch_bounds__(256) void k_syrk_sb(const InvJob *jobs, int kb0, int nk, int t0, int jhi, int next_kb, int *fail) {
  const InvJob jb_ = jobs[blockIdx.z];
  double *A = jb_.A;
  const int N = jb_.N;
  const int bi = blockIdx.x, bj = blockIdx.y;
  const int i0 = t0 + 64 * bi, j0 = t0 + 64 * bj;
  if (bj > bi || i0 >= N || j0 >= jhi) return;
  __shared__ double As[NB][65], Bs[NB][65];
  __shared__ double potrf_idg[NB];
  __shared__ int potrf_prog;
  const int tid = threadIdx.x;
  TileAcc acc;
  tile_zero(acc);
  for (int q = 0; q < nk; ++q) {
    const int k0 = (kb0 + q) * NB, nb = min(NB, N - k0);
    if (q > 0) __syncthreads();
    stage_tiles2(tid,
                 [&](int t) { return A + (size_t)(k0 + min(t >> 6, nb - 1)) * N + min(i0 + (t & 63), N - 1); },
                 [&](int t) { return (t >> 6) < nb && i0 + (t & 63) < N; },
                 [&](int t, double v) { As[t >> 6][t & 63] = v; },
                 [&](int t) { return A + (size_t)(k0 + min(t >> 6, nb - 1)) * N + min(j0 + (t & 63), N - 1); },
                 [&](int t) { return (t >> 6) < nb && j0 + (t & 63) < N; },
                 [&](int t, double v) { Bs[t >> 6][t & 63] = v; });
    __syncthreads();
    tile_mac(As, Bs, tid, acc);
  }
  {
    double old[2][2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = min(i0 + tile_row(tid, u), N - 1), j = min(j0 + tile_col(tid, v, q), N - 1);
          old[u][v][q] = A[(size_t)j * N + i];
        }
    const int jend = min(N, jhi);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = i0 + tile_row(tid, u), j = j0 + tile_col(tid, v, q);
          if (i < N && j < jend && i >= j) A[(size_t)j * N + i] = old[u][v][q] - acc.c[u][v][q];
        }
  }
  if (bi == 0 && bj == 0 && next_kb >= 0 && next_kb < jb_.nblk) {
    if (tid == 0) potrf_prog = 0;
    __threadfence_block();  // (see k_syrk)
    __syncthreads();
    if (tid < 128) potrf_diag_body(jb_, next_kb, fail, (int)blockIdx.z, &As[0][0], potrf_idg, &potrf_prog);
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
  const int tid = threadIdx.x;
  stage_tiles2(tid,
               [&](int t) { return L + (size_t)(k0 + min(t >> 6, kn - 1)) * N + min(i0 + (t & 63), N - 1); },  // L[i, k0 + k]
               [&](int t) { return (t >> 6) < kn && i0 + (t & 63) < N; },
               [&](int t, double v) { As[t >> 6][t & 63] = v; },
               [&](int t) { return W + (size_t)min(j0 + (t >> 5), jend - 1) * N + k0 + min(t & 31, kn - 1); },  // W[k0 + k, j]
               [&](int t) { return (t & 31) < kn && j0 + (t >> 5) < jend; },
               [&](int t, double v) { Bs[t & 31][t >> 5] = v; });
  __syncthreads();
  TileAcc acc;
  tile_zero(acc);
  tile_mac(As, Bs, tid, acc);
  {
    double old[2][2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = min(i0 + tile_row(tid, u), N - 1), j = min(j0 + tile_col(tid, v, q), jend - 1);
          old[u][v][q] = W[(size_t)j * N + i];
        }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int i = i0 + tile_row(tid, u), j = j0 + tile_col(tid, v, q);
          if (i < N && j < jend) W[(size_t)j * N + i] = old[u][v][q] + acc.c[u][v][q];
        }
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
  const int tid = threadIdx.x;
  TileAcc acc;
  tile_zero(acc);
  for (int kk = (i0 / NB) * NB; kk < N; kk += NB) {
    __syncthreads();
    stage_tiles2(tid,
                 [&](int t) { return W + (size_t)min(i0 + (t >> 5), N - 1) * N + min(kk + (t & 31), N - 1); },
                 [&](int t) { return kk + (t & 31) < N && i0 + (t >> 5) < N; },
                 [&](int t, double v) { As[t & 31][t >> 5] = v; },
                 [&](int t) { return W + (size_t)min(j0 + (t >> 5), N - 1) * N + min(kk + (t & 31), N - 1); },
                 [&](int t) { return kk + (t & 31) < N && j0 + (t >> 5) < N; },
                 [&](int t, double v) { Bs[t & 31][t >> 5] = v; });
    __syncthreads();
    tile_mac(As, Bs, tid, acc);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + tile_row(tid, u), j = j0 + tile_col(tid, v, q);
        if (i < N && j < N) {
          M[(size_t)j * N + i] = acc.c[u][v][q];
          M[(size_t)i * N + j] = acc.c[u][v][q];
        }
      }
}

int dense_spd_inverse_batched(hipStream_t stream, int count, double *const *A, double *const *work, double *const *M,
                              const int *N, bool work_is_zero) {
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
  // (scratch from the pool of team_internal.h: three hipMalloc / hipFree pairs per call were a tenth of a weight update)
  dpgo_host::DevBuf<double> b_linv;
  dpgo_host::DevBuf<int> b_fail;
  dpgo_host::DevBuf<InvJob> b_jobs;
  if (b_linv.alloc(linv_total) || b_fail.alloc(count) || b_jobs.alloc(count)) return -1;
  double *LinvAll = b_linv.p;
  int *fail_d = b_fail.p;
  InvJob *jobs_d = b_jobs.p;
  {
    size_t off = 0;
    for (int b = 0; b < count; ++b) { jobs[b].Linv = LinvAll + off; off += (size_t)jobs[b].nblk * NB * NB; }
  }
  (void)hipMemcpyAsync(jobs_d, jobs.data(), sizeof(InvJob) * count, hipMemcpyHostToDevice, stream);
  (void)hipMemsetAsync(fail_d, 0, sizeof(int) * count, stream);
  if (!work_is_zero)  // (one memset per matrix: a batch of a hundred subdomain blocks passes an area it has zeroed in one)
    for (int b = 0; b < count; ++b) (void)hipMemsetAsync(work[b], 0, sizeof(double) * (size_t)N[b] * N[b], stream);
  const unsigned nz = (unsigned)count;
  // The diagonal block of step kb + 1 is factored by the trailing update of step kb (k_syrk, tile (0,0)) -- in a batch of
  // FEW matrices, where that takes a launch off a chain nothing else fills (5 agents' M: 63 steps; the 8 Schur complements
  // of an UPDATE_WEIGHT round: 20 us per folded step against 15 + 8 in two launches).  In a batch of many small matrices
  // (the 112 subdomain blocks of the same round) the folded steps measured 40 / 110 / 108 / 26 / 21 / 37 us where the
  // trailing update alone takes 23 / 32 / 30 / 13 / 8 / 10 and the factorisation 16: there it stays a launch of its own
  // (profiles/experiments/jobs/r06_update_fold.sh; DPGO_SYRK_FOLD=0 / 1 forces either form).
  static const int fold_env = std::getenv("DPGO_SYRK_FOLD") ? std::atoi(std::getenv("DPGO_SYRK_FOLD")) : -1;
  const bool fold = fold_env >= 0 ? fold_env != 0 : count <= 32;
  for (int kb = 0; kb < max_blk; ++kb) {
    const int s0 = (kb + 1) * NB;
    if (kb == 0 || !fold) hipLaunchKernelGGL(k_potrf_diag, dim3(1, 1, nz), dim3(128), 0, stream, jobs_d, kb, fail_d);
    if (s0 < max_n) {
      hipLaunchKernelGGL(k_trsm_panel, dim3((max_n - s0 + 63) / 64, 1, nz), dim3(256), 0, stream, jobs_d, kb);
      const int nt = (max_n - s0 + 63) / 64;
      hipLaunchKernelGGL(k_syrk, dim3(nt, nt, nz), dim3(256), 0, stream, jobs_d, kb, fold ? fail_d : (int *)nullptr);
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
  (void)hipStreamSynchronize(stream);  // (the scratch goes back to the pool behind this: nothing queued still reads it)
  for (int b = 0; b < count; ++b) if (fail[b]) return fail[b] + (b << 24);
  return 0;
}

// ---- solve without the inverse: A X^T = B^T for RR right-hand sides from the Cholesky factor alone (a third of the
// arithmetic of the inverse; the chordal initialisation applies its 7500^2 system to 3 right-hand sides exactly once).
// B / Y: RR-vectors per row, [row][RR]; the solution overwrites B, Y is scratch.  Right-looking block substitutions, one
// launch per block step:
//   forward   y_kb = Linv_kb b_kb -> Y;       b_i  -= L[i, kb] y_kb   for the rows i below (in B)
//   backward  x_kb = Linv_kb^T y_kb -> B;     y_j  -= L[kb, j]^T x_kb for the rows j above (in Y)
template <int RR>
__global__ __launch_bounds__(256) void k_solve_fwd(const InvJob *jobs, int kb, double *B, double *Y) {
  const InvJob jb_ = jobs[0];
  const int N = jb_.N, k0 = kb * NB, nb = min(NB, N - k0);
  const double *Linv = jb_.Linv + (size_t)kb * NB * NB;  // [col j][row i] = W[i][j], W = inverse of the diagonal block
  __shared__ double ys[NB][RR], Ws[NB * NB], bs[NB][RR];
  const int tid = threadIdx.x;
  // the block's inverse factor and right-hand sides through LDS, straight-line (loads inside a data-dependent loop
  // are waited for one by one: 14 us per step)
#pragma unroll
  for (int q = 0; q < NB * NB / 256; ++q) Ws[tid + 256 * q] = Linv[tid + 256 * q];
  if (tid < NB * RR) bs[tid / RR][tid % RR] = B[(size_t)(k0 + min(tid / RR, nb - 1)) * RR + tid % RR];
  __syncthreads();
  if (tid < NB * RR) {
    const int i = tid / RR, a = tid - i * RR;
    double s = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) s += (j <= i && j < nb) ? Ws[j * NB + i] * bs[j][a] : 0.0;
    ys[i][a] = (i < nb) ? s : 0.0;
  }
  __syncthreads();
  const int i = k0 + nb + (int)blockIdx.x * 256 + tid;
  if (i < N) {
    double l[NB], acc[RR], old[RR];
#pragma unroll
    for (int k = 0; k < NB; ++k) l[k] = jb_.A[(size_t)(k0 + min(k, nb - 1)) * N + i];
#pragma unroll
    for (int a = 0; a < RR; ++a) { acc[a] = 0; old[a] = B[(size_t)i * RR + a]; }
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
      for (int a = 0; a < RR; ++a) acc[a] += l[k] * ys[k][a];  // (rows k >= nb of ys are zero)
#pragma unroll
    for (int a = 0; a < RR; ++a) B[(size_t)i * RR + a] = old[a] - acc[a];
  }
  // y goes to its own array: the rows of block kb in B are still being read by the other workgroups of this launch
  if (blockIdx.x == gridDim.x - 1 && tid < nb * RR) Y[(size_t)k0 * RR + tid] = ys[tid / RR][tid % RR];
}

template <int RR>
__global__ __launch_bounds__(256) void k_solve_bwd(const InvJob *jobs, int kb, double *Y, double *Xk) {
  const InvJob jb_ = jobs[0];
  const int N = jb_.N, k0 = kb * NB, nb = min(NB, N - k0);
  const double *Linv = jb_.Linv + (size_t)kb * NB * NB;
  __shared__ double xs[NB][RR], Ws[NB * NB], ysb[NB][RR];
  const int tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < NB * NB / 256; ++q) Ws[tid + 256 * q] = Linv[tid + 256 * q];
  if (tid < NB * RR) ysb[tid / RR][tid % RR] = Y[(size_t)(k0 + min(tid / RR, nb - 1)) * RR + tid % RR];
  __syncthreads();
  if (tid < NB * RR) {
    const int j = tid / RR, a = tid - j * RR;
    double s = 0;
#pragma unroll
    for (int i = 0; i < NB; ++i) s += (i >= j && i < nb) ? Ws[j * NB + i] * ysb[i][a] : 0.0;  // (Linv^T y)_j
    xs[j][a] = (j < nb) ? s : 0.0;
  }
  __syncthreads();
  const int c = (int)blockIdx.x * 256 + tid;  // a column (= row of the unknown) above the block
  if (c < k0) {
    double l[NB], acc[RR], old[RR];
    const double *col = jb_.A + (size_t)c * N + k0;  // L[k0 + r][c], r = 0 .. nb-1: contiguous
#pragma unroll
    for (int r = 0; r < NB; ++r) l[r] = col[min(r, nb - 1)];
#pragma unroll
    for (int a = 0; a < RR; ++a) { acc[a] = 0; old[a] = Y[(size_t)c * RR + a]; }
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int a = 0; a < RR; ++a) acc[a] += l[r] * xs[r][a];  // (rows r >= nb of xs are zero)
#pragma unroll
    for (int a = 0; a < RR; ++a) Y[(size_t)c * RR + a] = old[a] - acc[a];
  }
  if (blockIdx.x == gridDim.x - 1 && tid < nb * RR) Xk[(size_t)k0 * RR + tid] = xs[tid / RR][tid % RR];
}

// y of an AUGMENTED system: the caller appended the RR right-hand sides to A as rows N - RR .. N - 1 (with a huge diagonal
// behind them), so the factorisation's own panel solves and trailing updates have carried them along and row N - RR + a
// of the factor is y_a = L^-1 b_a -- the forward substitution without its launches (314 of them, 2 ms, for the chordal
// relaxation's 7500^2 system).  Rows of the appended block get y = 0: the backward substitution then leaves the first
// N - RR unknowns as they should be.
template <int RR>
__global__ void k_aug_extract(const double *A, int N, double *Y) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
#pragma unroll
  for (int a = 0; a < RR; ++a) Y[(size_t)k * RR + a] = (k < N - RR) ? A[(size_t)k * N + (N - RR) + a] : 0.0;
}

template <int RR>
int dense_spd_solve(hipStream_t stream, double *A, int N, double *B, double *Y, bool augmented) {
  InvJob job;
  job.A = A; job.W = nullptr; job.M = nullptr; job.N = N; job.nblk = (N + NB - 1) / NB;
  double *Linv = nullptr;
  int *fail_d = nullptr;
  InvJob *job_d = nullptr;
  if (hipMalloc(&Linv, sizeof(double) * (size_t)job.nblk * NB * NB) != hipSuccess) return -1;
  if (hipMalloc(&fail_d, sizeof(int)) != hipSuccess) { (void)hipFree(Linv); return -1; }
  if (hipMalloc(&job_d, sizeof(InvJob)) != hipSuccess) { (void)hipFree(Linv); (void)hipFree(fail_d); return -1; }
  job.Linv = Linv;
  (void)hipMemcpyAsync(job_d, &job, sizeof(InvJob), hipMemcpyHostToDevice, stream);
  (void)hipMemsetAsync(fail_d, 0, sizeof(int), stream);
  constexpr int SBK = DPGO_SOLVE_SBK;  // block columns per super-block (k_syrk_sb)
  for (int sb0 = 0; sb0 < job.nblk; sb0 += SBK) {
    const int sb1 = std::min(sb0 + SBK, job.nblk), c1 = std::min(N, sb1 * NB);
    for (int kb = sb0; kb < sb1; ++kb) {
      const int s0 = (kb + 1) * NB;
      if (kb == 0) hipLaunchKernelGGL(k_potrf_diag, dim3(1, 1, 1), dim3(128), 0, stream, job_d, kb, fail_d);
      if (s0 >= N) continue;
      const int nt = (N - s0 + 63) / 64;
      hipLaunchKernelGGL(k_trsm_panel, dim3(nt, 1, 1), dim3(256), 0, stream, job_d, kb);
      if (kb + 1 < sb1)  // inside the super-block: this block column alone, onto the columns up to the super-block's end
        hipLaunchKernelGGL(k_syrk_sb, dim3(nt, (c1 - s0 + 63) / 64, 1), dim3(256), 0, stream, job_d, kb, 1, s0, c1, kb + 1, fail_d);
      else               // its last block column: all of them onto everything beyond
        hipLaunchKernelGGL(k_syrk_sb, dim3(nt, nt, 1), dim3(256), 0, stream, job_d, sb0, sb1 - sb0, s0, N, kb + 1, fail_d);
    }
  }
  if (augmented) {
    hipLaunchKernelGGL(k_aug_extract<RR>, dim3((N + 255) / 256), dim3(256), 0, stream, A, N, Y);
  } else {
    for (int kb = 0; kb < job.nblk; ++kb) {
      const int below = N - std::min(N, (kb + 1) * NB);
      hipLaunchKernelGGL(k_solve_fwd<RR>, dim3(std::max(1, (below + 255) / 256)), dim3(256), 0, stream, job_d, kb, B, Y);
    }
  }
  for (int kb = job.nblk - 1; kb >= 0; --kb)
    hipLaunchKernelGGL(k_solve_bwd<RR>, dim3(std::max(1, (kb * NB + 255) / 256)), dim3(256), 0, stream, job_d, kb, Y, B);
  int fail = 0;
  (void)hipMemcpyAsync(&fail, fail_d, sizeof(int), hipMemcpyDeviceToHost, stream);
  (void)hipStreamSynchronize(stream);
  (void)hipFree(Linv); (void)hipFree(fail_d); (void)hipFree(job_d);
  return fail;
}
template int dense_spd_solve<3>(hipStream_t, double *, int, double *, double *, bool);

int dense_spd_inverse(hipStream_t stream, double *A, double *work, double *M, int N) {
  return dense_spd_inverse_batched(stream, 1, &A, &work, &M, &N) & 0xffffff;
}

}  // namespace dpgo
