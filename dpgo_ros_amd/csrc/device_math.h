// device_math.h -- per-pose lifted-SE(3) manifold arithmetic for gfx950 (fp64, registers only).
//
// Serves SURVEY 8a row a5 (LiftedPose / lifted SE manifold ops used at
// src/PGOAgentROS.cpp:1420-1422,1463-1466): tangent projection, QF retraction, polar projection.
// Everything is templated on the relaxation rank R so that a pose (R x 4 doubles) stays in VGPRs.
#pragma once
#include <hip/hip_runtime.h>

namespace dpgo {

typedef double v2d_t __attribute__((ext_vector_type(2)));

// p[i] as a load from / store to GLOBAL memory: gp(p)[i] (kernel_common.h says why; never for LDS)
template <class T>
__device__ __forceinline__ __attribute__((address_space(1))) T *gp(T *p) { return (__attribute__((address_space(1))) T *)p; }

// 3x3 symmetric eigen-decomposition by cyclic Jacobi.  S row-major; V columns = eigenvectors.
__device__ __forceinline__ void sym3_eig(const double S[9], double w[3], double V[9]) {
  double A[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { A[i] = S[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    const double dia = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-32 * dia) break;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        const double apq = A[3 * p + q];
        if (apq != 0.0) {
          const double theta = (A[4 * q] - A[4 * p]) / (2.0 * apq);
          const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const double akp = A[3 * k + p], akq = A[3 * k + q];
            A[3 * k + p] = c * akp - s * akq;
            A[3 * k + q] = s * akp + c * akq;
          }
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const double apk = A[3 * p + k], aqk = A[3 * q + k];
            A[3 * p + k] = c * apk - s * aqk;
            A[3 * q + k] = s * apk + c * aqk;
          }
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const double vkp = V[3 * k + p], vkq = V[3 * k + q];
            V[3 * k + p] = c * vkp - s * vkq;
            V[3 * k + q] = s * vkp + c * vkq;
          }
        }
      }
    }
  }
  w[0] = A[0]; w[1] = A[4]; w[2] = A[8];
}

// Gram matrix G = A^T A of the R x 3 block (symmetric, row-major 3x3)
template <int R>
__device__ __forceinline__ void gram3(const double *A, double S[9]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 3; ++j) {
      double s = 0;
#pragma unroll
      for (int a = 0; a < R; ++a) s += A[i * R + a] * A[j * R + a];
      S[3 * i + j] = s;
      S[3 * j + i] = s;
    }
}

// polar factor through the symmetric eigen-decomposition:  A (A^T A)^{-1/2}   (any full-rank A)
template <int R>
__device__ __forceinline__ void polar_eig_inplace(double *A) {
  double S[9], w[3], V[9], M[9];
  gram3<R>(A, S);
  sym3_eig(S, w, V);
  double iw[3] = {1.0 / sqrt(w[0]), 1.0 / sqrt(w[1]), 1.0 / sqrt(w[2])};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) s += V[3 * i + k] * V[3 * j + k] * iw[k];
      M[3 * i + j] = s;
    }
  double T[3 * R];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int a = 0; a < R; ++a) {
      double s = 0;
#pragma unroll
      for (int i = 0; i < 3; ++i) s += A[i * R + a] * M[3 * i + j];
      T[j * R + a] = s;
    }
#pragma unroll
  for (int i = 0; i < 3 * R; ++i) A[i] = T[i];
}

// polar factor of the R x 3 block A (column-major, A[c*R + a]) in place.
// The Nesterov sequences only ever project points that are close to the manifold (combinations of
// neighbouring Stiefel points), so the fast path is the Newton-Schulz iteration
//   A <- A (3 I - A^T A) / 2        (quadratic: |I - A^T A| -> 3/4 |I - A^T A|^2),
// pure FMAs with no divide / square root / rotation chain (a per-lane fp64 Jacobi costs ~3 us of
// dependent latency on CDNA4).  Anything further from the manifold than |I - A^T A|_F^2 = 0.1 is first scaled by
// 1 / sqrt(|A^T A|_inf) (>= the largest singular value), which puts every singular value into (0, 1]: from there the same iteration raises them monotonically to 1
// (sigma <- sigma (3 - sigma^2) / 2), a few linear steps and then the quadratic tail.  EIG = true (the stand-alone
// projection entry point, arbitrary input) takes the symmetric eigen-decomposition for that case instead; the solver
// kernels instantiate the compact form: three inlined copies of the eigen path were 20 KB of the step kernel's 44 KB,
// and the instruction cache holds 64 KB for two CUs.
template <int R, bool EIG = false>
__device__ __forceinline__ void polar_inplace(double *A) {
  double S[9];
  gram3<R>(A, S);
  double e0 = 1.0 - S[0], e1 = 1.0 - S[4], e2 = 1.0 - S[8];
  double dev = e0 * e0 + e1 * e1 + e2 * e2 + 2.0 * (S[1] * S[1] + S[2] * S[2] + S[5] * S[5]);
  int max_steps = 8;
  if (!(dev < 0.1)) {
    if (EIG) {
      polar_eig_inplace<R>(A);
      return;
    }
    const double r0 = fabs(S[0]) + fabs(S[1]) + fabs(S[2]), r1 = fabs(S[3]) + fabs(S[4]) + fabs(S[5]),
                 r2 = fabs(S[6]) + fabs(S[7]) + fabs(S[8]);
    const double sc2 = 1.0 / fmax(r0, fmax(r1, r2)), sc = sqrt(sc2);
#pragma unroll
    for (int i = 0; i < 3 * R; ++i) A[i] *= sc;
#pragma unroll
    for (int i = 0; i < 9; ++i) S[i] *= sc2;
    e0 = 1.0 - S[0]; e1 = 1.0 - S[4]; e2 = 1.0 - S[8];
    dev = e0 * e0 + e1 * e1 + e2 * e2 + 2.0 * (S[1] * S[1] + S[2] * S[2] + S[5] * S[5]);
    max_steps = 120;
  }
  // quadratic convergence: stop once |I - A^T A|_F^2 is at round-off (1e-31 ~ (3e-16)^2); at most 8 steps
  // (|E|_F: 0.32 -> 7.5e-2 -> 4e-3 -> 1.3e-5 -> 1.3e-10 -> 1e-20).  Points that are already on the manifold
  // (V = proj(V), late iterations) take 0 or 1 step instead of a fixed seven.
  double d = dev;
#pragma unroll 1
  for (int it = 0; it < max_steps && d > 1e-31; ++it) {
    double T[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) T[i] = -0.5 * S[i];
    T[0] += 1.5; T[4] += 1.5; T[8] += 1.5;
    double B[3 * R];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int a = 0; a < R; ++a) B[j * R + a] = A[a] * T[j] + A[R + a] * T[3 + j] + A[2 * R + a] * T[6 + j];
#pragma unroll
    for (int i = 0; i < 3 * R; ++i) A[i] = B[i];
    // E' = (3 E^2 + E^3) / 4  =>  |E'|_F^2 <= d^2: a step taken from d <= 3e-16 needs no check
    if (d * d <= 1e-31) break;
    gram3<R>(A, S);
    const double f0 = 1.0 - S[0], f1 = 1.0 - S[4], f2 = 1.0 - S[8];
    d = f0 * f0 + f1 * f1 + f2 * f2 + 2.0 * (S[1] * S[1] + S[2] * S[2] + S[5] * S[5]);
  }
}

// Q factor (positive diagonal R) of the R x 3 block A in place.
// The retraction is applied to Y + eta with Y on the manifold and eta tangent, i.e. to a block whose Gram matrix is
// I + eta^T eta: close to the identity whenever the step is small (the steady state of the iteration).  There the
// factor comes from the Cholesky factor of the Gram matrix (A = Q R  <=>  A^T A = R^T R, same positive-diagonal R):
// 6 short independent dot products, 3 reciprocal square roots and one triangular solve per row -- no column-by-column
// dependency chain and no divisions; its orthogonality error is eps * cond(A)^2, hence the guard.  Blocks further from
// orthonormal go through modified Gram-Schmidt (error eps * cond(A)).
template <int R>
__device__ __forceinline__ void qf_inplace(double *A) {
  double g00 = 0, g01 = 0, g02 = 0, g11 = 0, g12 = 0, g22 = 0;
#pragma unroll
  for (int a = 0; a < R; ++a) {
    g00 += A[a] * A[a]; g01 += A[a] * A[R + a]; g02 += A[a] * A[2 * R + a];
    g11 += A[R + a] * A[R + a]; g12 += A[R + a] * A[2 * R + a]; g22 += A[2 * R + a] * A[2 * R + a];
  }
  const double e0 = 1.0 - g00, e1 = 1.0 - g11, e2 = 1.0 - g22;
  const double dev = e0 * e0 + e1 * e1 + e2 * e2 + 2.0 * (g01 * g01 + g02 * g02 + g12 * g12);
  if (dev < 0.05) {
    const double i0 = rsqrt(g00);
    const double r01 = g01 * i0, r02 = g02 * i0;
    const double i1 = rsqrt(g11 - r01 * r01);
    const double r12 = (g12 - r01 * r02) * i1;
    const double i2 = rsqrt(g22 - r02 * r02 - r12 * r12);
#pragma unroll
    for (int a = 0; a < R; ++a) {
      const double q0 = A[a] * i0;
      const double q1 = (A[R + a] - r01 * q0) * i1;
      const double q2 = (A[2 * R + a] - r02 * q0 - r12 * q1) * i2;
      A[a] = q0; A[R + a] = q1; A[2 * R + a] = q2;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i < j) {
        double s = 0;
#pragma unroll
        for (int a = 0; a < R; ++a) s += A[i * R + a] * A[j * R + a];
#pragma unroll
        for (int a = 0; a < R; ++a) A[j * R + a] -= s * A[i * R + a];
      }
    }
    double nn = 0;
#pragma unroll
    for (int a = 0; a < R; ++a) nn += A[j * R + a] * A[j * R + a];
    nn = sqrt(nn);
#pragma unroll
    for (int a = 0; a < R; ++a) A[j * R + a] /= nn;
  }
}

// W <- W - Y sym(Y^T W) on the R x 3 rotation block (translation column untouched)
template <int R>
__device__ __forceinline__ void tangent_inplace(const double *Y, double *W) {
  double S[9];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      double s = 0;
#pragma unroll
      for (int a = 0; a < R; ++a) s += Y[p * R + a] * W[q * R + a];
      S[3 * p + q] = s;
    }
  double T[3 * R];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int a = 0; a < R; ++a) {
      double s = W[q * R + a];
#pragma unroll
      for (int p = 0; p < 3; ++p) s -= Y[p * R + a] * 0.5 * (S[3 * p + q] + S[3 * q + p]);
      T[q * R + a] = s;
    }
#pragma unroll
  for (int i = 0; i < 3 * R; ++i) W[i] = T[i];
}

// row `a` of  W - Y sym(Y^T W): Yp/Wp point at full R x 3 blocks (LDS or global), returns 3 values
template <int R>
__device__ __forceinline__ void tangent_row(const double *Yp, const double *Wp, int a, double out[3]) {
  double S[9];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      double s = 0;
#pragma unroll
      for (int b = 0; b < R; ++b) s += Yp[p * R + b] * Wp[q * R + b];
      S[3 * p + q] = s;
    }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    double s = Wp[q * R + a];
#pragma unroll
    for (int p = 0; p < 3; ++p) s -= Yp[p * R + a] * 0.5 * (S[3 * p + q] + S[3 * q + p]);
    out[q] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// Lane-parallel forms of the three routines above: ONE pose on a group of 16 lanes of one wave (sub-lane s = 0 .. 15), the
// pose in LDS.  Every scalar is produced by the SAME expression as in the serial routine, evaluated by one lane -- the
// Gram entries each keep their sequential sum over the rows, the entries of a product each their own three-term
// expression -- so the results are bitwise those of the serial code; what changes is that the six Gram sums run side by
// side instead of one after the other (R instead of 6R dependent multiply-adds) and the 3R entries of a product on 3R
// lanes.  A lane group executes its LDS operations in order (one wave); the fences only pin the compiler.  Groups of one
// wave may diverge from each other (their loops end at different trip counts): each touches its own LDS area.
// The rare branches (a block far from the manifold) fall back to the serial routine on sub-lane 0.
__device__ __forceinline__ void lanes_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// S (9 doubles of LDS) = A^T A of the R x 3 block A (LDS)
template <int R>
__device__ __forceinline__ void gram3_lanes(const double *A, double *S, int s) {
  if (s < 6) {
    const int i = (s < 3) ? 0 : ((s < 5) ? 1 : 2), j = (s < 3) ? s : ((s < 5) ? s - 2 : 2);
    double acc = 0;
#pragma unroll
    for (int a = 0; a < R; ++a) acc += A[i * R + a] * A[j * R + a];
    S[3 * i + j] = acc;
    S[3 * j + i] = acc;
  }
  lanes_sync();
}

template <int R>
__device__ __forceinline__ void polar_lanes(double *A, double *S, int s) {
  gram3_lanes<R>(A, S, s);
  double e0 = 1.0 - S[0], e1 = 1.0 - S[4], e2 = 1.0 - S[8];
  double dev = e0 * e0 + e1 * e1 + e2 * e2 + 2.0 * (S[1] * S[1] + S[2] * S[2] + S[5] * S[5]);
  if (!(dev < 0.1)) {
    lanes_sync();
    if (s == 0) {
      double a[3 * R];
#pragma unroll
      for (int i = 0; i < 3 * R; ++i) a[i] = A[i];
      polar_inplace<R>(a);
#pragma unroll
      for (int i = 0; i < 3 * R; ++i) A[i] = a[i];
    }
    lanes_sync();
    return;
  }
  double d = dev;
  const int j = (s < 3 * R) ? s / R : 0, a = (s < 3 * R) ? s - j * R : 0;
#pragma unroll 1
  for (int it = 0; it < 8 && d > 1e-31; ++it) {
    double T[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) T[k] = -0.5 * S[3 * k + j];
    if (j == 0) T[0] += 1.5;
    if (j == 1) T[1] += 1.5;
    if (j == 2) T[2] += 1.5;
    const double b = A[a] * T[0] + A[R + a] * T[1] + A[2 * R + a] * T[2];
    lanes_sync();
    if (s < 3 * R) A[s] = b;
    lanes_sync();
    if (d * d <= 1e-31) break;
    gram3_lanes<R>(A, S, s);
    const double f0 = 1.0 - S[0], f1 = 1.0 - S[4], f2 = 1.0 - S[8];
    d = f0 * f0 + f1 * f1 + f2 * f2 + 2.0 * (S[1] * S[1] + S[2] * S[2] + S[5] * S[5]);
  }
}

// Q factor of the R x 3 block A (LDS) in place; G: 6 doubles of LDS
template <int R>
__device__ __forceinline__ void qf_lanes(double *A, double *G, int s) {
  if (s < 6) {
    const int i = (s < 3) ? 0 : ((s < 5) ? 1 : 2), j = (s < 3) ? s : ((s < 5) ? s - 2 : 2);
    double acc = 0;
#pragma unroll
    for (int a = 0; a < R; ++a) acc += A[i * R + a] * A[j * R + a];
    G[s] = acc;  // g00 g01 g02 g11 g12 g22
  }
  lanes_sync();
  const double g00 = G[0], g01 = G[1], g02 = G[2], g11 = G[3], g12 = G[4], g22 = G[5];
  const double e0 = 1.0 - g00, e1 = 1.0 - g11, e2 = 1.0 - g22;
  const double dev = e0 * e0 + e1 * e1 + e2 * e2 + 2.0 * (g01 * g01 + g02 * g02 + g12 * g12);
  if (dev < 0.05) {
    const double i0 = rsqrt(g00);
    const double r01 = g01 * i0, r02 = g02 * i0;
    const double i1 = rsqrt(g11 - r01 * r01);
    const double r12 = (g12 - r01 * r02) * i1;
    const double i2 = rsqrt(g22 - r02 * r02 - r12 * r12);
    if (s < R) {
      const int a = s;
      const double q0 = A[a] * i0;
      const double q1 = (A[R + a] - r01 * q0) * i1;
      const double q2 = (A[2 * R + a] - r02 * q0 - r12 * q1) * i2;
      A[a] = q0; A[R + a] = q1; A[2 * R + a] = q2;
    }
    lanes_sync();
    return;
  }
  if (s == 0) {
    double a[3 * R];
#pragma unroll
    for (int i = 0; i < 3 * R; ++i) a[i] = A[i];
    qf_inplace<R>(a);
#pragma unroll
    for (int i = 0; i < 3 * R; ++i) A[i] = a[i];
  }
  lanes_sync();
}

// W <- W - Y sym(Y^T W) on the rotation block; Y, W: R x 3 blocks in LDS, S: 9 doubles of LDS
template <int R>
__device__ __forceinline__ void tangent_lanes(const double *Y, double *W, double *S, int s) {
  if (s < 9) {
    const int p = s / 3, q = s - 3 * p;
    double acc = 0;
#pragma unroll
    for (int a = 0; a < R; ++a) acc += Y[p * R + a] * W[q * R + a];
    S[3 * p + q] = acc;
  }
  lanes_sync();
  double t = 0;
  if (s < 3 * R) {
    const int q = s / R, a = s - q * R;
    t = W[q * R + a];
#pragma unroll
    for (int p = 0; p < 3; ++p) t -= Y[p * R + a] * 0.5 * (S[3 * p + q] + S[3 * q + p]);
  }
  lanes_sync();
  if (s < 3 * R) W[s] = t;
  lanes_sync();
}

// wave-wide sum, every lane receives the result (fixed order -> bitwise reproducible).  DPP path: quad swaps, half-row
// and row mirrors, then the four row totals read back with v_readlane and added in row order -- ~0.1 us where a
// ds_bpermute xor butterfly costs 0.6 with one wave per SIMD (the sparse kernels run one 64-thread workgroup per CU and
// end their dependent chains with two to four of these sums).  The result is wave-uniform.  Call with all 64 lanes
// active (every call site is reached by whole waves).
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double x) {
  x += dpp_move<0xB1>(x);   // quad_perm [1,0,3,2]
  x += dpp_move<0x4E>(x);   // quad_perm [2,3,0,1]
  x += dpp_move<0x141>(x);  // row_half_mirror
  x += dpp_move<0x140>(x);  // row_mirror: every lane of a 16-lane row holds the row total
  const int lo = __double2loint(x), hi = __double2hiint(x);
  double r[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    r[q] = __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * q), __builtin_amdgcn_readlane(lo, 16 * q));
  return (r[0] + r[1]) + (r[2] + r[3]);
}

// sum of a short global array of partials, identical order in every wave that calls it.  The loads of a
// lane are issued as one batch (a runtime-bounded loop would wait for each cold load in turn).
__device__ __forceinline__ double sum_partials(const double *p, int count, int stride, int lane) {
  double s = 0;
  for (int base = 0; base < count; base += 512) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + lane + 64 * u;
      v[u] = (i < count) ? gp(p)[(size_t)i * stride] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  return wave_sum(s);
}

// two adjacent partials per entry (p[i*stride], p[i*stride + 1]) with one 16-byte load each
__device__ __forceinline__ void sum_partials2(const double *p, int count, int stride, int lane, double &s0, double &s1) {
  double a = 0, b = 0;
  for (int base = 0; base < count; base += 512) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + lane + 64 * u;
      if (i < count) { const v2d_t t = *(const __attribute__((address_space(1))) v2d_t *)(p + (size_t)i * stride); v[u] = make_double2(t.x, t.y); } else v[u] = make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a += v[u].x; b += v[u].y; }
  }
  s0 = wave_sum(a);
  s1 = wave_sum(b);
}

}  // namespace dpgo
