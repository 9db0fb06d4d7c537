/* orc_core.c -- CPU oracle: manifold ops, connection Laplacian, sparse Cholesky, RTR/RGD.
 * TEST INFRASTRUCTURE ONLY (see dpgo_oracle.h: parity unpinned).
 *
 * Follows (bodies are external to /root/reference, so each block cites
 * the call site that proves the dependency plus the published algorithm it restates):
 *   a2  PoseGraph data matrices Q / G / preconditioner   (src/PGOAgentROS.cpp:1351, :237)
 *   a3  QuadraticProblem f / EucGrad / EucHessianEta / PreConditioner (src/PGOAgentROS.cpp:169-172)
 *   a4  QuadraticOptimizer RTR (Steihaug tCG) | RGD     (src/PGOAgentROSNode.cpp:85,90,96-100)
 *   a5  lifted SE manifold: tangent projection, QF retraction, polar projection
 *                                                       (src/PGOAgentROS.cpp:1420-1422,1463-1466)
 */
#include "orc_internal.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

double orc_dot(const double *a, const double *b, size_t n) {
  double s = 0;
  for (size_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

/* ------------------------------------------------------------------ 3x3 symmetric eigen (Jacobi) */
static void sym3_eig(const double S[9], double w[3], double V[9]) {
  double A[9];
  memcpy(A, S, sizeof A);
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
    double dia = A[0] * A[0] + A[4] * A[4] + A[8] * A[8];
    if (off <= 1e-32 * dia) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double apq = A[3 * p + q];
        if (apq == 0.0) continue;
        double theta = (A[4 * q] - A[4 * p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { /* A <- A J */
          double akp = A[3 * k + p], akq = A[3 * k + q];
          A[3 * k + p] = c * akp - s * akq;
          A[3 * k + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) { /* A <- J^T A */
          double apk = A[3 * p + k], aqk = A[3 * q + k];
          A[3 * p + k] = c * apk - s * aqk;
          A[3 * q + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          double vkp = V[3 * k + p], vkq = V[3 * k + q];
          V[3 * k + p] = c * vkp - s * vkq;
          V[3 * k + q] = s * vkp + c * vkq;
        }
      }
  }
  w[0] = A[0]; w[1] = A[4]; w[2] = A[8];
}

/* polar factor U V^T of the r x 3 column-major matrix A:  A (A^T A)^{-1/2} */
void orc_project_stiefel(const double *A, int r, double *out) {
  double S[9], w[3], V[9], M[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int a = 0; a < r; ++a) s += A[i * r + a] * A[j * r + a];
      S[3 * i + j] = s;
    }
  sym3_eig(S, w, V);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += V[3 * i + k] * V[3 * j + k] / sqrt(w[k]);
      M[3 * i + j] = s;
    }
  double tmp[3 * 16];
  for (int j = 0; j < 3; ++j)
    for (int a = 0; a < r; ++a) {
      double s = 0;
      for (int i = 0; i < 3; ++i) s += A[i * r + a] * M[3 * i + j];
      tmp[j * r + a] = s;
    }
  memcpy(out, tmp, sizeof(double) * 3 * r);
}

/* nearest rotation to the 3x3 column-major A */
void orc_project_rotation(const double *A, double *out) {
  double S[9], w[3], V[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int a = 0; a < 3; ++a) s += A[i * 3 + a] * A[j * 3 + a];
      S[3 * i + j] = s;
    }
  sym3_eig(S, w, V);
  double det = A[0] * (A[4] * A[8] - A[7] * A[5]) - A[3] * (A[1] * A[8] - A[7] * A[2]) +
               A[6] * (A[1] * A[5] - A[4] * A[2]);
  int kmin = 0;
  if (w[1] < w[kmin]) kmin = 1;
  if (w[2] < w[kmin]) kmin = 2;
  double M[9], tmp[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) {
        double sg = (k == kmin && det < 0) ? -1.0 : 1.0;
        s += sg * V[3 * i + k] * V[3 * j + k] / sqrt(w[k]);
      }
      M[3 * i + j] = s;
    }
  for (int j = 0; j < 3; ++j)
    for (int a = 0; a < 3; ++a) {
      double s = 0;
      for (int i = 0; i < 3; ++i) s += A[i * 3 + a] * M[3 * i + j];
      tmp[j * 3 + a] = s;
    }
  memcpy(out, tmp, sizeof tmp);
}

/* Q factor (positive diagonal R) of Y + eta, r x 3, modified Gram-Schmidt */
void orc_retract_qf(const double *Y, const double *eta, int r, double *out) {
  double A[3 * 16];
  for (int i = 0; i < 3 * r; ++i) A[i] = Y[i] + eta[i];
  for (int j = 0; j < 3; ++j) {
    for (int i = 0; i < j; ++i) {
      double s = 0;
      for (int a = 0; a < r; ++a) s += A[i * r + a] * A[j * r + a];
      for (int a = 0; a < r; ++a) A[j * r + a] -= s * A[i * r + a];
    }
    double nn = 0;
    for (int a = 0; a < r; ++a) nn += A[j * r + a] * A[j * r + a];
    nn = sqrt(nn);
    for (int a = 0; a < r; ++a) A[j * r + a] /= nn;
  }
  memcpy(out, A, sizeof(double) * 3 * r);
}

void orc_project_manifold(const double *Xin, int r, int n, double *Xout) {
  for (int i = 0; i < n; ++i) {
    orc_project_stiefel(Xin + (size_t)i * 4 * r, r, Xout + (size_t)i * 4 * r);
    for (int a = 0; a < r; ++a) Xout[((size_t)i * 4 + 3) * r + a] = Xin[((size_t)i * 4 + 3) * r + a];
  }
}

/* V - Y sym(Y^T V) on rotation blocks, identity on translations */
void orc_tangent_project(const double *X, const double *V, int r, int n, double *out) {
  for (int i = 0; i < n; ++i) {
    const double *Y = X + (size_t)i * 4 * r, *W = V + (size_t)i * 4 * r;
    double *o = out + (size_t)i * 4 * r;
    double S[9];
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q) {
        double s = 0;
        for (int a = 0; a < r; ++a) s += Y[p * r + a] * W[q * r + a];
        S[3 * p + q] = s;
      }
    double tmp[4 * 16];
    for (int q = 0; q < 3; ++q)
      for (int a = 0; a < r; ++a) {
        double s = W[q * r + a];
        for (int p = 0; p < 3; ++p) s -= Y[p * r + a] * 0.5 * (S[3 * p + q] + S[3 * q + p]);
        tmp[q * r + a] = s;
      }
    for (int a = 0; a < r; ++a) tmp[3 * r + a] = W[3 * r + a];
    memcpy(o, tmp, sizeof(double) * 4 * r);
  }
}

void orc_retract(const double *X, const double *eta, int r, int n, double *out) {
  for (int i = 0; i < n; ++i) {
    size_t o = (size_t)i * 4 * r;
    orc_retract_qf(X + o, eta + o, r, out + o);
    for (int a = 0; a < r; ++a) out[o + 3 * r + a] = X[o + 3 * r + a] + eta[o + 3 * r + a];
  }
}

/* ------------------------------------------------------------------ connection Laplacian blocks */
/* 4x4 col-major blocks of one edge:  TO = T Omega = [kR, tau t; 0, tau],
 * TOT = T Omega T^T = [kI + tau t t^T, tau t; tau t^T, tau], Om = diag(k,k,k,tau); all x weight. */
void orc_edge_blocks(const orc_meas_t *m, double TO[16], double TOT[16], double Om[16]) {
  double w = m->weight, k = m->kappa, tau = m->tau;
  memset(TO, 0, 16 * sizeof(double));
  memset(TOT, 0, 16 * sizeof(double));
  memset(Om, 0, 16 * sizeof(double));
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) {
      TO[a + 4 * b] = w * k * m->R[3 * a + b];
      TOT[a + 4 * b] = w * ((a == b ? k : 0.0) + tau * (m->t[a] * m->t[b]));
    }
    TO[a + 4 * 3] = w * tau * m->t[a];
    TOT[a + 4 * 3] = w * tau * m->t[a];
    TOT[3 + 4 * a] = w * tau * m->t[a];
    Om[a + 4 * a] = w * k;
  }
  TO[15] = w * tau; TOT[15] = w * tau; Om[15] = w * tau;
}

static int trip_cmp(const void *a, const void *b) {
  const orc_trip_t *x = (const orc_trip_t *)a, *y = (const orc_trip_t *)b;
  if (x->row != y->row) return x->row < y->row ? -1 : 1;
  if (x->col != y->col) return x->col < y->col ? -1 : 1;
  return 0;
}

typedef struct { orc_trip_t t; int k; } orc_keyed_t;
static int keyed_cmp(const void *a, const void *b) {
  const orc_keyed_t *x = (const orc_keyed_t *)a, *y = (const orc_keyed_t *)b;
  int c = trip_cmp(&x->t, &y->t);
  if (c) return c;
  return x->k < y->k ? -1 : (x->k > y->k);
}

/* duplicates (row,col) are merged in insertion order, so the summation order is deterministic */
void orc_bsr_from_triplets(orc_trip_t *t, int nt, int n, orc_bsr_t *Q) {
  orc_keyed_t *kt = (orc_keyed_t *)malloc(sizeof(orc_keyed_t) * (nt > 0 ? nt : 1));
  for (int i = 0; i < nt; ++i) { kt[i].t = t[i]; kt[i].k = i; }
  qsort(kt, nt, sizeof(orc_keyed_t), keyed_cmp);
  int nb = 0;
  for (int i = 0; i < nt; ++i)
    if (i == 0 || trip_cmp(&kt[i].t, &kt[i - 1].t) != 0) ++nb;
  Q->n = n; Q->nb = nb;
  Q->rowptr = (int *)calloc(n + 1, sizeof(int));
  Q->col = (int *)malloc(sizeof(int) * (nb > 0 ? nb : 1));
  Q->val = (double *)calloc((size_t)16 * (nb > 0 ? nb : 1), sizeof(double));
  int b = -1;
  for (int i = 0; i < nt; ++i) {
    if (i == 0 || trip_cmp(&kt[i].t, &kt[i - 1].t) != 0) {
      ++b;
      Q->col[b] = kt[i].t.col;
      Q->rowptr[kt[i].t.row + 1]++;
    }
    for (int e = 0; e < 16; ++e) Q->val[(size_t)16 * b + e] += kt[i].t.v[e];
  }
  for (int i = 0; i < n; ++i) Q->rowptr[i + 1] += Q->rowptr[i];
  free(kt);
}

void orc_bsr_free(orc_bsr_t *Q) {
  free(Q->rowptr); free(Q->col); free(Q->val);
  memset(Q, 0, sizeof *Q);
}

void orc_bsr_mult(const orc_bsr_t *Q, const double *X, int r, double *out) {
  for (int j = 0; j < Q->n; ++j) {
    double acc[4 * 16];
    for (int e = 0; e < 4 * r; ++e) acc[e] = 0;
    for (int p = Q->rowptr[j]; p < Q->rowptr[j + 1]; ++p) {
      const double *B = Q->val + (size_t)16 * p;
      const double *Xi = X + (size_t)Q->col[p] * 4 * r;
      for (int c = 0; c < 4; ++c)
        for (int cp = 0; cp < 4; ++cp) {
          double b = B[cp + 4 * c];
          if (b == 0.0) continue;
          for (int a = 0; a < r; ++a) acc[c * r + a] += Xi[cp * r + a] * b;
        }
    }
    memcpy(out + (size_t)j * 4 * r, acc, sizeof(double) * 4 * r);
  }
}

/* ------------------------------------------------------------------ sparse Cholesky (up-looking) */
/* exact minimum-degree ordering on the pose graph (elimination-graph form; n is a few thousand) */
static void min_degree_order(int n, const int *rowptr, const int *col, int *order) {
  int **adj = (int **)malloc(sizeof(int *) * n);
  int *deg = (int *)calloc(n, sizeof(int)), *cap = (int *)malloc(sizeof(int) * n);
  char *dead = (char *)calloc(n, 1);
  int *mark = (int *)malloc(sizeof(int) * n);
  for (int i = 0; i < n; ++i) {
    mark[i] = -1;
    cap[i] = rowptr[i + 1] - rowptr[i] + 4;
    adj[i] = (int *)malloc(sizeof(int) * cap[i]);
    for (int p = rowptr[i]; p < rowptr[i + 1]; ++p)
      if (col[p] != i) adj[i][deg[i]++] = col[p];
  }
  for (int step = 0; step < n; ++step) {
    int v = -1;
    for (int i = 0; i < n; ++i)
      if (!dead[i] && (v < 0 || deg[i] < deg[v])) v = i;
    order[step] = v;
    dead[v] = 1;
    int nv = deg[v];
    const int *nb = adj[v];
    for (int a = 0; a < nv; ++a) {
      int u = nb[a];
      int m = 0;
      for (int q = 0; q < deg[u]; ++q) /* drop v from adj[u] */
        if (adj[u][q] != v) adj[u][m++] = adj[u][q];
      deg[u] = m;
      for (int q = 0; q < deg[u]; ++q) mark[adj[u][q]] = u;
      mark[u] = u;
      for (int b = 0; b < nv; ++b) { /* clique among the neighbours of v */
        int w = nb[b];
        if (mark[w] == u) continue;
        if (deg[u] == cap[u]) { cap[u] = cap[u] * 2 + 4; adj[u] = (int *)realloc(adj[u], sizeof(int) * cap[u]); }
        adj[u][deg[u]++] = w;
        mark[w] = u;
      }
      for (int q = 0; q < deg[u]; ++q) mark[adj[u][q]] = -1;
      mark[u] = -1;
    }
  }
  for (int i = 0; i < n; ++i) free(adj[i]);
  free(adj); free(deg); free(cap); free(dead); free(mark);
}

int orc_chol_factor(const orc_bsr_t *Q, double shift, orc_chol_t *C) {
  int n = Q->n, N = 4 * n;
  memset(C, 0, sizeof *C);
  C->N = N;
  int *order = (int *)malloc(sizeof(int) * n), *inv = (int *)malloc(sizeof(int) * n);
  min_degree_order(n, Q->rowptr, Q->col, order);
  for (int i = 0; i < n; ++i) inv[order[i]] = i;
  C->perm = (int *)malloc(sizeof(int) * N);
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < 4; ++c) C->perm[4 * i + c] = 4 * order[i] + c;
  /* upper-triangular CSC of the permuted matrix (entry (row,col), row <= col) */
  int *Cp = (int *)calloc(N + 1, sizeof(int));
  for (int j = 0; j < n; ++j)
    for (int p = Q->rowptr[j]; p < Q->rowptr[j + 1]; ++p) {
      int i = Q->col[p]; /* block Q_ij: rows of pose i, cols of pose j */
      for (int c = 0; c < 4; ++c)
        for (int cp = 0; cp < 4; ++cp) {
          int row = 4 * inv[i] + cp, colm = 4 * inv[j] + c;
          if (row <= colm && (Q->val[(size_t)16 * p + cp + 4 * c] != 0.0 || row == colm)) Cp[colm + 1]++;
        }
    }
  for (int k = 0; k < N; ++k) Cp[k + 1] += Cp[k];
  int nnz = Cp[N];
  int *Ci = (int *)malloc(sizeof(int) * nnz), *fill = (int *)malloc(sizeof(int) * N);
  double *Cx = (double *)malloc(sizeof(double) * nnz);
  memcpy(fill, Cp, sizeof(int) * N);
  for (int j = 0; j < n; ++j)
    for (int p = Q->rowptr[j]; p < Q->rowptr[j + 1]; ++p) {
      int i = Q->col[p];
      for (int c = 0; c < 4; ++c)
        for (int cp = 0; cp < 4; ++cp) {
          int row = 4 * inv[i] + cp, colm = 4 * inv[j] + c;
          double v = Q->val[(size_t)16 * p + cp + 4 * c];
          if (row <= colm && (v != 0.0 || row == colm)) {
            int q = fill[colm]++;
            Ci[q] = row;
            Cx[q] = v + (row == colm ? shift : 0.0);
          }
        }
    }
  /* elimination tree */
  int *parent = (int *)malloc(sizeof(int) * N), *anc = (int *)malloc(sizeof(int) * N);
  for (int k = 0; k < N; ++k) {
    parent[k] = -1; anc[k] = -1;
    for (int p = Cp[k]; p < Cp[k + 1]; ++p) {
      int i = Ci[p];
      while (i != -1 && i < k) {
        int inext = anc[i];
        anc[i] = k;
        if (inext == -1) parent[i] = k;
        i = inext;
      }
    }
  }
  int *s = (int *)malloc(sizeof(int) * N), *w = (int *)malloc(sizeof(int) * N);
  int *cnt = (int *)calloc(N, sizeof(int));
#define EREACH(k, top)                                                        \
  do {                                                                        \
    top = N; w[k] = k;                                                        \
    for (int p_ = Cp[k]; p_ < Cp[k + 1]; ++p_) {                              \
      int i_ = Ci[p_];                                                        \
      if (i_ > k) continue;                                                   \
      int len_ = 0;                                                           \
      for (; w[i_] != k; i_ = parent[i_]) { s[len_++] = i_; w[i_] = k; }      \
      while (len_ > 0) s[--top] = s[--len_];                                  \
    }                                                                         \
  } while (0)
  for (int k = 0; k < N; ++k) w[k] = -1;
  for (int k = 0; k < N; ++k) {
    int top;
    EREACH(k, top);
    for (int q = top; q < N; ++q) cnt[s[q]]++;
    cnt[k]++;
  }
  C->Lp = (int *)malloc(sizeof(int) * (N + 1));
  C->Lp[0] = 0;
  for (int k = 0; k < N; ++k) C->Lp[k + 1] = C->Lp[k] + cnt[k];
  int lnz = C->Lp[N];
  C->Li = (int *)malloc(sizeof(int) * lnz);
  C->Lx = (double *)malloc(sizeof(double) * lnz);
  int *c = (int *)malloc(sizeof(int) * N);
  double *x = (double *)calloc(N, sizeof(double));
  memcpy(c, C->Lp, sizeof(int) * N);
  for (int k = 0; k < N; ++k) w[k] = -1;
  int ok = 1;
  for (int k = 0; k < N && ok; ++k) {
    int top;
    EREACH(k, top);
    x[k] = 0;
    for (int p = Cp[k]; p < Cp[k + 1]; ++p)
      if (Ci[p] <= k) x[Ci[p]] = Cx[p];
    double d = x[k];
    x[k] = 0;
    for (; top < N; ++top) {
      int i = s[top];
      double lki = x[i] / C->Lx[C->Lp[i]];
      x[i] = 0;
      for (int p = C->Lp[i] + 1; p < c[i]; ++p) x[C->Li[p]] -= C->Lx[p] * lki;
      d -= lki * lki;
      int p = c[i]++;
      C->Li[p] = k;
      C->Lx[p] = lki;
    }
    if (d <= 0) { ok = 0; break; }
    int p = c[k]++;
    C->Li[p] = k;
    C->Lx[p] = sqrt(d);
  }
#undef EREACH
  C->work = (double *)malloc(sizeof(double) * (size_t)N * 16);
  free(order); free(inv); free(Cp); free(Ci); free(Cx); free(fill); free(parent); free(anc);
  free(s); free(w); free(cnt); free(c); free(x);
  return ok ? 0 : -1;
}

void orc_chol_free(orc_chol_t *C) {
  free(C->perm); free(C->Lp); free(C->Li); free(C->Lx); free(C->work);
  memset(C, 0, sizeof *C);
}

/* out = B (Q + sI)^{-1}, B and out are r x N column-major (each scalar column = r-vector) */
void orc_chol_solve(const orc_chol_t *C, const double *B, int r, double *out) {
  int N = C->N;
  double *y = C->work;
  for (int k = 0; k < N; ++k)
    for (int a = 0; a < r; ++a) y[(size_t)k * r + a] = B[(size_t)C->perm[k] * r + a];
  for (int j = 0; j < N; ++j) {
    double dj = C->Lx[C->Lp[j]];
    double *yj = y + (size_t)j * r;
    for (int a = 0; a < r; ++a) yj[a] /= dj;
    for (int p = C->Lp[j] + 1; p < C->Lp[j + 1]; ++p) {
      double l = C->Lx[p];
      double *yi = y + (size_t)C->Li[p] * r;
      for (int a = 0; a < r; ++a) yi[a] -= l * yj[a];
    }
  }
  for (int j = N - 1; j >= 0; --j) {
    double *yj = y + (size_t)j * r;
    for (int p = C->Lp[j] + 1; p < C->Lp[j + 1]; ++p) {
      double l = C->Lx[p];
      const double *yi = y + (size_t)C->Li[p] * r;
      for (int a = 0; a < r; ++a) yj[a] -= l * yi[a];
    }
    double dj = C->Lx[C->Lp[j]];
    for (int a = 0; a < r; ++a) yj[a] /= dj;
  }
  for (int k = 0; k < N; ++k)
    for (int a = 0; a < r; ++a) out[(size_t)C->perm[k] * r + a] = y[(size_t)k * r + a];
}

/* ------------------------------------------------------------------ QuadraticProblem (a3) */
/* f = 1/2 <XQ, X> + <G, X>;  egrad = XQ + G  (egrad must be a valid buffer) */
double orc_problem_f(const orc_problem_t *P, const double *X, double *egrad) {
  size_t N = (size_t)P->r * 4 * P->n;
  orc_bsr_mult(&P->Q, X, P->r, egrad);
  double f = 0;
  for (size_t i = 0; i < N; ++i) {
    f += 0.5 * egrad[i] * X[i] + P->G[i] * X[i];
    egrad[i] += P->G[i];
  }
  return f;
}

/* Riemannian Hessian:  P_X( eta Q - eta_Y sym(Y^T egrad_Y) ) on rotation blocks, (eta Q)_p on
 * translations (Weingarten form for the embedded Stiefel manifold, Absil et al. 2008 s.5.3). */
void orc_problem_hessvec(const orc_problem_t *P, const double *X, const double *egrad,
                         const double *eta, double *out) {
  int r = P->r, n = P->n;
  orc_bsr_mult(&P->Q, eta, r, out);
  for (int i = 0; i < n; ++i) {
    size_t o = (size_t)i * 4 * r;
    const double *Y = X + o, *E = egrad + o, *H = eta + o;
    double S[9];
    for (int p = 0; p < 3; ++p)
      for (int q = 0; q < 3; ++q) {
        double s = 0;
        for (int a = 0; a < r; ++a) s += Y[p * r + a] * E[q * r + a];
        S[3 * p + q] = s;
      }
    for (int q = 0; q < 3; ++q)
      for (int a = 0; a < r; ++a) {
        double s = 0;
        for (int p = 0; p < 3; ++p) s += H[p * r + a] * 0.5 * (S[3 * p + q] + S[3 * q + p]);
        out[o + q * r + a] -= s;
      }
  }
  orc_tangent_project(X, out, r, n, out);
}

/* inverse of an SPD 4x4 block by Gauss-Jordan (no pivoting needed) */
static void inv4_spd(const double *B, double *out) {
  double a[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i][j] = B[i + 4 * j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
  for (int k = 0; k < 4; ++k) {
    double p = 1.0 / a[k][k];
    for (int j = 0; j < 8; ++j) a[k][j] *= p;
    for (int i = 0; i < 4; ++i) if (i != k) { double f = a[i][k]; for (int j = 0; j < 8; ++j) a[i][j] -= f * a[k][j]; }
  }
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) out[i + 4 * j] = a[i][4 + j];
}

void orc_block_jacobi(const orc_bsr_t *Q, double shift, double **dinv) {
  double *D = (double *)calloc((size_t)16 * Q->n, sizeof(double));
  for (int j = 0; j < Q->n; ++j) {
    double B[16] = {0};
    for (int p = Q->rowptr[j]; p < Q->rowptr[j + 1]; ++p)
      if (Q->col[p] == j) memcpy(B, Q->val + (size_t)16 * p, sizeof B);
    for (int c = 0; c < 4; ++c) B[5 * c] += shift;
    inv4_spd(B, D + (size_t)16 * j);
  }
  *dinv = D;
}

/* PreConditioner: solve with Q + shift I (or, precond_mode 2, with its 4x4 block diagonal), then project to the tangent space */
void orc_problem_precond(const orc_problem_t *P, const double *X, const double *V, double *out) {
  if (P->dinv) {
    int r = P->r;
    for (int j = 0; j < P->n; ++j) {
      const double *B = P->dinv + (size_t)16 * j;
      double tmp[4 * 16];
      for (int c = 0; c < 4; ++c)
        for (int a = 0; a < r; ++a) {
          double s = 0;
          for (int cp = 0; cp < 4; ++cp) s += V[((size_t)4 * j + cp) * r + a] * B[cp + 4 * c];
          tmp[c * r + a] = s;
        }
      memcpy(out + (size_t)4 * j * r, tmp, sizeof(double) * 4 * r);
    }
  } else {
    orc_chol_solve(&P->chol, V, P->r, out);
  }
  orc_tangent_project(X, out, P->r, P->n, out);
}

void orc_problem_free(orc_problem_t *P) {
  orc_bsr_free(&P->Q);
  free(P->G);
  if (P->has_chol) orc_chol_free(&P->chol);
  free(P->dinv);
  memset(P, 0, sizeof *P);
}

/* ------------------------------------------------------------------ QuadraticOptimizer (a4) */
enum { TCG_MAXITER = 0, TCG_NEGCURV = 1, TCG_EXCREGION = 2, TCG_LCON = 3, TCG_SCON = 4 };

typedef struct {
  double *eta, *r, *z, *delta, *Hd, *tmp;
} tcg_ws_t;

/* Steihaug-Toint truncated CG in the tangent space at x (preconditioned), eta0 = 0.
 * Absil/Baker/Gallivan 2007 Alg. 2 as implemented by ROPTLIB SolversTR::tCG_TR
 * (theta = 1, kappa = 0.1, Min_Inner_Iter = 0) [UPSTREAM-RECALL]. */
static int tcg(const orc_problem_t *P, const orc_params_t *prm, const double *x, const double *egrad,
               const double *gf, double Delta, tcg_ws_t *ws, orc_opt_result_t *res) {
  size_t N = (size_t)P->r * 4 * P->n;
  const double theta = 1.0, kappa = 0.1;
  double *eta = ws->eta, *r = ws->r, *z = ws->z, *delta = ws->delta, *Hd = ws->Hd;
  memset(eta, 0, sizeof(double) * N);
  memcpy(r, gf, sizeof(double) * N);
  double r_r = orc_dot(r, r, N), norm_r0 = sqrt(r_r);
  orc_problem_precond(P, x, r, z);
  res->precond_count++;
  double z_r = orc_dot(z, r, N), d_Pd = z_r, e_Pd = 0, e_Pe = 0;
  for (size_t i = 0; i < N; ++i) delta[i] = -z[i];
  int status = TCG_MAXITER;
  for (int j = 0; j < prm->rtr_tcg_iterations; ++j) {
    orc_problem_hessvec(P, x, egrad, delta, Hd);
    res->hessvec_count++;
    res->tcg_iters_total++;
    double d_Hd = orc_dot(delta, Hd, N);
    double alpha = z_r / d_Hd;
    double e_Pe_new = e_Pe + 2.0 * alpha * e_Pd + alpha * alpha * d_Pd;
    if (d_Hd <= 0 || e_Pe_new >= Delta * Delta) {
      double tau = (-e_Pd + sqrt(e_Pd * e_Pd + d_Pd * (Delta * Delta - e_Pe))) / d_Pd;
      for (size_t i = 0; i < N; ++i) eta[i] += tau * delta[i];
      status = (d_Hd <= 0) ? TCG_NEGCURV : TCG_EXCREGION;
      break;
    }
    e_Pe = e_Pe_new;
    for (size_t i = 0; i < N; ++i) { eta[i] += alpha * delta[i]; r[i] += alpha * Hd[i]; }
    r_r = orc_dot(r, r, N);
    double nr = sqrt(r_r);
    double thr = pow(norm_r0, theta);
    if (nr <= norm_r0 * (thr < kappa ? thr : kappa)) {
      status = (kappa < thr) ? TCG_LCON : TCG_SCON;
      break;
    }
    orc_problem_precond(P, x, r, z);
    res->precond_count++;
    double zold_rold = z_r;
    z_r = orc_dot(z, r, N);
    double beta = z_r / zold_rold;
    for (size_t i = 0; i < N; ++i) delta[i] = -z[i] + beta * delta[i];
    e_Pd = beta * (e_Pd + alpha * d_Pd);
    d_Pd = z_r + beta * beta * d_Pd;
  }
  return status;
}

void orc_optimize(const orc_problem_t *P, const orc_params_t *prm, const double *X0, double *Xout,
                  orc_opt_result_t *res) {
  int r = P->r, n = P->n;
  size_t N = (size_t)r * 4 * n;
  memset(res, 0, sizeof *res);
  double *egrad = (double *)malloc(sizeof(double) * N), *gf = (double *)malloc(sizeof(double) * N);
  double f1 = orc_problem_f(P, X0, egrad);
  orc_tangent_project(X0, egrad, r, n, gf);
  double ngf = sqrt(orc_dot(gf, gf, N));
  res->f_init = f1;
  res->gradnorm_init = ngf;
  if (prm->method == ORC_METHOD_RGD) {
    /* single (preconditioned) Riemannian gradient step with fixed stepsize */
    double *dir = (double *)malloc(sizeof(double) * N);
    if (prm->rgd_use_preconditioner) { orc_problem_precond(P, X0, gf, dir); res->precond_count++; }
    else memcpy(dir, gf, sizeof(double) * N);
    if (prm->rgd_line_search) {
      /* Armijo backtracking along the retraction curve (Absil, Mahony, Sepulchre 2008, Def. 4.2.2): the first trial step
       * t_j = stepsize * shrink^j with sufficient decrease; none within max_backoffs: no step */
      double slope = orc_dot(gf, dir, N), step = prm->rgd_stepsize;
      double *eta = (double *)malloc(sizeof(double) * N), *eg = (double *)malloc(sizeof(double) * N);
      int j, ok = 0;
      for (j = 0; j <= prm->rgd_ls_max_backoffs; ++j) {
        for (size_t i = 0; i < N; ++i) eta[i] = -step * dir[i];
        orc_retract(X0, eta, r, n, Xout);
        double ft = orc_problem_f(P, Xout, eg);
        if (ft <= f1 - prm->rgd_ls_sigma * step * slope) { ok = 1; break; }
        step *= prm->rgd_ls_shrink;
      }
      if (!ok) memcpy(Xout, X0, sizeof(double) * N);
      res->ls_backoffs = j;
      res->accepted = ok;
      free(eta); free(eg);
    } else {
      for (size_t i = 0; i < N; ++i) dir[i] *= -prm->rgd_stepsize;
      orc_retract(X0, dir, r, n, Xout);
      res->accepted = 1;
    }
    free(dir);
  } else {
    double *x1 = (double *)malloc(sizeof(double) * N), *x2 = (double *)malloc(sizeof(double) * N);
    double *egrad2 = (double *)malloc(sizeof(double) * N), *Heta = (double *)malloc(sizeof(double) * N);
    tcg_ws_t ws;
    ws.eta = (double *)malloc(sizeof(double) * N); ws.r = (double *)malloc(sizeof(double) * N);
    ws.z = (double *)malloc(sizeof(double) * N); ws.delta = (double *)malloc(sizeof(double) * N);
    ws.Hd = (double *)malloc(sizeof(double) * N); ws.tmp = NULL;
    memcpy(x1, X0, sizeof(double) * N);
    double Delta = prm->rtr_initial_radius;
    for (int it = 0; it < prm->rtr_iterations; ++it) {
      if (ngf < prm->gradnorm_tol) break; /* StopCrit GRAD_F */
      int status = tcg(P, prm, x1, egrad, gf, Delta, &ws, res);
      res->rtr_outer_iters++;
      orc_retract(x1, ws.eta, r, n, x2);
      double f2 = orc_problem_f(P, x2, egrad2);
      orc_problem_hessvec(P, x1, egrad, ws.eta, Heta);
      res->hessvec_count++;
      double rho = (f1 - f2) / (-orc_dot(gf, ws.eta, N) - 0.5 * orc_dot(ws.eta, Heta, N));
      if (rho > 0.75) {
        if (status == TCG_NEGCURV || status == TCG_EXCREGION) {
          Delta = 2.0 * Delta;
          if (Delta > prm->rtr_max_radius) Delta = prm->rtr_max_radius;
        }
      } else if (rho < 0.25) {
        Delta = 0.25 * Delta;
      }
      if (rho > 0.1) { /* Acceptence_Rho */
        memcpy(x1, x2, sizeof(double) * N);
        memcpy(egrad, egrad2, sizeof(double) * N);
        f1 = f2;
        orc_tangent_project(x1, egrad, r, n, gf);
        ngf = sqrt(orc_dot(gf, gf, N));
        res->accepted++;
      }
    }
    memcpy(Xout, x1, sizeof(double) * N);
    free(x1); free(x2); free(egrad2); free(Heta);
    free(ws.eta); free(ws.r); free(ws.z); free(ws.delta); free(ws.Hd);
  }
  res->f_opt = orc_problem_f(P, Xout, egrad);
  orc_tangent_project(Xout, egrad, r, n, gf);
  res->gradnorm_opt = sqrt(orc_dot(gf, gf, N));
  res->success = 1;
  free(egrad); free(gf);
}
