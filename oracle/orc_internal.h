/* orc_internal.h -- internal declarations of the CPU oracle (test infrastructure only). */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H
#include "dpgo_oracle.h"
#include <stddef.h>

#define ORC_K 4 /* d + 1 */

/* block-sparse symmetric matrix, 4x4 blocks.  Row j lists blocks (i, B=Q_ij col-major) such that
 * (XQ)_j[:,c] = sum_i sum_c' X_i[:,c'] * B[c' + 4*c]. */
typedef struct {
  int n, nb;
  int *rowptr, *col;
  double *val;
} orc_bsr_t;

/* sparse Cholesky of P (Q + shift I) P^T, scalar CSC lower-triangular L */
typedef struct {
  int N;
  int *perm;  /* perm[new] = old scalar index */
  int *Lp, *Li;
  double *Lx;
  double *work;
} orc_chol_t;

typedef struct {
  int r, n;
  orc_bsr_t Q;
  double *G;       /* r x 4n */
  orc_chol_t chol; /* of Q + shift I */
  int has_chol;
  double *dinv;    /* precond_mode 2: n inverted 4x4 diagonal blocks of Q + shift I (column-major), else NULL */
} orc_problem_t;

void orc_bsr_free(orc_bsr_t *Q);
/* triplet builder */
typedef struct {
  int row, col;
  double v[16];
} orc_trip_t;
void orc_bsr_from_triplets(orc_trip_t *t, int nt, int n, orc_bsr_t *Q);
void orc_bsr_mult(const orc_bsr_t *Q, const double *X, int r, double *out); /* out = X Q */

int orc_chol_factor(const orc_bsr_t *Q, double shift, orc_chol_t *C);
void orc_chol_solve(const orc_chol_t *C, const double *B, int r, double *out); /* out = B (Q+sI)^-1 */
void orc_chol_free(orc_chol_t *C);
/* inverses of the 4x4 diagonal blocks of Q + shift I (n x 16 doubles, column-major blocks) */
void orc_block_jacobi(const orc_bsr_t *Q, double shift, double **dinv);

void orc_edge_blocks(const orc_meas_t *m, double TO[16], double TOT[16], double Om[16]);

double orc_problem_f(const orc_problem_t *P, const double *X, double *egrad /*nullable scratch req*/);
void orc_problem_hessvec(const orc_problem_t *P, const double *X, const double *egrad,
                         const double *eta, double *out);
void orc_problem_precond(const orc_problem_t *P, const double *X, const double *V, double *out);
void orc_problem_free(orc_problem_t *P);

void orc_optimize(const orc_problem_t *P, const orc_params_t *prm, const double *X0, double *Xout,
                  orc_opt_result_t *res);

double orc_dot(const double *a, const double *b, size_t n);
void orc_quat_to_rot(double qx, double qy, double qz, double qw, double R[9]);
#endif
