/* orc_agent.c -- CPU oracle: PGOAgent::iterate, Nesterov sequences, GNC-TLS, team schedule.
 * TEST INFRASTRUCTURE ONLY (see dpgo_oracle.h: parity unpinned).
 *
 * Follows (bodies external, call sites cited):
 *   a1 PGOAgent::iterate(bool)                          src/PGOAgentROS.cpp:160 (true), :1185 (false)
 *   a6 Nesterov gamma/alpha/Y/V + restart               src/PGOAgentROSNode.cpp:126-130
 *   a7 update(Aux)NeighborPoses / get(Aux)SharedPoseDict src/PGOAgentROS.cpp:1276,1278,666,668
 *   a8 robust path                                      src/PGOAgentROS.cpp:1218,1049,1050,1341,210
 *   a9 status / termination                             src/PGOAgentROS.cpp:208,616; tests/testUtils.cpp:56
 *   schedule (token passing, everyone else iterate(false)) src/PGOAgentROS.cpp:129-220,443-504,1161-1189
 */
#include "orc_internal.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void orc_default_params(orc_params_t *p, int r, int num_robots) {
  memset(p, 0, sizeof *p);
  p->d = 3; p->r = r; p->num_robots = num_robots;
  p->method = ORC_METHOD_RTR;
  p->rgd_stepsize = 1e-3;        /* launch/PGOAgent.launch:16 */
  p->rgd_use_preconditioner = 1; /* :17 */
  p->rtr_iterations = 3;         /* :18 */
  p->rtr_tcg_iterations = 50;    /* :19 */
  p->gradnorm_tol = 1e-2;        /* :20 */
  p->rtr_initial_radius = 100.0;
  p->rtr_max_radius = 500.0;
  p->precond_shift = 0.1;
  p->acceleration = 0;           /* :24 */
  p->restart_interval = 50;      /* :25 */
  p->rel_change_tol = 0.1;       /* :38 */
  p->max_num_iters = 1000;       /* :37 */
  p->robust_cost_type = ORC_COST_L2;
  p->gnc_barc = 5.0;             /* PGOAgentROSNode.cpp:205 */
  p->gnc_mu_step = 2.0;          /* launch :30 */
  p->gnc_init_mu = 1e-5;         /* launch :31 */
  p->robust_opt_num_weight_updates = 4;
  p->robust_opt_inner_iters = 10 * num_robots; /* PGOAgentROSNode.cpp:216-218 */
  p->robust_opt_min_convergence_ratio = 0.8;
  p->weights_as_float32 = 0;
  p->robust_opt_num_resets = 0;  /* launch/PGOAgent.launch:33 */
  p->precond_mode = 0;
  p->status_every_iterate = 0;
  p->rgd_line_search = 0;
  p->rgd_ls_max_backoffs = 7;
  p->rgd_ls_shrink = 0.5;
  p->rgd_ls_sigma = 1e-4;
  p->tls_threshold = 10.0;
  p->huber_threshold = 3.0;
}

struct orc_agent {
  int id;
  orc_params_t prm;
  orc_meas_t *odom, *priv, *shared;
  int nodom, npriv, nshared, codom, cpriv, cshared;
  int n, index_dirty;
  double *X, *XPrev, *Y, *V;
  double gamma, alpha;
  int state, instance, iter;
  /* neighbour pose dictionary (sorted by robot, frame) */
  int nnp;
  int *np_robot, *np_frame;
  double *np_pose, *np_aux;
  char *np_has, *np_has_aux;
  orc_problem_t prob;
  int prob_valid;
  orc_status_t status;
  orc_opt_result_t opt;
  int publish_requested;
  double mu;
  int weight_update_count, robust_inner_iter;
};

static void push_meas(orc_meas_t **arr, int *n, int *cap, const orc_meas_t *m) {
  if (*n == *cap) {
    *cap = *cap ? *cap * 2 : 64;
    *arr = (orc_meas_t *)realloc(*arr, sizeof(orc_meas_t) * *cap);
  }
  (*arr)[(*n)++] = *m;
}

orc_agent_t *orc_agent_new(int id, const orc_params_t *p) {
  orc_agent_t *a = (orc_agent_t *)calloc(1, sizeof *a);
  a->id = id;
  a->prm = *p;
  a->state = ORC_STATE_WAIT_FOR_DATA;
  a->mu = p->gnc_init_mu;
  a->index_dirty = 1;
  return a;
}

static void free_np(orc_agent_t *a) {
  free(a->np_robot); free(a->np_frame); free(a->np_pose); free(a->np_aux);
  free(a->np_has); free(a->np_has_aux);
  a->np_robot = a->np_frame = NULL; a->np_pose = a->np_aux = NULL; a->np_has = a->np_has_aux = NULL;
  a->nnp = 0;
}

void orc_agent_free(orc_agent_t *a) {
  if (!a) return;
  free(a->odom); free(a->priv); free(a->shared);
  free(a->X); free(a->XPrev); free(a->Y); free(a->V);
  free_np(a);
  if (a->prob_valid) orc_problem_free(&a->prob);
  free(a);
}

void orc_agent_add_measurement(orc_agent_t *a, const orc_meas_t *m) {
  if (m->r1 == a->id && m->r2 == a->id) {
    if (m->p1 + 1 == m->p2) push_meas(&a->odom, &a->nodom, &a->codom, m);
    else push_meas(&a->priv, &a->npriv, &a->cpriv, m);
  } else if (m->r1 == a->id || m->r2 == a->id) {
    push_meas(&a->shared, &a->nshared, &a->cshared, m);
  } else {
    return;
  }
  a->index_dirty = 1;
  if (a->state == ORC_STATE_WAIT_FOR_DATA) a->state = ORC_STATE_WAIT_FOR_INITIALIZATION;
}

static int pair_cmp(const void *x, const void *y) {
  const int *a = (const int *)x, *b = (const int *)y;
  if (a[0] != b[0]) return a[0] < b[0] ? -1 : 1;
  if (a[1] != b[1]) return a[1] < b[1] ? -1 : 1;
  return 0;
}

static void rebuild_index(orc_agent_t *a) {
  if (!a->index_dirty) return;
  int n = 0;
  for (int k = 0; k < a->nodom; ++k) { if (a->odom[k].p1 + 1 > n) n = a->odom[k].p1 + 1; if (a->odom[k].p2 + 1 > n) n = a->odom[k].p2 + 1; }
  for (int k = 0; k < a->npriv; ++k) { if (a->priv[k].p1 + 1 > n) n = a->priv[k].p1 + 1; if (a->priv[k].p2 + 1 > n) n = a->priv[k].p2 + 1; }
  int *pairs = (int *)malloc(sizeof(int) * 2 * (a->nshared + 1));
  for (int k = 0; k < a->nshared; ++k) {
    const orc_meas_t *m = &a->shared[k];
    if (m->r1 == a->id) { if (m->p1 + 1 > n) n = m->p1 + 1; pairs[2 * k] = m->r2; pairs[2 * k + 1] = m->p2; }
    else { if (m->p2 + 1 > n) n = m->p2 + 1; pairs[2 * k] = m->r1; pairs[2 * k + 1] = m->p1; }
  }
  qsort(pairs, a->nshared, 2 * sizeof(int), pair_cmp);
  int nu = 0;
  for (int k = 0; k < a->nshared; ++k)
    if (k == 0 || pair_cmp(&pairs[2 * k], &pairs[2 * (k - 1)]) != 0) { pairs[2 * nu] = pairs[2 * k]; pairs[2 * nu + 1] = pairs[2 * k + 1]; ++nu; }
  free_np(a);
  a->nnp = nu;
  int r = a->prm.r;
  a->np_robot = (int *)malloc(sizeof(int) * (nu + 1));
  a->np_frame = (int *)malloc(sizeof(int) * (nu + 1));
  a->np_pose = (double *)calloc((size_t)(nu + 1) * 4 * r, sizeof(double));
  a->np_aux = (double *)calloc((size_t)(nu + 1) * 4 * r, sizeof(double));
  a->np_has = (char *)calloc(nu + 1, 1);
  a->np_has_aux = (char *)calloc(nu + 1, 1);
  for (int k = 0; k < nu; ++k) { a->np_robot[k] = pairs[2 * k]; a->np_frame[k] = pairs[2 * k + 1]; }
  free(pairs);
  a->n = n;
  a->index_dirty = 0;
}

static int find_np(const orc_agent_t *a, int robot, int frame) {
  int lo = 0, hi = a->nnp - 1;
  while (lo <= hi) {
    int mid = (lo + hi) / 2;
    int key[2] = {a->np_robot[mid], a->np_frame[mid]}, q[2] = {robot, frame};
    int c = pair_cmp(key, q);
    if (c == 0) return mid;
    if (c < 0) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

int orc_agent_num_poses(const orc_agent_t *a) { rebuild_index((orc_agent_t *)a); return a->n; }

int orc_agent_num_measurements(const orc_agent_t *a, int *odom, int *priv, int *shared) {
  if (odom) *odom = a->nodom;
  if (priv) *priv = a->npriv;
  if (shared) *shared = a->nshared;
  return a->nodom + a->npriv + a->nshared;
}

int orc_agent_num_neighbors(const orc_agent_t *a, int *ids) {
  rebuild_index((orc_agent_t *)a);
  int cnt = 0, last = -1;
  for (int k = 0; k < a->nnp; ++k)
    if (a->np_robot[k] != last) { last = a->np_robot[k]; if (ids) ids[cnt] = last; ++cnt; }
  return cnt;
}

static int int_cmp(const void *x, const void *y) { int a = *(const int *)x, b = *(const int *)y; return a < b ? -1 : a > b; }

int orc_agent_public_pose_ids(const orc_agent_t *a, int nbr, int *frames) {
  int *tmp = (int *)malloc(sizeof(int) * (a->nshared + 1)), c = 0;
  for (int k = 0; k < a->nshared; ++k) {
    const orc_meas_t *m = &a->shared[k];
    if (m->r1 == a->id && m->r2 == nbr) tmp[c++] = m->p1;
    else if (m->r2 == a->id && m->r1 == nbr) tmp[c++] = m->p2;
  }
  qsort(tmp, c, sizeof(int), int_cmp);
  int nu = 0;
  for (int k = 0; k < c; ++k) if (k == 0 || tmp[k] != tmp[k - 1]) { if (frames) frames[nu] = tmp[k]; ++nu; }
  free(tmp);
  return nu;
}

int orc_agent_neighbor_pose_ids(const orc_agent_t *a, int nbr, int *frames) {
  rebuild_index((orc_agent_t *)a);
  int c = 0;
  for (int k = 0; k < a->nnp; ++k) if (a->np_robot[k] == nbr) { if (frames) frames[c] = a->np_frame[k]; ++c; }
  return c;
}

void orc_agent_set_X(orc_agent_t *a, const double *X) {
  rebuild_index(a);
  size_t N = (size_t)a->prm.r * 4 * a->n;
  free(a->X); free(a->XPrev); free(a->Y); free(a->V);
  a->X = (double *)malloc(sizeof(double) * N); a->XPrev = (double *)malloc(sizeof(double) * N);
  a->Y = (double *)malloc(sizeof(double) * N); a->V = (double *)malloc(sizeof(double) * N);
  memcpy(a->X, X, sizeof(double) * N); memcpy(a->XPrev, X, sizeof(double) * N);
  memcpy(a->Y, X, sizeof(double) * N); memcpy(a->V, X, sizeof(double) * N);
  a->gamma = 0; a->alpha = 0;
  a->state = ORC_STATE_INITIALIZED;
}
void orc_agent_get_X(const orc_agent_t *a, double *X) { memcpy(X, a->X, sizeof(double) * (size_t)a->prm.r * 4 * a->n); }
void orc_agent_get_Y(const orc_agent_t *a, double *Y) { memcpy(Y, a->Y, sizeof(double) * (size_t)a->prm.r * 4 * a->n); }
void orc_agent_get_V(const orc_agent_t *a, double *V) { memcpy(V, a->V, sizeof(double) * (size_t)a->prm.r * 4 * a->n); }

int orc_agent_get_public_poses(const orc_agent_t *a, int nbr, int aux, double *out) {
  int *frames = (int *)malloc(sizeof(int) * (a->nshared + 1));
  int c = orc_agent_public_pose_ids(a, nbr, frames);
  size_t B = (size_t)4 * a->prm.r;
  const double *src = aux ? a->Y : a->X;
  for (int k = 0; k < c; ++k) memcpy(out + k * B, src + frames[k] * B, sizeof(double) * B);
  free(frames);
  return c;
}

void orc_agent_update_neighbor_poses(orc_agent_t *a, int nbr, int aux, int count, const int *frames,
                                     const double *poses) {
  rebuild_index(a);
  size_t B = (size_t)4 * a->prm.r;
  for (int k = 0; k < count; ++k) {
    int q = find_np(a, nbr, frames[k]);
    if (q < 0) continue; /* not needed by any shared edge: dropped */
    if (aux) { memcpy(a->np_aux + q * B, poses + k * B, sizeof(double) * B); a->np_has_aux[q] = 1; }
    else { memcpy(a->np_pose + q * B, poses + k * B, sizeof(double) * B); a->np_has[q] = 1; }
  }
}

/* ---------------------------------------------------------------- data matrices (a2) */
static void build_Q(orc_agent_t *a) {
  rebuild_index(a);
  int nt = 4 * (a->nodom + a->npriv) + a->nshared + a->n;
  orc_trip_t *t = (orc_trip_t *)calloc(nt, sizeof(orc_trip_t));
  int c = 0;
  double TO[16], TOT[16], Om[16];
  for (int i = 0; i < a->n; ++i) { t[c].row = i; t[c].col = i; ++c; } /* every pose owns a diagonal block */
  for (int pass = 0; pass < 2; ++pass) {
    const orc_meas_t *arr = pass ? a->priv : a->odom;
    int cnt = pass ? a->npriv : a->nodom;
    for (int k = 0; k < cnt; ++k) {
      const orc_meas_t *m = &arr[k];
      int i = m->p1, j = m->p2;
      orc_edge_blocks(m, TO, TOT, Om);
      t[c].row = i; t[c].col = i; memcpy(t[c].v, TOT, sizeof TOT); ++c;
      t[c].row = j; t[c].col = j; memcpy(t[c].v, Om, sizeof Om); ++c;
      t[c].row = j; t[c].col = i; for (int e = 0; e < 16; ++e) t[c].v[e] = -TO[e]; ++c;           /* Q_ij */
      t[c].row = i; t[c].col = j;
      for (int cp = 0; cp < 4; ++cp) for (int cc = 0; cc < 4; ++cc) t[c].v[cp + 4 * cc] = -TO[cc + 4 * cp]; /* Q_ji */
      ++c;
    }
  }
  for (int k = 0; k < a->nshared; ++k) {
    const orc_meas_t *m = &a->shared[k];
    orc_edge_blocks(m, TO, TOT, Om);
    if (m->r1 == a->id) { t[c].row = m->p1; t[c].col = m->p1; memcpy(t[c].v, TOT, sizeof TOT); ++c; }
    else { t[c].row = m->p2; t[c].col = m->p2; memcpy(t[c].v, Om, sizeof Om); ++c; }
  }
  orc_bsr_from_triplets(t, c, a->n, &a->prob.Q);
  free(t);
}

static int build_G(orc_agent_t *a, int aux) {
  int r = a->prm.r;
  size_t N = (size_t)r * 4 * a->n;
  memset(a->prob.G, 0, sizeof(double) * N);
  double TO[16], TOT[16], Om[16];
  for (int k = 0; k < a->nshared; ++k) {
    const orc_meas_t *m = &a->shared[k];
    orc_edge_blocks(m, TO, TOT, Om);
    int out = (m->r1 == a->id);
    int q = out ? find_np(a, m->r2, m->p2) : find_np(a, m->r1, m->p1);
    if (q < 0) return -1;
    if (aux ? !a->np_has_aux[q] : !a->np_has[q]) return -1;
    const double *Xn = (aux ? a->np_aux : a->np_pose) + (size_t)q * 4 * r;
    double *Gi = a->prob.G + (size_t)(out ? m->p1 : m->p2) * 4 * r;
    for (int cc = 0; cc < 4; ++cc)
      for (int cp = 0; cp < 4; ++cp) {
        double b = out ? TO[cc + 4 * cp] : TO[cp + 4 * cc];
        if (b == 0.0) continue;
        for (int x = 0; x < r; ++x) Gi[cc * r + x] -= Xn[cp * r + x] * b;
      }
  }
  return 0;
}

static int ensure_problem(orc_agent_t *a, int aux) {
  rebuild_index(a);
  if (!a->prob_valid) {
    memset(&a->prob, 0, sizeof a->prob);
    a->prob.r = a->prm.r; a->prob.n = a->n;
    build_Q(a);
    a->prob.G = (double *)calloc((size_t)a->prm.r * 4 * a->n, sizeof(double));
    if (a->prm.precond_mode == 2) {
      orc_block_jacobi(&a->prob.Q, a->prm.precond_shift, &a->prob.dinv);
    } else {
      if (orc_chol_factor(&a->prob.Q, a->prm.precond_shift, &a->prob.chol) != 0) {
        fprintf(stderr, "[oracle] Cholesky of Q + shift I failed (agent %d)\n", a->id);
      }
      a->prob.has_chol = 1;
    }
    a->prob_valid = 1;
  }
  return build_G(a, aux);
}

void orc_agent_clear_data_matrices(orc_agent_t *a) {
  if (a->prob_valid) { orc_problem_free(&a->prob); a->prob_valid = 0; }
}

void orc_agent_build_problem(orc_agent_t *a, int aux) { ensure_problem(a, aux); }

double orc_agent_eval(orc_agent_t *a, const double *X, double *egrad, double *rgrad) {
  size_t N = (size_t)a->prm.r * 4 * a->n;
  double *eg = egrad ? egrad : (double *)malloc(sizeof(double) * N);
  double f = orc_problem_f(&a->prob, X, eg);
  if (rgrad) orc_tangent_project(X, eg, a->prm.r, a->n, rgrad);
  if (!egrad) free(eg);
  return f;
}

void orc_agent_hessvec(orc_agent_t *a, const double *X, const double *eta, double *out) {
  size_t N = (size_t)a->prm.r * 4 * a->n;
  double *eg = (double *)malloc(sizeof(double) * N);
  orc_problem_f(&a->prob, X, eg);
  orc_problem_hessvec(&a->prob, X, eg, eta, out);
  free(eg);
}

void orc_agent_precondition(orc_agent_t *a, const double *X, const double *V, double *out) {
  orc_problem_precond(&a->prob, X, V, out);
}

int orc_agent_get_Q(orc_agent_t *a, int *rowptr, int *col, double *val) {
  if (rowptr) memcpy(rowptr, a->prob.Q.rowptr, sizeof(int) * (a->n + 1));
  if (col) memcpy(col, a->prob.Q.col, sizeof(int) * a->prob.Q.nb);
  if (val) memcpy(val, a->prob.Q.val, sizeof(double) * 16 * (size_t)a->prob.Q.nb);
  return a->prob.Q.nb;
}
void orc_agent_get_G(orc_agent_t *a, double *G) { memcpy(G, a->prob.G, sizeof(double) * (size_t)a->prm.r * 4 * a->n); }

/* ---------------------------------------------------------------- iterate (a1, a6) */
static int update_X(orc_agent_t *a, int do_opt, int accel) {
  size_t N = (size_t)a->prm.r * 4 * a->n;
  if (!do_opt) {
    if (accel) memcpy(a->X, a->Y, sizeof(double) * N);
    return 1;
  }
  if (ensure_problem(a, accel) != 0) return 0; /* a neighbour pose is missing: skip */
  double *Xn = (double *)malloc(sizeof(double) * N);
  orc_optimize(&a->prob, &a->prm, accel ? a->Y : a->X, Xn, &a->opt);
  memcpy(a->X, Xn, sizeof(double) * N);
  free(Xn);
  return 1;
}

static void reset_acceleration(orc_agent_t *a) {
  size_t N = (size_t)a->prm.r * 4 * a->n;
  memcpy(a->V, a->X, sizeof(double) * N);
  memcpy(a->Y, a->X, sizeof(double) * N);
  a->gamma = 0; a->alpha = 0;
}

static double converged_ratio(const orc_agent_t *a) {
  int total = a->npriv + a->nshared, conv = 0;
  for (int k = 0; k < a->npriv; ++k) if (a->priv[k].weight == 1.0 || a->priv[k].weight == 0.0) ++conv;
  for (int k = 0; k < a->nshared; ++k) if (a->shared[k].weight == 1.0 || a->shared[k].weight == 0.0) ++conv;
  return total ? (double)conv / total : 1.0;
}

int orc_agent_iterate(orc_agent_t *a, int do_opt) {
  a->iter++;
  if (a->prm.robust_cost_type != ORC_COST_L2) a->robust_inner_iter++;
  if (a->state != ORC_STATE_INITIALIZED) return 0;
  int r = a->prm.r, n = a->n;
  size_t N = (size_t)r * 4 * n;
  double Nr = (double)a->prm.num_robots;
  memcpy(a->XPrev, a->X, sizeof(double) * N);
  int success;
  if (a->prm.acceleration) {
    a->gamma = (1.0 + sqrt(1.0 + 4.0 * Nr * Nr * a->gamma * a->gamma)) / (2.0 * Nr);
    a->alpha = 1.0 / (a->gamma * Nr);
    double *tmp = (double *)malloc(sizeof(double) * N);
    for (size_t i = 0; i < N; ++i) tmp[i] = (1.0 - a->alpha) * a->X[i] + a->alpha * a->V[i];
    orc_project_manifold(tmp, r, n, a->Y);
    success = update_X(a, do_opt, 1);
    for (size_t i = 0; i < N; ++i) tmp[i] = a->V[i] + a->gamma * (a->X[i] - a->Y[i]);
    orc_project_manifold(tmp, r, n, a->V);
    free(tmp);
    if ((a->iter + 1) % a->prm.restart_interval == 0) { /* shouldRestart */
      memcpy(a->X, a->XPrev, sizeof(double) * N);
      update_X(a, do_opt, 0);
      reset_acceleration(a);
    }
    a->publish_requested = 1;
  } else {
    success = update_X(a, do_opt, 0);
    if (do_opt) a->publish_requested = 1;
  }
  a->status.agent_id = a->id;
  a->status.state = a->state;
  a->status.instance_number = a->instance;
  a->status.iteration_number = a->iter;
  if (do_opt || a->prm.status_every_iterate) {
    /* [UPSTREAM-RECALL] the status block sits under `if (doOptimization)`: a robot's relativeChange /
     * readyToTerminate describe its last block update, not the Nesterov bookkeeping of iterate(false) */
    double s = 0;
    for (size_t i = 0; i < N; ++i) { double d = a->X[i] - a->XPrev[i]; s += d * d; }
    a->status.relative_change = sqrt(s / n);
    int ready = success && (a->status.relative_change <= a->prm.rel_change_tol);
    /* robustOptMinConvergenceRatio (PGOAgentROSNode.cpp:214) [UPSTREAM-RECALL]: share of loop closures whose GNC
     * weight has converged to 0 or 1 */
    if (a->prm.robust_cost_type != ORC_COST_L2 && converged_ratio(a) < a->prm.robust_opt_min_convergence_ratio) ready = 0;
    a->status.ready_to_terminate = ready;
  }
  return success;
}

void orc_agent_get_status(const orc_agent_t *a, orc_status_t *s) {
  *s = a->status; /* relativeChange / readyToTerminate: as of the last refresh */
  s->agent_id = a->id; s->state = a->state; s->instance_number = a->instance; s->iteration_number = a->iter;
}
void orc_agent_get_opt_result(const orc_agent_t *a, orc_opt_result_t *r) { *r = a->opt; }
int orc_agent_iteration_number(const orc_agent_t *a) { return a->iter; }

/* ---------------------------------------------------------------- robust path (a8) */
static const double *pose_ptr(const orc_agent_t *a, int robot, int frame) {
  if (robot == a->id) return a->X + (size_t)frame * 4 * a->prm.r;
  int q = find_np(a, robot, frame);
  if (q < 0 || !a->np_has[q]) return NULL;
  return a->np_pose + (size_t)q * 4 * a->prm.r;
}

/* sqrt(kappa |Y_j - Y_i R|_F^2 + tau |p_j - p_i - Y_i t|^2) */
int orc_agent_compute_residual(const orc_agent_t *a, const orc_meas_t *m, double *res) {
  rebuild_index((orc_agent_t *)a);
  const double *Xi = pose_ptr(a, m->r1, m->p1), *Xj = pose_ptr(a, m->r2, m->p2);
  if (!Xi || !Xj) return 0;
  int r = a->prm.r;
  double sr = 0, st = 0;
  for (int x = 0; x < r; ++x) {
    for (int c = 0; c < 3; ++c) {
      double v = Xj[c * r + x];
      for (int b = 0; b < 3; ++b) v -= Xi[b * r + x] * m->R[3 * b + c];
      sr += v * v;
    }
    double v = Xj[3 * r + x] - Xi[3 * r + x];
    for (int b = 0; b < 3; ++b) v -= Xi[b * r + x] * m->t[b];
    st += v * v;
  }
  *res = sqrt(m->kappa * sr + m->tau * st);
  return 1;
}

double orc_robust_weight(const orc_agent_t *a, double residual) {
  /* the six types the wrapper names (PGOAgentROSNode.cpp:178-188); formulas [UPSTREAM-RECALL] mit-acl/dpgo
   * src/DPGO_robust.cpp RobustCost::weight -- the library is absent from the mount */
  switch (a->prm.robust_cost_type) {
    case ORC_COST_L2: return 1.0;
    case ORC_COST_L1: return 1.0 / residual;
    case ORC_COST_HUBER: return residual < a->prm.huber_threshold ? 1.0 : a->prm.huber_threshold / residual;
    case ORC_COST_TLS: return residual < a->prm.tls_threshold ? 1.0 : 0.0;
    case ORC_COST_GM: { double s = 1.0 + residual * residual; return 1.0 / (s * s); }
    default: break; /* GNC_TLS */
  }
  double r2 = residual * residual, b2 = a->prm.gnc_barc * a->prm.gnc_barc, mu = a->mu;
  double upper = (mu + 1.0) / mu * b2, lower = mu / (mu + 1.0) * b2;
  if (r2 >= upper) return 0.0;
  if (r2 <= lower) return 1.0;
  return sqrt(b2 * mu * (mu + 1.0) / r2) - mu;
}

void orc_agent_update_measurement_weights(orc_agent_t *a) {
  for (int pass = 0; pass < 2; ++pass) {
    orc_meas_t *arr = pass ? a->shared : a->priv;
    int cnt = pass ? a->nshared : a->npriv;
    for (int k = 0; k < cnt; ++k) {
      orc_meas_t *m = &arr[k];
      if (m->fixed_weight) continue;
      if (pass) { /* the lower-ID endpoint owns a shared edge's weight (PGOAgentROS.cpp:732,1340) */
        int other = (m->r1 == a->id) ? m->r2 : m->r1;
        if (other < a->id) continue;
      }
      double res;
      if (!orc_agent_compute_residual(a, m, &res)) continue;
      m->weight = orc_robust_weight(a, res);
    }
  }
  a->weight_update_count++;
  a->mu *= a->prm.gnc_mu_step;
  a->robust_inner_iter = 0;
  orc_agent_clear_data_matrices(a);
  if (a->prm.acceleration && a->X) reset_acceleration(a);
}

int orc_agent_set_measurement_weight(orc_agent_t *a, int r1, int p1, int r2, int p2, double w, int fixed) {
  for (int pass = 0; pass < 3; ++pass) {
    orc_meas_t *arr = pass == 0 ? a->odom : (pass == 1 ? a->priv : a->shared);
    int cnt = pass == 0 ? a->nodom : (pass == 1 ? a->npriv : a->nshared);
    for (int k = 0; k < cnt; ++k)
      if (arr[k].r1 == r1 && arr[k].p1 == p1 && arr[k].r2 == r2 && arr[k].p2 == p2) {
        arr[k].weight = w; arr[k].fixed_weight = fixed;
        return 1;
      }
  }
  return 0;
}

int orc_agent_get_measurements(const orc_agent_t *a, orc_meas_t *out) {
  int c = 0;
  if (out) {
    memcpy(out + c, a->odom, sizeof(orc_meas_t) * a->nodom); c += a->nodom;
    memcpy(out + c, a->priv, sizeof(orc_meas_t) * a->npriv); c += a->npriv;
    memcpy(out + c, a->shared, sizeof(orc_meas_t) * a->nshared); c += a->nshared;
  } else c = a->nodom + a->npriv + a->nshared;
  return c;
}

int orc_agent_should_update_weights(const orc_agent_t *a) {
  if (a->prm.robust_cost_type == ORC_COST_L2) return 0;
  if (a->weight_update_count >= a->prm.robust_opt_num_weight_updates) return 0;
  return a->robust_inner_iter >= a->prm.robust_opt_inner_iters;
}

/* chi-square inverse CDF: barc = sqrt(chi2inv(q, dim)) (PGOAgentROSNode.cpp:201).  Regularised
 * lower incomplete gamma by series, inverted by bisection. */
static double gammainc_lower_reg(double s, double x) {
  if (x <= 0) return 0;
  double sum = 1.0 / s, term = 1.0 / s;
  for (int k = 1; k < 2000; ++k) { term *= x / (s + k); sum += term; if (term < 1e-17 * sum) break; }
  return exp(-x + s * log(x) - lgamma(s)) * sum;
}
double orc_error_threshold_at_quantile(double q, int dim) {
  double lo = 0, hi = 1000;
  for (int it = 0; it < 200; ++it) {
    double mid = 0.5 * (lo + hi);
    if (gammainc_lower_reg(0.5 * dim, 0.5 * mid) < q) lo = mid; else hi = mid;
  }
  return sqrt(0.5 * (lo + hi));
}

/* ---------------------------------------------------------------- team (schedule, exchange, cost) */
struct orc_team {
  int N, nm, num_poses;
  orc_params_t prm;
  orc_meas_t *m;
  orc_agent_t **ag;
  int *offset; /* global pose offset of each robot */
  int *sched, sched_len, iter;
};

orc_team_t *orc_team_new(const orc_meas_t *m, int nm, int num_poses, const orc_params_t *p, int weight_mode) {
  (void)weight_mode;
  orc_team_t *t = (orc_team_t *)calloc(1, sizeof *t);
  t->N = p->num_robots; t->nm = nm; t->num_poses = num_poses; t->prm = *p;
  t->m = (orc_meas_t *)malloc(sizeof(orc_meas_t) * nm);
  memcpy(t->m, m, sizeof(orc_meas_t) * nm);
  t->ag = (orc_agent_t **)malloc(sizeof(orc_agent_t *) * t->N);
  for (int k = 0; k < t->N; ++k) t->ag[k] = orc_agent_new(k, p);
  for (int e = 0; e < nm; ++e) {
    orc_agent_add_measurement(t->ag[m[e].r1], &m[e]);
    if (m[e].r2 != m[e].r1) orc_agent_add_measurement(t->ag[m[e].r2], &m[e]);
  }
  t->offset = (int *)malloc(sizeof(int) * (t->N + 1));
  t->offset[0] = 0;
  for (int k = 0; k < t->N; ++k) t->offset[k + 1] = t->offset[k] + orc_agent_num_poses(t->ag[k]);
  t->sched_len = t->N;
  t->sched = (int *)malloc(sizeof(int) * t->N);
  for (int k = 0; k < t->N; ++k) t->sched[k] = k; /* RoundRobin from the leader (PGOAgentROS.cpp:464-473,1138-1151) */
  return t;
}

void orc_team_free(orc_team_t *t) {
  if (!t) return;
  for (int k = 0; k < t->N; ++k) orc_agent_free(t->ag[k]);
  free(t->ag); free(t->m); free(t->offset); free(t->sched); free(t);
}

orc_agent_t *orc_team_agent(orc_team_t *t, int id) { return t->ag[id]; }
int orc_team_iteration(const orc_team_t *t) { return t->iter; }

void orc_team_set_schedule(orc_team_t *t, const int *order, int len) {
  free(t->sched);
  t->sched = (int *)malloc(sizeof(int) * len);
  memcpy(t->sched, order, sizeof(int) * len);
  t->sched_len = len;
}

static void team_publish(orc_team_t *t, int b, int with_aux) {
  orc_agent_t *a = t->ag[b];
  int nn = orc_agent_num_neighbors(a, NULL);
  int *ids = (int *)malloc(sizeof(int) * (nn + 1));
  orc_agent_num_neighbors(a, ids);
  size_t B = (size_t)4 * t->prm.r;
  for (int q = 0; q < nn; ++q) {
    int c = ids[q];
    int np = orc_agent_public_pose_ids(a, c, NULL);
    int *frames = (int *)malloc(sizeof(int) * (np + 1));
    double *buf = (double *)malloc(sizeof(double) * B * (np + 1));
    orc_agent_public_pose_ids(a, c, frames);
    orc_agent_get_public_poses(a, c, 0, buf);
    orc_agent_update_neighbor_poses(t->ag[c], b, 0, np, frames, buf);
    if (with_aux) {
      orc_agent_get_public_poses(a, c, 1, buf);
      orc_agent_update_neighbor_poses(t->ag[c], b, 1, np, frames, buf);
    }
    free(frames); free(buf);
  }
  free(ids);
  a->publish_requested = 0;
}

void orc_team_exchange_all(orc_team_t *t) {
  for (int b = 0; b < t->N; ++b) team_publish(t, b, 1);
}

void orc_team_set_initial(orc_team_t *t, const double *T, const double *YLift) {
  int r = t->prm.r;
  for (int k = 0; k < t->N; ++k) {
    int n = orc_agent_num_poses(t->ag[k]);
    double *X = (double *)malloc(sizeof(double) * (size_t)r * 4 * n);
    orc_lift(T + (size_t)12 * t->offset[k], n, YLift, r, X);
    orc_agent_set_X(t->ag[k], X);
    free(X);
  }
  orc_team_exchange_all(t);
}

/* one global iteration: the selected robot optimizes, every other robot calls iterate(false) first
 * (PGOAgentROS.cpp:1183-1186); with acceleration the selected robot waits for their iteration-k
 * (auxiliary) poses (:136-149). */
int orc_team_iterate(orc_team_t *t) {
  int sel = t->sched[t->iter % t->sched_len];
  for (int b = 0; b < t->N; ++b) {
    if (b == sel) continue;
    orc_agent_iterate(t->ag[b], 0);
    if (t->ag[b]->publish_requested) team_publish(t, b, t->prm.acceleration);
  }
  orc_agent_iterate(t->ag[sel], 1);
  if (t->ag[sel]->publish_requested) team_publish(t, sel, t->prm.acceleration);
  t->iter++;
  return sel;
}

void orc_team_get_global_X(orc_team_t *t, double *X) {
  size_t B = (size_t)4 * t->prm.r;
  for (int k = 0; k < t->N; ++k) orc_agent_get_X(t->ag[k], X + B * t->offset[k]);
}

double orc_team_cost(orc_team_t *t) {
  int r = t->prm.r;
  double f = 0;
  for (int k = 0; k < t->N; ++k) {
    orc_agent_t *a = t->ag[k];
    for (int pass = 0; pass < 3; ++pass) {
      const orc_meas_t *arr = pass == 0 ? a->odom : (pass == 1 ? a->priv : a->shared);
      int cnt = pass == 0 ? a->nodom : (pass == 1 ? a->npriv : a->nshared);
      for (int e = 0; e < cnt; ++e) {
        const orc_meas_t *m = &arr[e];
        if (pass == 2 && (m->r1 < m->r2 ? m->r1 : m->r2) != k) continue; /* count shared edges once (owner copy) */
        const double *Xi = t->ag[m->r1]->X + (size_t)m->p1 * 4 * r, *Xj = t->ag[m->r2]->X + (size_t)m->p2 * 4 * r;
        double sr = 0, st = 0;
        for (int x = 0; x < r; ++x) {
          for (int c = 0; c < 3; ++c) {
            double v = Xj[c * r + x];
            for (int b = 0; b < 3; ++b) v -= Xi[b * r + x] * m->R[3 * b + c];
            sr += v * v;
          }
          double v = Xj[3 * r + x] - Xi[3 * r + x];
          for (int b = 0; b < 3; ++b) v -= Xi[b * r + x] * m->t[b];
          st += v * v;
        }
        f += 0.5 * m->weight * (m->kappa * sr + m->tau * st);
      }
    }
  }
  return f;
}

/* UPDATE_WEIGHT round (PGOAgentROS.cpp:1211-1233, 721-754, 1315-1353): every robot re-weights the edges
 * it owns, then sends shared-edge weights (float32 on the wire when weights_as_float32) to the
 * higher-ID endpoint, which applies them and clears its data matrices. */
int orc_team_update_weights(orc_team_t *t) {
  int changed = 0;
  for (int k = 0; k < t->N; ++k) orc_agent_update_measurement_weights(t->ag[k]);
  for (int k = 0; k < t->N; ++k) {
    orc_agent_t *a = t->ag[k];
    for (int e = 0; e < a->nshared; ++e) {
      const orc_meas_t *m = &a->shared[e];
      int other = (m->r1 == k) ? m->r2 : m->r1;
      if (other < k) continue;
      double w = m->weight;
      if (t->prm.weights_as_float32) w = (double)(float)w;
      if (orc_agent_set_measurement_weight(t->ag[other], m->r1, m->p1, m->r2, m->p2, w, m->fixed_weight)) ++changed;
      orc_agent_clear_data_matrices(t->ag[other]);
    }
  }
  for (int k = 0; k < t->N; ++k) team_publish(t, k, t->prm.acceleration);
  return changed;
}

/* PGOAgent::shouldTerminate() from the leader's mTeamStatus (messages delivered at once) [UPSTREAM-RECALL for the
 * body; in-tree: evaluated by the leader only, right after its own iterate(true), PGOAgentROS.cpp:206-214; the
 * robust max_num_iters rule, PGOAgentROSNode.cpp:228-232] */
int orc_team_should_terminate(const orc_team_t *t) {
  const orc_agent_t *lead = t->ag[0];
  if (lead->iter > t->prm.max_num_iters) return 1;
  if (t->prm.robust_cost_type != ORC_COST_L2 && lead->weight_update_count < t->prm.robust_opt_num_weight_updates) return 0;
  for (int k = 0; k < t->N; ++k) {
    if (t->ag[k]->state != ORC_STATE_INITIALIZED) return 0;
    if (!t->ag[k]->status.ready_to_terminate) return 0;
  }
  return 1;
}

int orc_team_run_schedule(orc_team_t *t, int max_iters, int *terminated, int *weight_rounds) {
  int done = 0, term = 0, rounds = 0;
  while (done < max_iters) {
    int sel = orc_team_iterate(t);
    ++done;
    if (sel != 0) continue; /* only the leader decides (:206) */
    if (orc_team_should_terminate(t)) { term = 1; break; }
    if (orc_agent_should_update_weights(t->ag[0])) { orc_team_update_weights(t); ++rounds; }
  }
  if (terminated) *terminated = term;
  if (weight_rounds) *weight_rounds = rounds;
  return done;
}

/* ---------------------------------------------------------------- initialisation helpers (8f-1) */
void orc_fixed_stiefel(int r, double *YLift) {
  /* deterministic r x 3 Stiefel point: first three columns of I_r (acceptable if declared) */
  memset(YLift, 0, sizeof(double) * 3 * r);
  for (int c = 0; c < 3; ++c) YLift[c * r + c] = 1.0;
}

void orc_lift(const double *T, int num_poses, const double *YLift, int r, double *X) {
  for (int i = 0; i < num_poses; ++i)
    for (int c = 0; c < 4; ++c)
      for (int a = 0; a < r; ++a) {
        double s = 0;
        for (int b = 0; b < 3; ++b) s += YLift[b * r + a] * T[(size_t)12 * i + 3 * c + b];
        X[((size_t)4 * i + c) * r + a] = s;
      }
}

/* chain the odometry edges i -> i+1 (global single-robot numbering); T is 3 x 4 num_poses col-major */
void orc_odometry_init(const orc_meas_t *m, int nm, int num_poses, double *T) {
  const orc_meas_t **odo = (const orc_meas_t **)calloc(num_poses, sizeof(void *));
  for (int e = 0; e < nm; ++e)
    if (m[e].r1 == m[e].r2 && m[e].p2 == m[e].p1 + 1 && !odo[m[e].p1]) odo[m[e].p1] = &m[e];
  memset(T, 0, sizeof(double) * 12 * (size_t)num_poses);
  T[0] = T[4] = T[8] = 1.0;
  for (int i = 0; i + 1 < num_poses; ++i) {
    const double *Ti = T + (size_t)12 * i;
    double *Tn = T + (size_t)12 * (i + 1);
    if (!odo[i]) { memcpy(Tn, Ti, sizeof(double) * 12); continue; }
    const orc_meas_t *e = odo[i];
    for (int c = 0; c < 3; ++c)
      for (int a = 0; a < 3; ++a) {
        double s = 0;
        for (int b = 0; b < 3; ++b) s += Ti[3 * b + a] * e->R[3 * b + c];
        Tn[3 * c + a] = s;
      }
    for (int a = 0; a < 3; ++a) {
      double s = Ti[9 + a];
      for (int b = 0; b < 3; ++b) s += Ti[3 * b + a] * e->t[b];
      Tn[9 + a] = s;
    }
  }
  free(odo);
}

/* two-stage chordal relaxation (Carlone et al. 2015; SE-Sync "chordal initialization"):
 * rotations from the kappa-weighted linear system with R_0 = I, then translations. */
int orc_chordal_init(const orc_meas_t *m, int nm, int num_poses, double *T) {
  int n = num_poses, r = 3;
  size_t N = (size_t)r * 4 * n;
  double *B = (double *)calloc(N, sizeof(double)), *S = (double *)calloc(N, sizeof(double));
  int rc = 0;
  for (int stage = 0; stage < 2; ++stage) {
    orc_trip_t *t = (orc_trip_t *)calloc((size_t)4 * nm + n, sizeof(orc_trip_t));
    int c = 0;
    memset(B, 0, sizeof(double) * N);
    for (int i = 0; i < n; ++i) {
      t[c].row = i; t[c].col = i;
      if (stage == 0) t[c].v[15] = 1.0; else { t[c].v[0] = t[c].v[5] = t[c].v[10] = 1.0; }
      if (i == 0) { t[c].v[0] = t[c].v[5] = t[c].v[10] = t[c].v[15] = 1.0; }
      ++c;
    }
    for (int e = 0; e < nm; ++e) {
      int i = m[e].p1, j = m[e].p2;
      double w = m[e].weight;
      if (stage == 0) {
        double k = w * m[e].kappa;
        /* k |R_j - R_i Rt|^2 : Q_ii += kI, Q_jj += kI, Q_ij += -k Rt, Q_ji += -k Rt^T */
        double D[16] = {0}, O[16] = {0}, Ot[16] = {0};
        for (int a = 0; a < 3; ++a) { D[5 * a] = k; for (int b = 0; b < 3; ++b) { O[a + 4 * b] = -k * m[e].R[3 * a + b]; Ot[b + 4 * a] = -k * m[e].R[3 * a + b]; } }
        if (i != 0) { t[c].row = i; t[c].col = i; memcpy(t[c].v, D, sizeof D); ++c; }
        if (j != 0) { t[c].row = j; t[c].col = j; memcpy(t[c].v, D, sizeof D); ++c; }
        if (i != 0 && j != 0) {
          t[c].row = j; t[c].col = i; memcpy(t[c].v, O, sizeof O); ++c;
          t[c].row = i; t[c].col = j; memcpy(t[c].v, Ot, sizeof Ot); ++c;
        } else if (i == 0 && j != 0) { /* R_0 = I: rhs_j -= I * Q_0j = -(-k Rt) */
          for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) B[((size_t)4 * j + b) * r + a] += k * m[e].R[3 * a + b];
        } else if (j == 0 && i != 0) { /* rhs_i -= I * Q_0i = k Rt^T */
          for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) B[((size_t)4 * i + b) * r + a] += k * m[e].R[3 * b + a];
        }
      } else {
        double tau = w * m[e].tau;
        const double *Ri = T + (size_t)12 * i;
        double v[3];
        for (int a = 0; a < 3; ++a) { v[a] = 0; for (int b = 0; b < 3; ++b) v[a] += Ri[3 * b + a] * m[e].t[b]; }
        if (i != 0) { t[c].row = i; t[c].col = i; t[c].v[15] = tau; ++c; for (int a = 0; a < 3; ++a) B[((size_t)4 * i + 3) * r + a] -= tau * v[a]; }
        if (j != 0) { t[c].row = j; t[c].col = j; t[c].v[15] = tau; ++c; for (int a = 0; a < 3; ++a) B[((size_t)4 * j + 3) * r + a] += tau * v[a]; }
        if (i != 0 && j != 0) {
          t[c].row = j; t[c].col = i; t[c].v[15] = -tau; ++c;
          t[c].row = i; t[c].col = j; t[c].v[15] = -tau; ++c;
        }
      }
    }
    if (stage == 0) for (int a = 0; a < 3; ++a) B[(size_t)a * r + a] = 1.0; /* pose 0 rows: identity */
    orc_bsr_t A;
    orc_bsr_from_triplets(t, c, n, &A);
    free(t);
    orc_chol_t C;
    if (orc_chol_factor(&A, 0.0, &C) != 0) { rc = -1; orc_bsr_free(&A); orc_chol_free(&C); break; }
    orc_chol_solve(&C, B, r, S);
    orc_chol_free(&C);
    orc_bsr_free(&A);
    if (stage == 0) {
      for (int i = 0; i < n; ++i) {
        double Rm[9];
        for (int cc = 0; cc < 3; ++cc) for (int a = 0; a < 3; ++a) Rm[3 * cc + a] = S[((size_t)4 * i + cc) * r + a];
        orc_project_rotation(Rm, T + (size_t)12 * i);
      }
    } else {
      for (int i = 0; i < n; ++i) for (int a = 0; a < 3; ++a) T[(size_t)12 * i + 9 + a] = S[((size_t)4 * i + 3) * r + a];
    }
  }
  free(B); free(S);
  return rc;
}

double orc_measurement_cost(const orc_meas_t *m, int nm, const double *X, int r) {
  double f = 0;
  for (int e = 0; e < nm; ++e) {
    const double *Xi = X + (size_t)m[e].p1 * 4 * r, *Xj = X + (size_t)m[e].p2 * 4 * r;
    double sr = 0, st = 0;
    for (int x = 0; x < r; ++x) {
      for (int c = 0; c < 3; ++c) {
        double v = Xj[c * r + x];
        for (int b = 0; b < 3; ++b) v -= Xi[b * r + x] * m[e].R[3 * b + c];
        sr += v * v;
      }
      double v = Xj[3 * r + x] - Xi[3 * r + x];
      for (int b = 0; b < 3; ++b) v -= Xi[b * r + x] * m[e].t[b];
      st += v * v;
    }
    f += 0.5 * m[e].weight * (m[e].kappa * sr + m[e].tau * st);
  }
  return f;
}
