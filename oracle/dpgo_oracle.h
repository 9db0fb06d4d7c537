/*
 * dpgo_oracle.h -- CPU restatement of the RBCD hot path (TEST INFRASTRUCTURE ONLY).
 *
 * *** PARITY UNPINNED ***  The arithmetic of this path lives in the external,
 * un-vendored, un-pinned library mit-acl/dpgo (README.md:9-13, CMakeLists.txt:6,151-154 of the
 * reference) which cannot be built in this image (needs Eigen3, ROPTLIB, SuiteSparse).  The
 * reference's own tests (tests/testUtils.cpp:16-70) pin no solver output.  This file therefore
 * restates the *published* algorithm (Tian et al., T-RO 2021 "Distributed certifiably correct
 * pose-graph optimization"; Absil/Baker/Gallivan RTR-tCG; Yang et al. GNC) and anchors on the
 * reference's call sites, wire layouts and schedule.  It is pinned by
 *   (1) first-principles known-answer tests (finite differences, symmetry, descent, gauge),
 *   (2) the SE-Sync published optima of the bundled datasets (sphere2500 2f*=1687.0, torus 24227),
 *   (3) an independent numpy implementation (oracle/np_crosscheck.py -> tests/golden/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this library.
 * The product (dpgo_ros_amd/) never links, imports or calls it.
 *
 * Conventions (d = 3 fixed, k = d+1 = 4, r = relaxation rank >= 3):
 *   X is r x (k n), column-major:  X[(4*i + c)*r + a],  c<3 -> column c of Y_i, c=3 -> p_i.
 *   (identical to Eigen::MatrixXd(r, 4n).data(), and -- per pose -- to the *transpose* of the
 *   row-major MatrixMsg payload of src/utils.cpp:20-61.)
 */
#ifndef DPGO_ORACLE_H
#define DPGO_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int r1, p1, r2, p2;      /* src robot/frame, dst robot/frame (utils.cpp:128-133) */
  double R[9];             /* row-major 3x3 rotation of the relative measurement */
  double t[3];
  double kappa, tau;       /* rotation / translation precision */
  double weight;           /* GNC weight in [0,1] */
  int fixed_weight;        /* utils.cpp:147-149: odometry => fixed */
  int is_known_inlier;
} orc_meas_t;

enum { ORC_METHOD_RTR = 0, ORC_METHOD_RGD = 1 };
enum { ORC_COST_L2 = 0, ORC_COST_L1 = 1, ORC_COST_HUBER = 2, ORC_COST_TLS = 3, ORC_COST_GM = 4, ORC_COST_GNC_TLS = 5 };
enum { ORC_STATE_WAIT_FOR_DATA = 0, ORC_STATE_WAIT_FOR_INITIALIZATION = 1, ORC_STATE_INITIALIZED = 2 };
enum { ORC_WEIGHT_LIBRARY = 0, ORC_WEIGHT_WRAPPER = 1 };

typedef struct {
  int d, r, num_robots;
  int method;                 /* PGOAgentROSNode.cpp:82-93 */
  double rgd_stepsize;        /* :96 */
  int rgd_use_preconditioner; /* :97 */
  int rtr_iterations;         /* :98 */
  int rtr_tcg_iterations;     /* :99 */
  double gradnorm_tol;        /* :100 */
  double rtr_initial_radius;  /* [UPSTREAM-RECALL] 100 */
  double rtr_max_radius;      /* [UPSTREAM-RECALL] 5 x initial */
  double precond_shift;       /* [UPSTREAM-RECALL] 0.1 : P = Q + shift*I */
  int acceleration;           /* :126 */
  int restart_interval;       /* :130 */
  double rel_change_tol;      /* :134 */
  int max_num_iters;          /* :228-232 */
  int robust_cost_type;       /* :175-188: ORC_COST_* */
  double gnc_barc, gnc_mu_step, gnc_init_mu; /* :196-210 */
  int robust_opt_num_weight_updates, robust_opt_inner_iters; /* :212-221 */
  double robust_opt_min_convergence_ratio;
  int weights_as_float32;     /* msg/RelativeMeasurementWeights.msg:8 wire rounding */
  int robust_opt_num_resets;  /* PGOAgentROSNode.cpp:213: set by the wrapper, never read by it; semantics live in the
                               * absent library and are not recoverable here -> carried, no effect */
  int precond_mode;           /* 0 / 1: sparse Cholesky of Q + shift I (the reference's preconditioner);
                               * 2: block-Jacobi -- the inverses of the 4x4 diagonal blocks of Q + shift I.  NOT the
                               * reference's preconditioner: the product's declared O(n)-memory fallback for agents whose
                               * dense inverse does not fit, restated here so that the fallback has a checker too */
  int status_every_iterate;   /* 0 (default): relativeChange / readyToTerminate are refreshed by iterate(true) only
                               * [UPSTREAM-RECALL, and the only rule under which the leader's check at
                               * PGOAgentROS.cpp:206-214 is meaningful without acceleration: iterate(false) leaves
                               * X = XPrev]; 1: refreshed by every iterate (round-1 behaviour) */
  /* RGD with a backtracking (Armijo) line search on the retraction curve [UPSTREAM-RECALL: the library has a line-search
   * variant of its gradient step; no call site of the wrapper selects it (src/PGOAgentROSNode.cpp:86-97 writes method,
   * RGD_stepsize, RGD_use_preconditioner only) and its constants are not recoverable here -- the textbook rule (Absil,
   * Mahony, Sepulchre 2008, Def. 4.2.2) with its parameters as fields]:  trial step t_j = rgd_stepsize * shrink^j,
   * j = 0 .. max_backoffs; the first j with  f(Retr_x(-t_j d)) <= f(x) - sigma t_j <grad f(x), d>  is taken (d = the
   * preconditioned gradient or the gradient); if none qualifies x stays put. */
  int rgd_line_search;        /* 0 (default): fixed step */
  int rgd_ls_max_backoffs;    /* 7: eight trial steps */
  double rgd_ls_shrink;       /* 0.5 */
  double rgd_ls_sigma;        /* 1e-4 */
  double tls_threshold;       /* [UPSTREAM-RECALL] RobustCostParameters::TLSThreshold, 10 */
  double huber_threshold;     /* [UPSTREAM-RECALL] RobustCostParameters::HuberThreshold, 3 */
} orc_params_t;

typedef struct {
  int success;
  double f_init, f_opt, gradnorm_init, gradnorm_opt;
  int rtr_outer_iters, tcg_iters_total, hessvec_count, precond_count, accepted;
  int ls_backoffs;  /* line search: back-offs before the accepted step (max_backoffs + 1 and accepted = 0: none qualified) */
} orc_opt_result_t;

typedef struct {
  int agent_id, state, instance_number, iteration_number, ready_to_terminate;
  double relative_change;
} orc_status_t;

void orc_default_params(orc_params_t *p, int r, int num_robots);

/* ---------- IO: g2o / tunnels csv / partition ---------- */
/* returns number of measurements (or <0), allocates *out (free with orc_free). */
int orc_read_g2o(const char *path, int weight_mode, orc_meas_t **out, int *num_poses);
int orc_read_measurements_csv(const char *path, int weight_mode, orc_meas_t **out);
/* PGODatasetPublisherNode.cpp:84-135: contiguous blocks; rewrites r1,p1,r2,p2 in place. */
void orc_partition(orc_meas_t *m, int nm, int num_poses, int num_robots, int weight_mode);
void orc_free(void *p);

/* ---------- small manifold ops (exposed for unit parity) ---------- */
void orc_project_stiefel(const double *A, int r, double *out);         /* polar factor of r x 3 */
void orc_project_rotation(const double *A, double *out);               /* 3x3 col-major -> SO(3) */
void orc_retract_qf(const double *Y, const double *eta, int r, double *out); /* qf(Y+eta), r x 3 */
void orc_project_manifold(const double *Xin, int r, int n, double *Xout);
void orc_tangent_project(const double *X, const double *V, int r, int n, double *out);
void orc_retract(const double *X, const double *eta, int r, int n, double *out);

/* ---------- agent ---------- */
typedef struct orc_agent orc_agent_t;
orc_agent_t *orc_agent_new(int id, const orc_params_t *p);
void orc_agent_free(orc_agent_t *a);
void orc_agent_add_measurement(orc_agent_t *a, const orc_meas_t *m);
int orc_agent_num_poses(const orc_agent_t *a);
int orc_agent_num_measurements(const orc_agent_t *a, int *odom, int *priv, int *shared);
int orc_agent_num_neighbors(const orc_agent_t *a, int *ids /* may be NULL */);
/* number of (distinct) public poses shared with neighbor; fills local frame ids (sorted). */
int orc_agent_public_pose_ids(const orc_agent_t *a, int nbr, int *frames);
/* neighbor poses this agent needs from nbr (sorted frames). */
int orc_agent_neighbor_pose_ids(const orc_agent_t *a, int nbr, int *frames);
void orc_agent_set_X(orc_agent_t *a, const double *X);     /* r x 4n; also X->Y,V; state=INITIALIZED */
void orc_agent_get_X(const orc_agent_t *a, double *X);
void orc_agent_get_Y(const orc_agent_t *a, double *Y);
void orc_agent_get_V(const orc_agent_t *a, double *V);
/* copy own public poses (r x 4 each, column-major) for nbr into out, order = public_pose_ids */
int orc_agent_get_public_poses(const orc_agent_t *a, int nbr, int aux, double *out);
/* updateNeighborPoses / updateAuxNeighborPoses (PGOAgentROS.cpp:1276,1278) */
void orc_agent_update_neighbor_poses(orc_agent_t *a, int nbr, int aux, int count, const int *frames,
                                     const double *poses);
int orc_agent_iterate(orc_agent_t *a, int do_optimization);  /* PGOAgentROS.cpp:160,1185 */
void orc_agent_get_status(const orc_agent_t *a, orc_status_t *s);
void orc_agent_get_opt_result(const orc_agent_t *a, orc_opt_result_t *r);
int orc_agent_iteration_number(const orc_agent_t *a);
/* problem-level evaluation at an arbitrary point (uses current neighbor poses; aux selects set) */
void orc_agent_build_problem(orc_agent_t *a, int aux);
double orc_agent_eval(orc_agent_t *a, const double *X, double *egrad /*nullable*/, double *rgrad /*nullable*/);
void orc_agent_hessvec(orc_agent_t *a, const double *X, const double *eta, double *out);
void orc_agent_precondition(orc_agent_t *a, const double *X, const double *V, double *out);
/* raw data matrices for parity: Q as BSR (rowptr n+1, col nb, val 16*nb col-major blocks), G r x 4n */
int orc_agent_get_Q(orc_agent_t *a, int *rowptr, int *col, double *val);
void orc_agent_get_G(orc_agent_t *a, double *G);
/* robust path */
int orc_agent_compute_residual(const orc_agent_t *a, const orc_meas_t *m, double *res); /* :1049 */
double orc_robust_weight(const orc_agent_t *a, double residual);                        /* :1050 */
void orc_agent_update_measurement_weights(orc_agent_t *a);                              /* :1218 */
int orc_agent_set_measurement_weight(orc_agent_t *a, int r1, int p1, int r2, int p2, double w, int fixed); /* :1341 */
int orc_agent_get_measurements(const orc_agent_t *a, orc_meas_t *out /* nullable */);
int orc_agent_should_update_weights(const orc_agent_t *a);                              /* :210 */
void orc_agent_clear_data_matrices(orc_agent_t *a);                                     /* :1351 */
double orc_error_threshold_at_quantile(double q, int dim);                              /* Node.cpp:201 */

/* ---------- team driver: synchronous schedule of PGOAgentROS.cpp:129-220,443-504,1161-1189 ---------- */
typedef struct orc_team orc_team_t;
orc_team_t *orc_team_new(const orc_meas_t *m, int nm, int num_poses, const orc_params_t *p,
                         int weight_mode);
void orc_team_free(orc_team_t *t);
orc_agent_t *orc_team_agent(orc_team_t *t, int id);
void orc_team_set_schedule(orc_team_t *t, const int *order, int len); /* default round robin */
/* initial guess: global trajectory T (3 x 4 num_poses col-major), lifted by YLift (r x 3 col-major) */
void orc_team_set_initial(orc_team_t *t, const double *T, const double *YLift);
void orc_team_exchange_all(orc_team_t *t);
int orc_team_iterate(orc_team_t *t);       /* one global RBCD iteration; returns selected agent */
double orc_team_cost(orc_team_t *t);       /* f of the concatenated iterate: sum_e w/2 (k|.|^2 + tau|.|^2) */
int orc_team_iteration(const orc_team_t *t);
void orc_team_get_global_X(orc_team_t *t, double *X); /* r x 4 num_poses */
int orc_team_update_weights(orc_team_t *t); /* UPDATE_WEIGHT round (:1211-1233); returns #changed */
/* PGOAgent::shouldTerminate() as the leader (robot 0) evaluates it from mTeamStatus (PGOAgentROS.cpp:208) */
int orc_team_should_terminate(const orc_team_t *t);
/* the synchronous schedule as the wrapper runs it (PGOAgentROS.cpp:129-220): iterate; after every iteration in which
 * the leader optimized: TERMINATE if shouldTerminate(), else an UPDATE_WEIGHT round if shouldUpdateMeasurementWeights(),
 * else pass the token.  Returns the number of iterations executed (<= max_iters); *terminated, *weight_rounds optional. */
int orc_team_run_schedule(orc_team_t *t, int max_iters, int *terminated, int *weight_rounds);

/* ---------- initialisation + centralized reference solve ---------- */
void orc_odometry_init(const orc_meas_t *m, int nm, int num_poses, double *T);
int orc_chordal_init(const orc_meas_t *m, int nm, int num_poses, double *T);
void orc_lift(const double *T, int num_poses, const double *YLift, int r, double *X);
void orc_fixed_stiefel(int r, double *YLift); /* deterministic r x 3 Stiefel point */
double orc_measurement_cost(const orc_meas_t *m, int nm, const double *X, int r); /* single-robot ids */
/* robust frame alignment from n candidate transforms (3x4 column-major each): 0 ok, 1 not enough inliers */
int orc_robust_frame_alignment(const double *Tc, int n, double max_rot_rad, double max_trans, int min_inliers,
                               double *T_out, int *inlier /* nullable */);

#ifdef __cplusplus
}
#endif
#endif
