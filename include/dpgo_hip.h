/*
 * dpgo_hip.h -- C-ABI of the MI355X-native RBCD hot path (libdpgo_hip.so).
 *
 * This is the drop-in boundary underneath the C++ facade `namespace DPGO` (headers under include/DPGO/) that
 * the ROS wrapper of mit-acl/dpgo_ros subclasses (include/dpgo_ros/PGOAgentROS.h:121
 * `class PGOAgentROS : public PGOAgent`).  Each entry point cites the reference call site it
 * serves (paths relative to /root/reference).  Plain pointers and sizes only; no torch types.
 *
 * Conventions (d = 3, k = 4, r = relaxation rank in [3,8]):
 *   X is r x (4 n) column-major: X[(4*i + c)*r + a]; c<3 -> column c of Y_i, c=3 -> p_i.
 *   A "pose" is the r x 4 block of 4r contiguous doubles (Eigen column-major LiftedPose::getData()).
 *   All arithmetic is fp64 on the device.  Host pointers unless the name says _device.
 * Return codes: 0 = ok, >0 = not available yet (the facade maps to `false`), <0 = fatal (CHECK).
 */
#ifndef DPGO_HIP_H
#define DPGO_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* RelativeSEMeasurement(r1,r2,p1,p2,R,t,kappa,tau) + weight, fixedWeight
 * (src/utils.cpp:109-149; src/PGOAgentROS.cpp:740-745) */
typedef struct {
  int r1, p1, r2, p2;
  double R[9]; /* row-major 3x3 */
  double t[3];
  double kappa, tau, weight;
  int fixed_weight;
  int is_known_inlier;
} dpgo_measurement_t;

enum { DPGO_METHOD_RTR = 0, DPGO_METHOD_RGD = 1 };       /* ROptParameters::ROptMethod */
/* RobustCostParameters::Type, the six names src/PGOAgentROSNode.cpp:178-188 accepts.  Weight functions w(r) of the residual
 * r (dpgo_agent_robust_weight; [UPSTREAM-RECALL] mit-acl/dpgo src/DPGO_robust.cpp, the library is absent from the mount):
 *   L2 1 | L1 1 / r | Huber r < huber_threshold ? 1 : huber_threshold / r | TLS r < tls_threshold ? 1 : 0 |
 *   GM 1 / (1 + r^2)^2 | GNC_TLS Yang et al. RA-L 2020 with the running mu */
enum { DPGO_COST_L2 = 0, DPGO_COST_L1 = 1, DPGO_COST_HUBER = 2, DPGO_COST_TLS = 3, DPGO_COST_GM = 4, DPGO_COST_GNC_TLS = 5 };
enum { DPGO_WAIT_FOR_DATA = 0, DPGO_WAIT_FOR_INITIALIZATION = 1, DPGO_INITIALIZED = 2 }; /* msg/Status.msg:1-3 */
enum { DPGO_WEIGHT_LIBRARY = 0, DPGO_WEIGHT_WRAPPER = 1 }; /* info-matrix vs kappa=1e4,tau=1e2 */
enum { DPGO_OK = 0, DPGO_NOT_READY = 1, DPGO_ERR = -1 };
/* preconditioner of the local solves.  The reference's is a sparse Cholesky solve with Q + shift I.  Two forms
 * of that SAME operator are built here: DENSE, the explicit inverse (N^2 doubles, N = 4 poses; small agents), and
 * TWO_LEVEL, the exact nested-dissection / Schur-complement form (dense inverses of p subdomains + the separator block:
 * 8 MB instead of 32 MB at 500 poses on sphere2500, 0.9 GB instead of 4.2 GB for cubicle as one agent, 4 GB for a
 * 60 000-pose chain) -- equal to round-off.  BLOCK_JACOBI (the inverses of the 4 x 4 diagonal blocks) is NOT the
 * reference's preconditioner: it runs only when asked for (precond_mode 2; restated in the oracle for its own parity
 * tests) or when neither exact form fits the device. */
enum { DPGO_PRECOND_AUTO = 0, DPGO_PRECOND_DENSE = 1, DPGO_PRECOND_BLOCK_JACOBI = 2, DPGO_PRECOND_TWO_LEVEL = 3 };
/* largest pose index dpgo_agent_add_measurements accepts (a dense preconditioner of (4n)^2 doubles is the limit
 * long before this; see dpgo_agent memory guard in DESIGN.md 3) */
#define DPGO_MAX_POSE_INDEX 1000000

/* PGOAgentParameters fields written by src/PGOAgentROSNode.cpp:80-231 */
typedef struct {
  int d, r, num_robots;
  int method;
  double rgd_stepsize;
  int rgd_use_preconditioner;
  int rtr_iterations;
  int rtr_tcg_iterations;
  double gradnorm_tol;
  double rtr_initial_radius;
  double rtr_max_radius;
  double precond_shift;
  int acceleration;
  int restart_interval;
  double rel_change_tol;
  int max_num_iters;
  int robust_cost_type;
  double gnc_barc, gnc_mu_step, gnc_init_mu;
  int robust_opt_num_weight_updates, robust_opt_inner_iters;
  double robust_opt_min_convergence_ratio;
  int weights_as_float32;
  int robust_opt_num_resets;  /* src/PGOAgentROSNode.cpp:213: written by the wrapper, never read by it; its semantics live
                               * in the absent library -> carried, validated (>= 0), no effect */
  int precond_mode;           /* DPGO_PRECOND_*: 0 automatic (dense inverse for small agents, the two-level form where it
                               * streams less than half the dense bytes), 1 dense inverse (error if it does not fit),
                               * 2 block-Jacobi, 3 two-level for every agent */
  int status_every_iterate;   /* 0 (default): relativeChange / readyToTerminate describe the last iterate(true) of the
                               * agent [UPSTREAM-RECALL]; 1: refreshed by every iterate (round-1 behaviour) */
  /* RGD with a backtracking (Armijo) line search on the retraction curve (north_star "RTR/RGD line search"; SURVEY App. B:
   * "a backtracking variant exists" [UPSTREAM-RECALL]; no wrapper call site selects it -- src/PGOAgentROSNode.cpp:86-97
   * writes method, RGD_stepsize, RGD_use_preconditioner only -- so it is off unless asked for).  Trial steps
   * t_j = rgd_stepsize * rgd_ls_shrink^j, j = 0 .. rgd_ls_max_backoffs (at most 7: eight trial points, all evaluated by
   * one pass over Q); the first j with f(Retr_x(-t_j d)) <= f(x) - rgd_ls_sigma t_j <grad f(x), d> is taken; none: x
   * stays put (dpgo_opt_result_t.accepted = 0). */
  int rgd_line_search;
  int rgd_ls_max_backoffs;
  double rgd_ls_shrink;
  double rgd_ls_sigma;
  double tls_threshold;       /* RobustCostParameters::TLSThreshold [UPSTREAM-RECALL: 10]; no wrapper call site writes it */
  double huber_threshold;     /* RobustCostParameters::HuberThreshold [UPSTREAM-RECALL: 3]; likewise */
} dpgo_params_t;

/* mLocalOptResult.{success,fInit,fOpt,gradNormInit,gradNormOpt} (src/PGOAgentROS.cpp:169-172) */
typedef struct {
  int success;
  double f_init, f_opt, gradnorm_init, gradnorm_opt;
  int rtr_outer_iters, tcg_iters_total, hessvec_count, precond_count, accepted;
  int ls_backoffs; /* RGD line search: back-offs before the accepted step */
} dpgo_opt_result_t;

/* PGOAgentStatus(agentID,state,instanceNumber,iterationNumber,readyToTerminate,relativeChange)
 * (src/utils.cpp:262-281; tests/testUtils.cpp:56-65) */
typedef struct {
  int agent_id, state, instance_number, iteration_number, ready_to_terminate;
  double relative_change;
} dpgo_status_t;

void dpgo_default_params(dpgo_params_t *p, int r, int num_robots);
const char *dpgo_last_error(void);

/* ---- dataset input: read_g2o_file (src/PGODatasetPublisherNode.cpp:80),
 *      PGOLogger::loadMeasurements (:168), contiguous partition (:84-135) ---- */
int dpgo_read_g2o(const char *path, int weight_mode, dpgo_measurement_t **out, int *num_poses);
int dpgo_read_measurements_csv(const char *path, int weight_mode, dpgo_measurement_t **out);
/* robust inter-robot frame alignment: n candidate transforms T_world_robot (3x4 column-major
 * each, one per shared loop closure with an initialised neighbour; updateNeighborPoses ->
 * initializeInGlobalFrame, src/PGOAgentROS.cpp:1276, 353-358) -> GNC-TLS rotation averaging (chordal metric,
 * threshold = chord of max_rotation_error_rad) then GNC-TLS translation averaging on the rotation inliers.
 * DPGO_OK, or DPGO_NOT_READY when fewer than min_inliers (robustInitMinInliers, Node.cpp:150) agree.
 * inlier: n flags or NULL.  Host arithmetic only. */
int dpgo_robust_frame_alignment(const double *T_candidates, int n, double max_rotation_error_rad,
                                double max_translation_error, int min_inliers, double *T_out, int *inlier);
/* robust local initialisation (InitializationMethod::GNC_TLS, src/PGOAgentROSNode.cpp:111-112): single-robot
 * GNC-TLS solve on the device with r = d = 3 -- odometry chain as the initial guess and fixed at weight 1,
 * loop closures re-weighted robust_opt_num_weight_updates times, robust_opt_inner_iters RTR iterations
 * per weighting (gnc_*, rtr_*, gradnorm_tol of `gnc` are used; r / num_robots / method / acceleration are
 * overridden).  T_out: 3x4 column-major per pose; weights_out: nm final weights in input order, or NULL. */
int dpgo_robust_local_init(int device, const dpgo_measurement_t *m, int nm, int num_poses, const dpgo_params_t *gnc,
                           double *T_out, double *weights_out);
/* writers (the PGOLogger::logMeasurements / logTrajectory role): the CSV of loadMeasurements
 * (data/tunnels/robot0/measurements.csv:1, weights and inlier flags included so GNC results round-trip), g2o
 * with isotropic information blocks (global index = robot_offsets[robot] + frame; NULL = single robot; T =
 * 3x4 column-major poses or NULL for edges only), and a trajectory CSV "pose_index,qx,qy,qz,qw,tx,ty,tz".
 * Return the number of records written, -1 on I/O failure. */
int dpgo_write_measurements_csv(const char *path, const dpgo_measurement_t *m, int nm);
int dpgo_write_g2o(const char *path, const dpgo_measurement_t *m, int nm, const double *T, int num_poses,
                   const int *robot_offsets);
int dpgo_write_trajectory_csv(const char *path, const double *T, int num_poses);
void dpgo_partition(dpgo_measurement_t *m, int nm, int num_poses, int num_robots, int weight_mode);
void dpgo_free(void *p);
void dpgo_odometry_init(const dpgo_measurement_t *m, int nm, int num_poses, double *T /* 3x4 per pose */);
/* two-stage chordal relaxation on the GPU (dense SPD solves), single-robot numbering; T as above.
 * PGOAgent::initialize() with InitializationMethod::Chordal (src/PGOAgentROS.cpp:348, Node.cpp:106-112) */
int dpgo_chordal_init(int device, const dpgo_measurement_t *m, int nm, int num_poses, double *T);
void dpgo_fixed_stiefel(int r, double *YLift);
void dpgo_lift(const double *T, int num_poses, const double *YLift, int r, double *X);

/* ---- a team = the agents resident on one GPU (one process per GPU) ---- */
typedef struct dpgo_team dpgo_team_t;
/* agent_ids: global robot ids hosted here.  stream: hipStream_t or NULL (library-owned stream). */
dpgo_team_t *dpgo_team_create(int device, const dpgo_params_t *p, int num_local, const int *agent_ids,
                              void *stream);
void dpgo_team_destroy(dpgo_team_t *t);
int dpgo_team_num_local(const dpgo_team_t *t);
void *dpgo_team_stream(dpgo_team_t *t);
int dpgo_team_synchronize(dpgo_team_t *t);

/* PGOAgent::addMeasurement (src/PGOAgentROS.cpp:277,1307).  Measurements not touching `id` are ignored. */
int dpgo_agent_add_measurements(dpgo_team_t *t, int id, const dpgo_measurement_t *m, int count);
int dpgo_agent_num_poses(dpgo_team_t *t, int id);                       /* num_poses() :285 */
int dpgo_agent_num_measurements(dpgo_team_t *t, int id, int *odom, int *priv, int *shared); /* :343-345 */
int dpgo_agent_get_neighbors(dpgo_team_t *t, int id, int *ids);         /* getNeighbors() :663 */
int dpgo_agent_public_pose_ids(dpgo_team_t *t, int id, int nbr, int *frames);
int dpgo_agent_neighbor_pose_ids(dpgo_team_t *t, int id, int nbr, int *frames); /* activeNeighborPublicPoseIDs :1394 */
/* initializeInGlobalFrame-equivalent entry: set the lifted iterate (r x 4n); X->XPrev,Y,V; INITIALIZED */
int dpgo_agent_set_X(dpgo_team_t *t, int id, const double *X);
/* which: 0 X, 1 Y (auxiliary), 2 V, 3 XPrev */
int dpgo_agent_get_X(dpgo_team_t *t, int id, int which, double *X);
/* getSharedPoseDictWithNeighbor / getAuxSharedPoseDictWithNeighbor (:668,:666); order = public_pose_ids */
int dpgo_agent_get_public_poses(dpgo_team_t *t, int id, int nbr, int aux, double *poses);
/* updateNeighborPoses / updateAuxNeighborPoses (:1276,:1278) */
int dpgo_agent_update_neighbor_poses(dpgo_team_t *t, int id, int nbr, int aux, int count,
                                     const int *frames, const double *poses);
/* same exchange with device buffers (packed slab, order = public_pose_ids / neighbor_pose_ids):
 * the RCCL point-to-point payload that replaces msg/PublicPoses.msg (:662-690, :1255-1284) */
int dpgo_agent_pack_public_poses_device(dpgo_team_t *t, int id, int nbr, int aux, double *dev_out);
int dpgo_agent_unpack_neighbor_poses_device(dpgo_team_t *t, int id, int nbr, int aux, const double *dev_in);

int dpgo_agent_iterate(dpgo_team_t *t, int id, int do_optimization);    /* :160 (true), :1185 (false) */
int dpgo_agent_get_status(dpgo_team_t *t, int id, dpgo_status_t *s);     /* getStatus() :616 */
int dpgo_agent_get_opt_result(dpgo_team_t *t, int id, dpgo_opt_result_t *r); /* :169-172 */
int dpgo_agent_iteration_number(dpgo_team_t *t, int id);                 /* iteration_number() :139 */
/* preconditioner the agent actually runs (DPGO_PRECOND_DENSE, _TWO_LEVEL or _BLOCK_JACOBI), after its data matrices
 * were built; <0 on error */
int dpgo_agent_preconditioner(dpgo_team_t *t, int id);
/* diagnostic, out[8]: {mode, subdomains, separator poses, workgroups of an apply, workgroups that own separator poses,
 * bytes one apply streams, bytes of the dense inverse, largest subdomain (poses)} */
int dpgo_agent_preconditioner_info(dpgo_team_t *t, int id, double *out);
/* the dissection behind the two-level form, host arithmetic only (csrc/twolevel_plan.cpp): block-CSR pattern of Q (row j
 * lists the poses coupled to j, diagonal included) -> sub_of[n] (subdomain of a pose, -1 = separator); max_sub <= 0: the
 * subdomain size that streams the fewest bytes.  info[6]: {subdomains, separator poses, workgroups, producer workgroups,
 * bytes per apply, 1 if the automatic mode would pick this form over the dense inverse} */
int dpgo_two_level_plan(int n, const int *rowptr, const int *col, int max_sub, int *sub_of, double *info);
int dpgo_agent_publish_requested(dpgo_team_t *t, int id, int clear);     /* mPublishPublicPosesRequested :109-112 */
int dpgo_agent_set_iteration_number(dpgo_team_t *t, int id, int iteration); /* mIterationNumber = ... (RECOVER, :1196) */

/* ---- QuadraticProblem surface for parity (f, EucGrad, RieGrad, Hessian, PreConditioner) ---- */
int dpgo_agent_build_problem(dpgo_team_t *t, int id, int aux);
int dpgo_agent_eval(dpgo_team_t *t, int id, const double *X, double *f, double *egrad, double *rgrad);
int dpgo_agent_hessvec(dpgo_team_t *t, int id, const double *X, const double *eta, double *out);
int dpgo_agent_precondition(dpgo_team_t *t, int id, const double *X, const double *V, double *out);
/* |z (Q + shift I) - v| / |v| for a fixed pseudo-random v and z = the device's preconditioner apply: how well the operator
 * the kernels run (dense inverse or two-level form) inverts Q + shift I (the reference solves with a Cholesky factor) */
int dpgo_agent_preconditioner_residual(dpgo_team_t *t, int id, double *rel);
int dpgo_agent_get_Q(dpgo_team_t *t, int id, int *rowptr, int *col, double *val); /* returns #blocks */
int dpgo_agent_get_G(dpgo_team_t *t, int id, double *G);

/* ---- lifted SE manifold ops on the device (host in/out), n poses of r x 4 ---- */
int dpgo_project_manifold(dpgo_team_t *t, const double *X, int n, double *out);
int dpgo_tangent_project(dpgo_team_t *t, const double *X, const double *V, int n, double *out);
int dpgo_retract(dpgo_team_t *t, const double *X, const double *eta, int n, double *out);

/* ---- robust path (src/PGOAgentROS.cpp:1218,1049,1050,1341,210,1351; Node.cpp:201) ---- */
int dpgo_agent_compute_residual(dpgo_team_t *t, int id, const dpgo_measurement_t *m, double *residual);
double dpgo_agent_robust_weight(dpgo_team_t *t, int id, double residual);
int dpgo_agent_update_measurement_weights(dpgo_team_t *t, int id);
int dpgo_agent_set_measurement_weight(dpgo_team_t *t, int id, int r1, int p1, int r2, int p2, double w, int fixed);
int dpgo_agent_get_measurements(dpgo_team_t *t, int id, dpgo_measurement_t *out);
/* the same three calls for every stored measurement at once (order of dpgo_agent_get_measurements: odometry,
 * private loop closures, shared loop closures, each in insertion order): one residual launch and one copy per
 * UPDATE_WEIGHT instead of one per loop closure (src/PGOAgentROS.cpp:1049 is called in a loop over all of them).
 * available[k] = 0 where a neighbour pose is missing.  Returns the count. */
int dpgo_agent_compute_residuals(dpgo_team_t *t, int id, double *residuals, int *available);
int dpgo_agent_set_measurement_weights(dpgo_team_t *t, int id, const double *weights, const int *fixed, int count);
/* restart of the Nesterov sequences (V = Y = X, gamma = alpha = 0) that follows a weight update */
int dpgo_agent_reset_acceleration(dpgo_team_t *t, int id);
int dpgo_agent_should_update_weights(dpgo_team_t *t, int id);
int dpgo_agent_clear_data_matrices(dpgo_team_t *t, int id);
double dpgo_error_threshold_at_quantile(double quantile, int dim);

/* ---- synchronous schedule on the device (src/PGOAgentROS.cpp:129-220,443-504,1161-1189):
 *      all agents of the problem live in this team; exchange is device-to-device ---- */
int dpgo_team_set_schedule(dpgo_team_t *t, const int *order, int len);
/* UpdateRule::Uniform (include/dpgo_ros/PGOAgentROS.h:35-41,76; src/PGOAgentROS.cpp:446-463): `length` token holders drawn
 * uniformly with replacement by the wrapper's own recipe (std::discrete_distribution over the robots, std::mt19937) from an
 * engine of the given seed (the wrapper seeds from std::random_device), installed as the schedule; order_out may be NULL */
int dpgo_team_set_uniform_schedule(dpgo_team_t *t, unsigned seed, int length, int *order_out);
int dpgo_team_set_initial(dpgo_team_t *t, const double *T, const double *YLift, const int *offsets);
int dpgo_team_exchange_all(dpgo_team_t *t);
/* run `iters` global RBCD iterations without host synchronisation (RGD: one hipGraph replay each) */
int dpgo_team_run(dpgo_team_t *t, int iters);
/* capture and instantiate, without executing anything, every hipGraph that dpgo_team_run(t, iters) would replay from
 * the current iteration counter (a first run otherwise builds them on the fly: ~0.3 ms per distinct batch size) */
int dpgo_team_prepare(dpgo_team_t *t, int iters);
/* the same iteration split around the neighbour exchange, for teams that hold only part of the agents
 * (one process per GPU): begin = iterate(false) part of every local agent; [exchange]; end = local solve
 * of `sel_id` if it lives here + bookkeeping.  sel_id is a global robot id. */
int dpgo_team_step_begin(dpgo_team_t *t, int sel_id);
int dpgo_team_step_end(dpgo_team_t *t, int sel_id);
/* colour-parallel sweeps: agents without a shared edge take their block update in the same
 * launches; identical to the sequential schedule that visits the colour classes in order (class 0 first,
 * members in id order).  One sweep = one block update of every agent.  Needs acceleration = 0. */
int dpgo_team_get_coloring(dpgo_team_t *t, int *color_of_agent); /* returns the number of classes */
/* simultaneous updates (the ASAPP configuration: preconditioned RGD, no acceleration): every local agent takes
 * one RGD step per tick in the same launches, from the neighbour poses as of the beginning of the tick -- the
 * deterministic instance of the asynchronous mode (src/PGOAgentROS.cpp:119-127) in which all clocks fire together.
 * Each agent's iteration number advances by `ticks`. */
int dpgo_team_run_simultaneous(dpgo_team_t *t, int ticks);
int dpgo_team_run_colored(dpgo_team_t *t, int sweeps);
/* the same with an explicit (global) colouring, one class at a time, for one-process-per-GPU runs */
int dpgo_team_set_groups(dpgo_team_t *t, int num_groups, const int *group_ptr, const int *member_ids);
int dpgo_team_run_group(dpgo_team_t *t, int group, int count);
int dpgo_team_iteration(dpgo_team_t *t);
/* global cost of the concatenated iterate, evaluated on the device */
int dpgo_team_cost(dpgo_team_t *t, double *f);
int dpgo_team_update_weights(dpgo_team_t *t);
/* PGOAgent::shouldTerminate() as the leader (robot 0) evaluates it from the team's statuses
 * (src/PGOAgentROS.cpp:208): 1 terminate, 0 continue, <0 error.  All robots must live in this team. */
int dpgo_team_should_terminate(dpgo_team_t *t);
/* the synchronous schedule with the leader's decisions (src/PGOAgentROS.cpp:129-220): iterate; after every iteration
 * in which the leader optimized: stop if shouldTerminate() (:208), else an UPDATE_WEIGHT round if
 * shouldUpdateMeasurementWeights() (:210), else pass the token (:213).  Returns the number of iterations executed
 * (<= max_iters) or <0; *terminated / *weight_rounds may be NULL. */
int dpgo_team_run_schedule(dpgo_team_t *t, int max_iters, int *terminated, int *weight_rounds);
/* per-iteration log (createIterationLog / logIteration / logString, src/PGOAgentROS.cpp:853-909): one CSV per
 * local robot, <directory>/dpgo_log_robot<id>.csv, the reference's header and column order -- robot_id, cluster_id,
 * num_active_robots, iteration, num_poses, bytes_received, iter_time_sec, total_time_sec, rel_change -- followed by
 * global_cost; a row after every block update of the robot (:189), the strings UPDATE_WEIGHT (:1217) and TERMINATE (:1042)
 * in every robot's file.  Written by dpgo_team_run_schedule, which runs one iteration per host round trip while a log is
 * open.  directory = NULL closes the files. */
int dpgo_team_set_iteration_log(dpgo_team_t *t, const char *directory);
/* refresh this agent's neighbour slabs from co-resident agents (device-to-device) */
int dpgo_agent_pull_local(dpgo_team_t *t, int id);
/* average HIP-event duration of one launch of a hot kernel on the team stream.
 * which: 0 dense preconditioner apply, 1 cost+gradient SpMM, 2 Hessian-vector SpMM; 9 / 10 / 11 the step kernel of the
 * pipelined sequence back to back / that sequence launched eagerly / its evaluation launches alone; 14 the one-launch
 * iteration (csrc/step_fused.hip) launched eagerly, real iterations (fails where the team cannot take that form) */
int dpgo_team_time_kernel(dpgo_team_t *t, int id, int which, int reps, double *avg_ms, double *algorithmic_bytes);
/* peer access for one-process-per-GPU runs of the asynchronous mode (src/PGOAgentROS.cpp:119-127): a robot's X / Y arrays
 * exported as a 64-byte HIP IPC handle (+ the offsets of X and Y in doubles and its pose count), imported by the
 * processes that hold its neighbours; from then on its public poses are read in place (xGMI peer loads), one-sided:
 * no PublicPoses message, no rendezvous.  The exporting process must outlive the importers' use.  Once a team has imported
 * a peer, dpgo_agent_iterate reads every neighbour that is readable in place (imported or co-resident) from its owner's
 * arrays instead of from what dpgo_agent_update_neighbor_poses last supplied. */
int dpgo_agent_export_state(dpgo_team_t *t, int id, unsigned char *handle64, long long *offset_x, long long *offset_y, int *n);
int dpgo_team_import_peer(dpgo_team_t *t, int robot_id, const unsigned char *handle64, long long offset_x, long long offset_y, int n);
/* The synchronous schedule across processes WITHOUT the host in the loop (the UPDATE token of src/PGOAgentROS.cpp:136-149,
 * 443-504, 1161-1189 on the device): every team exports a mailbox of 64-bit words (IPC handle) that the teams holding its
 * robots' neighbours import (robot_ids: the robots that live in the exporting team); dpgo_team_run_peer then enqueues
 * `iters` global iterations -- robot sel_ids[q] holds the token in the q-th -- with wait / signal kernels around the
 * launches that read a neighbour in place or overwrite what a neighbour was reading.  Every process passes the same
 * list; nothing synchronises with the host; the iterates are those of dpgo_team_step_begin / _end with messages. */
int dpgo_team_export_mailbox(dpgo_team_t *t, unsigned char *handle64);
int dpgo_team_import_mailbox(dpgo_team_t *t, const unsigned char *handle64, const int *robot_ids, int count);
int dpgo_team_run_peer(dpgo_team_t *t, const int *sel_ids, int iters);
/* ---- one process per GPU, the exchange carried by RCCL from inside the library (csrc/rank_exchange.cpp) ----
 * Replaces the ROS transport of this path: PublicPoses messages (msg/PublicPoses.msg:1-8; sent src/PGOAgentROS.cpp:662-690,
 * received :1255-1284) become packed r x 4 fp64 slabs moved by ncclSend / ncclRecv that the library enqueues on the team
 * stream; the staleness gate (:136-149, maxDelayedIterations include/dpgo_ros/PGOAgentROS.h:83) decides which slabs are
 * sent; the UPDATE token (:443-504, 1161-1189) is the list of token holders every rank is handed.
 * RCCL is bound at run time (dlopen librccl.so.1; DPGO_RCCL_LIBRARY overrides): single-GPU users never load it.
 * A communicator is an object of its own because ranks that own no robot (5 robots on 8 GPUs) still take part in its
 * creation and in the reductions.  id128: DPGO_COMM_ID_BYTES bytes obtained on ONE rank and handed to all (any channel). */
#define DPGO_COMM_ID_BYTES 128
typedef struct dpgo_comm dpgo_comm_t;
int dpgo_comm_unique_id(unsigned char *id128);
dpgo_comm_t *dpgo_comm_create(int device, const unsigned char *id128, int rank, int world); /* collective; NULL on error */
void dpgo_comm_destroy(dpgo_comm_t *c);
int dpgo_comm_rank(const dpgo_comm_t *c);
int dpgo_comm_world(const dpgo_comm_t *c);
/* which RCCL was bound: path + version code into out; returns the version code (<0 on error) */
int dpgo_comm_library(char *out, int cap);
/* small collectives (<= 64 doubles, host in/out, synchronous) on `stream` (hipStream_t or NULL) */
int dpgo_comm_allreduce_sum(dpgo_comm_t *c, void *stream, double *inout, int n);
int dpgo_comm_allreduce_max(dpgo_comm_t *c, void *stream, double *inout, int n);
/* owner_rank_of_robot[num_robots]: the rank that holds each robot (every robot of this team must map to the communicator's
 * rank).  max_delayed_iterations: the staleness gate.  loopback (world size 1 only): every neighbour pair -- co-resident
 * ones included -- exchanges through RCCL self-sends and nothing is read in place: the message path end to end on one GPU. */
int dpgo_team_attach_comm(dpgo_team_t *t, dpgo_comm_t *c, const int *owner_rank_of_robot, int max_delayed_iterations, int loopback);
int dpgo_team_detach_comm(dpgo_team_t *t);
/* every neighbour pair that crosses ranks, both directions, X and Y (after set_initial; before a cost evaluation) */
int dpgo_team_exchange_all_ranks(dpgo_team_t *t);
/* `iters` global iterations, robot sel_ids[q] holding the token in the q-th: per iteration the iterate(false) part of every
 * local robot (:1183-1186), the token holder's neighbours' public poses by ncclSend / ncclRecv (one message per pair of
 * ranks), the block update (:160).  Nothing synchronises with the host; every rank that owns a robot passes the same
 * list.  Iterates equal dpgo_team_step_begin / messages / dpgo_team_step_end bit for bit. */
int dpgo_team_run_ranks(dpgo_team_t *t, const int *sel_ids, int iters);
/* the lockstep ASAPP ticks (dpgo_team_run_simultaneous) and the classes of a colour-parallel sweep (dpgo_team_run_group, after
 * dpgo_team_set_groups with the global classes) with their boundary slabs moved by the library in the same way */
int dpgo_team_run_simultaneous_ranks(dpgo_team_t *t, int ticks);
int dpgo_team_run_group_ranks(dpgo_team_t *t, int group, int count);
/* the planning layer of the exchange replayed for ONE rank of a world, host arithmetic only (no device, no RCCL): the full
 * exchange, then `iters` iterations of the token schedule with the staleness gate.  npub[b * num_robots + a] = public poses of
 * robot b that robot a needs (0: not neighbours).  out[(1 + iters) * world * 4]: per batch and peer rank {doubles sent, doubles
 * received, hash of the sent slabs in order, hash of the received slabs in order} -- what rank r sends to p must be what p
 * receives from r (tests/test_rank_plan.py replays every rank of worlds of 2 .. 8). */
int dpgo_rank_plan_simulate(int num_robots, int world, int rank, const int *owner, const int *npub, int acceleration,
                            int max_delayed_iterations, int r, const int *sel_ids, int iters, long long *out);
/* global cost: this team's owned-edge partial sums (t may be NULL on a rank without robots) + a 1-double all-reduce */
int dpgo_comm_global_cost(dpgo_comm_t *c, dpgo_team_t *t, void *stream, double *f);
/* out[4]: point-to-point messages sent / received by this rank, bytes sent / received */
int dpgo_team_comm_counters(dpgo_team_t *t, double *out4);
/* diagnostic: hand-off words of an agent's one-launch RTR solve (rtr_fused.hip; phase stamps in trace builds) */
int dpgo_agent_read_rtr_handoff(dpgo_team_t *t, int id, unsigned long long *out, int n);
/* diagnostic: `n` doubles of an agent's device-side partial-sum scratch (csrc/dpgo_dev.h PART_*) from `offset` */
int dpgo_agent_read_partials(dpgo_team_t *t, int id, int offset, double *out, int n);
/* counters for the roofline report: launches and algorithmic bytes of the dominant kernels ([0] preconditioner applies,
 * [1] their bytes, [2] sparse evaluations, [3] their bytes, [4] iterations); diagnostics of the per-agent API: [5] host
 * microseconds between the launch of a report kernel and the arrival of its sequence word, [6] reports; [7] iterations
 * of dpgo_team_run that took the one-launch form (csrc/step_fused.hip), [8] those of them that found the row products of
 * their agent formed by the previous launch (carried rows), [9] the deep-carried ones (csrc/step_deep.hip); [10] reports
 * of the per-agent API that rode on the last launch of their iterate(true) (k_eval_report) */
int dpgo_team_get_counters(dpgo_team_t *t, double *out, int n);

#ifdef __cplusplus
}
#endif
#endif
