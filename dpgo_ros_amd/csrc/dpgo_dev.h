// dpgo_dev.h -- device-visible descriptors shared by the kernels and the host runtime.
#pragma once
#include <hip/hip_runtime.h>

namespace dpgo {

// work vectors of one agent, each r x 4n doubles resident in HBM
enum Buf {
  B_X = 0, B_XPREV, B_Y, B_V, B_G,
  B_EGRAD, B_GF, B_Z, B_ETA, B_R0, B_R1, B_D0, B_D1, B_HD,
  B_X2, B_EGRAD2, B_GF2, B_HETA, B_T0, B_T1, B_T2,
  // second copies of X and Y: the one-launch iteration (step_fused.hip) reads the poses of one parity and writes the
  // other, so that no workgroup of a launch overwrites what another still reads.  Both sit B_ALT vectors behind their
  // primaries (one displacement turns any pointer into a primary array into its twin)
  B_XALT, B_ALTPAD_, B_YALT,
  NBUF
};
// carried rows of the one-launch iteration (step_fused.hip): the evaluation point of the agent two iterations ahead
// ([pose][4r]) and the row products formed from it ([entry][pose]).  They live inside one captured run of one-launch
// iterations only, in two work vectors of the trust-region solver that an RGD team never touches meanwhile.
constexpr int B_CARRY_Y = B_R0, B_CARRY_W = B_R1, B_CARRY_X = B_D0;  // (W / X: row products and point of the PUBLIC poses, [entry][public pose])
constexpr int B_CARRY_G = B_D1;  // tangent projection of the row products at the point, [pose][4r]: the gradient of a pose without shared edges
constexpr int FE_CARRY_IN = 1, FE_CARRY_W = 2, FE_CARRY_Y = 4;
constexpr int FE_MAX_EDGES = 160;  // shared edges of an agent whose operands the one-launch iteration keeps in LDS (46 KB at r = 5)
constexpr int B_ALT = B_XALT - B_X;
static_assert(B_YALT - B_Y == B_ALT, "X and Y twins share one displacement");

// one shared (inter-robot) edge as seen by the G assembly: G_i[:,c] -= sum_cp Xn[:,cp] coef[cp+4c]
struct SharedEdgeDev {
  int lpose;  // local pose that receives the contribution
  int slot;   // index into the neighbour-pose slab
  int src_agent_local;  // local index of the neighbour agent in this team, or -1 if remote
  int src_frame;
  int src_robot, pad_;  // global id of the neighbour (locates an imported peer, see dpgo_team_import_peer)
  const double *src[2];  // the neighbour's pose in its agent's X ([0]) / Y ([1]) array when co-resident -- or when the
                         // neighbour lives in another process whose arrays were imported (IPC, peer access over xGMI) --
                         // else null: one load instead of the agents[src].buf[...] descriptor round trip
  const double *src_yalt;  // the same pose in the neighbour's B_YALT array (co-resident neighbours only, else null)
  double coef[16];
};

// per-edge record for residual / cost evaluation
struct EdgeDev {
  int i_local, j_local;    // local pose index or -1
  int i_slot, j_slot;      // neighbour slab slot when not local
  double R[9];             // row-major
  double t[3];
  double kappa, tau, weight;
  int count_in_cost;       // 1 if this agent owns the edge for the global cost
  int pad;
};

// trust-region / tCG scalar state (device resident, ping-pong [2])
struct RtrState {
  double f1, ngf, Delta;
  double z_r, d_Pd, e_Pd, e_Pe, norm_r0, alpha;
  double f_init, gn_init;
  int outer_it, outer_done, tcg_active, tcg_j, tcg_status;
  int need_init, pad0;  // need_init: the next tCG has not been set up yet (after begin / accept)
  int hv_count, pc_count, tcg_total, accepted, outer_count;
  int tcg_o[4];  // (Hess-vec, step) launch pairs the tCG of outer iteration o consumed: sizes the blind patterns of the next solve
};

struct NestState {
  double gamma, alpha;
  int iter;     // mIterationNumber of this agent
  int pad;
};

// two-level (nested-dissection / Schur-complement) form of the preconditioner, see twolevel.h.  Workgroup b of an apply
// owns the poses own[0], own[1] (8 columns of the operator); its slab holds, as [row pair][column][2] doubles in the
// order the lanes consume them, first the rows that meet the input vector itself (`pre_cnt` poses, listed in
// rowpose[b * rp_stride ..]: the poses of its subdomain(s), D_i -- or, for a workgroup that owns separator poses, the
// poses of the adjacent subdomains, -E_i), then the 4 ns separator rows that meet u (W_i, or Sc^-1).
struct TLWg {
  int own[2];          // poses (-1: padding slot)
  int pre_cnt, sep0;   // sep0: position of own[0] in the separator (rows of u it publishes), workgroups that own separator poses
  long long slab_off;  // doubles from TLDev::slabs
  long long pad1;
};
struct TLDev {
  int ns, nwg, nA, rp_stride;  // separator poses, workgroups, producers (the first nA: one separator pose each), row-list stride
  int nS2, prod_post;          // workgroups [nA, nA + nS2) own the separator poses' columns; producers' slabs also hold them
  const TLWg *wg;
  const int *rowpose;
  const double *slabs;
  double *u;                   // [4 ns][R]: v_S - sum_i v_i E_i, published by the first nA workgroups of an apply
  unsigned long long *flag;    // [0] producers that have published (this apply), [16] workgroups past the exchange (the
                               // last one clears both), [32] completed exchanges of the persistent solve kernel
  int *err;                    // pinned host word raised when an exchange timed out
};

struct AgentDev {
  int id, n, nb, N4;
  int npub, nshared, nnp, nedges;
  const int *rowptr;          // full block-CSR (dense assembly, read-back)
  const int *col;
  const double *qval;
  int ell_w, soa_w;           // ELL part of the same matrix: slot-major [ell_w][n], padded with (j, 0)
  const int *ell_col;
  const double *ell_val;
  const double *soa_val;      // the ELL blocks once more as [tile of 64 poses][soa_w][16-byte chunk][lane], soa_w = max(ell_w, 5)
  const int *soa_col;         // slots (null when a row has a tail); their columns [tile][soa_w][lane]: padded slots hold (j, 0)
  const int *trowptr;         // CSR tail: blocks beyond ell_w of long rows
  const int *tcol;
  const double *tval;
  const int *pub_index;       // [n] index into pub_pose/pub_ptr or -1
  const int *pose_eptr;       // [n+1] CSR of `se` by local pose: the shared edges of pose j are [pose_eptr[j], pose_eptr[j+1])
                              // (the evaluation finds them with one round trip instead of pub_index -> pub_ptr)
  int fe_eptr[5], fe_code_ok; // pub_ptr[min(64 g, npub)], g = 0 .. 4: the shared edges of the 64 public poses a gradient wave of
                              // the carried one-launch iteration finishes, known without a round trip
  unsigned fe_code[FE_MAX_EDGES / 2];  // where the neighbour pose of shared edge e lives, 16 bits each (two per word): frame |
                              // local agent << 12 (teams whose neighbours are all co-resident, <= 160 edges: fe_code_ok) -- the
                              // carried one-launch iteration turns it into an address without a descriptor round trip
  unsigned char fe_ord[32];   // dense agents of <= 512 poses (one 2048-row pass of the stream): the order in which the 64-row
                              // chunks (16 poses each) of a column of M meet the vector -- chunks whose poses are all PRIVATE
                              // (no shared edge) first, the others behind them, each group ascending.  Every kernel that
                              // streams such an agent's M follows it (k_precond, k_step_fe), so their sums stay bitwise
                              // equal; the deep-carried one-launch iteration forms the private part one launch early
  int fe_npriv, fe_pad_;      // how many leading entries of fe_ord are private chunks
  const double *M;            // dense (Q + shift I)^-1, N4 x N4 column-major (symmetric); null: block-Jacobi agent
  const double *Dinv;         // block-Jacobi agents: the inverted 4 x 4 diagonal blocks of Q + shift I, [n][16] column-major
  TLDev tl;                   // two-level agents (tl.nwg > 0; M and Dinv null)
  const int *pub_pose;        // [npub] local poses that own >= 1 shared edge
  const int *pub_ptr;         // [npub+1] CSR into se
  const SharedEdgeDev *se;    // [nshared] sorted by lpose
  const double *fe_coef;      // [nshared][16]: their coefficients, packed
  const EdgeDev *edges;       // [nedges] all measurements of this agent
  double *nbr[2];             // neighbour pose slabs (0 main, 1 auxiliary), [nnp][4r]
  double *buf[NBUF];
  double *part;               // partial-sum scratch, [PART_STRIDE * MAX_PART]
  RtrState *st;               // [2]
  NestState *nest;            // [1]
  double *scal;               // [16] misc scalars: 5 owned-edge cost, 6 gamma' of the running iteration (published by k_nest_pre),
                              // 8..11 record of the last RGD line search (k_ls_apply: back-offs, accepted, f, step)
  double *resid;              // [nedges] residual scratch
};

constexpr int PART_STRIDE = 8;   // doubles per block of partials
constexpr int MAX_PART = 32768;  // max blocks contributing partials: agents of up to 65536 poses (8 columns per block of
                                 // the preconditioner kernels); 2 MB per region
// regions of AgentDev::part (each MAX_PART * PART_STRIDE doubles)
constexpr int PART_A = 0;                             // SpMM-type kernels (eval / Hess-vec)
constexpr int PART_B = MAX_PART * PART_STRIDE;        // preconditioner-type kernels ([2] |X - XPrev|^2 of the fused RGD step)
constexpr int PART_C = 2 * MAX_PART * PART_STRIDE;    // outer-step / initial evaluation
constexpr int PART_D = 3 * MAX_PART * PART_STRIDE;    // per-pose kernels: |X - XPrev|^2 partials (per 64-pose tile; one double per
                                                      // pose after a look-ahead Nesterov step)
constexpr int PART_E = 4 * MAX_PART * PART_STRIDE;    // |X - XPrev|^2 tiles of the agent's last iterate(true) (k_status, opt != 0):
                                                      // nothing an iterate(false) launches writes here
constexpr int PART_TOTAL = 5 * MAX_PART * PART_STRIDE;

constexpr int LOOKAHEAD_MAX_AGENTS = 8;  // look-ahead Nesterov steps locate a pose's agent with one 9-int fetch

// carried one-launch iteration: the Y array of every local agent and its pose count (by value in the launch)
struct FeBases {
  const double *ybase[LOOKAHEAD_MAX_AGENTS];
  int npose[LOOKAHEAD_MAX_AGENTS];
  double *part[LOOKAHEAD_MAX_AGENTS];  // the agents' partial-sum scratch (step_deep.hip: the look-aheads that leave a status)
};

// deep-carried one-launch iteration (step_deep.hip): what a launch needs of the agents one and two iterations ahead (by value)
constexpr int FD_IN = 1, FD_P = 2, FD_W = 4, FD_Y = 8;
// ... and what the last iterations of a run leave for a status query (k_precond's ahead bits 3 and 2): FD_STATS -- this step
// leaves its X2 snapshot and |X - XPrev|^2 (PART_B[2]); FD_LASTAT -- the look-aheads leave XPrev and |Y' - X|^2 per pose (PART_D)
constexpr int FD_STATS = 16, FD_LASTAT = 32;
struct FdNext {
  // d = the agent of the next iteration: the private part of its product is formed here
  const double *Md;            // its dense inverse
  const double *Gd;            // its B_CARRY_G (chunk-ordered)
  int N4d, nblk_d;
  unsigned char ord_d[32];     // its chunk order (AgentDev::fe_ord)
  // e = the agent two iterations ahead: its row products are formed here
  int n_e, soa_w_e, npub_e, pad_;
  const int *soa_col_e;
  const double *soa_val_e;
  const int *pub_index_e;
  const double *Ye;            // B_CARRY_Y: its evaluation point, [pose][4r]
  double *We, *Xe, *Ge;        // B_CARRY_W / B_CARRY_X ([entry][public pose]) and B_CARRY_G (chunk-ordered [place][16 poses][4r])
  unsigned char ord_e[32];
};

struct TeamDev {
  int num_agents;
  int sched_len;
  int iter;          // global iteration counter (device copy)
  int restart_interval;
  int cur_sel;       // agent selected in the running iteration (published by the first kernel)
  int stats_sel;     // agent of the iteration that just finished: its final statistics are evaluated by k_stats_nest
  int next_sel;      // pipelined RGD iterations: agent of iteration k+1, published by the step kernel of iteration k
  int pad0;
  int pose_prefix[LOOKAHEAD_MAX_AGENTS + 1];  // exclusive prefix sums of the agents' pose counts (teams of <= 8 agents)
  int pad1;
  const int *sched;  // [sched_len] local agent index selected at iteration k % sched_len
  const int *group_ptr;      // colour classes of the agent graph (CSR): agents of one class share no edge,
  const int *group_members;  // so they may take their block update in the same launches (blockIdx.y)
};

constexpr int SEL_GROUP0 = -16;  // sel <= SEL_GROUP0 selects colour class (SEL_GROUP0 - sel), member blockIdx.y
constexpr int SEL_ALL = SEL_GROUP0 - 4096;  // every local agent, agent blockIdx.y (the class behind the colouring, without its
                                            // two dependent table look-ups at the head of every kernel of a lockstep tick)

}  // namespace dpgo
