// twolevel.h -- the exact two-level (nested-dissection / Schur-complement) form of the preconditioner
//   z = v (Q + shift I)^-1          (the reference solves with a sparse Cholesky factor)
// The poses of an agent are split into p subdomains I_1 .. I_p that touch each other only through a vertex separator
// S.  With A = Q + shift I, D_i = A_ii^-1, E_i = D_i A_iS, Sc = A_SS - sum_i A_Si E_i (the Schur complement) and
// U = [-E; I]:
//   A^-1 = blockdiag(D_1 .. D_p, 0) + U Sc^-1 U^T
// -- the same operator as the dense inverse to round-off, stored as p small dense inverses plus the dense
// N_S x N block W = Sc^-1 U^T: sum_i m_i^2 + N_S N doubles instead of N^2 (8 MB instead of 32 MB at 500 poses on
// sphere2500, 0.87 GB instead of 4.2 GB for cubicle as one agent, 4 GB for a 60 000-pose chain whose dense inverse
// would take 460 GB).  The apply is two dense products with one exchange in between:
//   phase A   u = v_S - sum_i v_i E_i                     (the workgroups that own the separator poses)
//   phase B   z_i = v_i D_i + u W_i ,   z_S = u Sc^-1     (every workgroup, for the two poses it owns)
// Host side here: the dissection (plan) and the layout of the per-workgroup slabs; device side in twolevel.hip.
#pragma once
#include <cstddef>
#include <vector>

namespace dpgo_host {

struct TLPlan {
  int n = 0, ns = 0;                       // poses, separator poses
  std::vector<int> sep;                    // separator poses (ascending)
  std::vector<std::vector<int>> sub;       // subdomains: ascending pose lists
  std::vector<int> sub_of;                 // [n] subdomain of a pose, -1 = separator
  std::vector<int> sep_index;              // [n] position in `sep`, -1 = interior
  std::vector<std::vector<int>> adj_sep;   // per subdomain: separator positions (ascending) coupled to it
  std::vector<std::vector<int>> adj_sub;   // per separator pose: subdomains coupled to it (ascending)
  // workgroup layout of the apply: `order` lists the poses in ownership order (separator first, then subdomain by
  // subdomain), each part padded with -1 to an even number of slots; workgroup b owns slots 2b, 2b+1.  The first nA = ns workgroups own ONE separator pose each (they are dispatched first: every other
  // workgroup waits for what they publish, none of them waits for anybody).
  std::vector<int> order;
  int nwg = 0, nA = 0, nS2 = 0;             // workgroups; producers; workgroups that own the separator poses' columns
  bool prod_post = false;                  // producers' slabs also hold their pose's column of Sc^-1 (one-launch RTR solve)
  double bytes = 0;                        // bytes one apply streams (slabs)
};

// rowptr / col: block-CSR pattern of Q (row j lists the poses coupled to j, the diagonal included).
// max_sub <= 0: try a ladder of subdomain sizes and keep the plan that streams the fewest bytes.
TLPlan tl_make_plan(int n, const std::vector<int> &rowptr, const std::vector<int> &col, int max_sub = 0, int fit_pairs = 0,
                    int fit_wg = 0);
// poses whose rows of the input vector enter workgroup b's product before the exchange (see twolevel_plan.cpp)
std::vector<int> tl_pre_rows(const TLPlan &pl, int b);
// the most poses any workgroup's slab meets before the exchange (sizes the LDS slab of the one-launch RTR solve)
int tl_max_pre_poses(const TLPlan &pl);

}  // namespace dpgo_host
