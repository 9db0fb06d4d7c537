// twolevel.hip -- set-up of the two-level preconditioner (twolevel.h) on the device, for all re-assembled agents of a
// team at once:
//   1. A_ii = (Q + shift I)[I_i, I_i] of every subdomain and A_SS of the separator, gathered densely from the block-CSR
//   2. D_i = A_ii^-1                      one batched blocked inversion over every subdomain of every agent
//   3. E_i = D_i A_iS                     (m_i x 4 a_i: only the separator poses coupled to subdomain i)
//   4. Sc  = A_SS - sum_i A_Si E_i        the Schur complement
//   5. Sc^-1                              one batched inversion over the agents
//   6. the per-workgroup slabs of the apply (twolevel_dev.h): D_i / -E_i rows, then W_i = -Sc^-1[:, adj_i] E_i^T or
//      Sc^-1 rows, interleaved as [row pair][column][2]
// The arithmetic is the reference's preconditioner solve (P = chol(Q + eps I)) carried out as a
// block elimination; the result equals the dense inverse to round-off.  Set-up only: simple kernels, no tuning beyond
// coalesced access.
#include "team_internal.h"
#include "twolevel.h"
#include "twolevel_dev.h"

using namespace dpgo;

namespace dpgo {

struct TLSetupAgent {
  const int *rowptr, *col;
  const double *qval;
  const int *blk_of, *lidx;      // [n] block of a pose (subdomain, or P = separator) and its index inside the block
  const int *subptr, *subposes;  // CSR block -> poses (blocks 0 .. P-1: subdomains, block P: separator)
  const int *adjptr, *adjlist;   // per subdomain: positions of the separator poses coupled to it
  const long long *Doff, *Eoff;  // per block / subdomain: offsets (doubles) into A / D and into E
  double *A, *D, *E;
  int P, ns, n, pad;
  double shift;
  TLDev tl;
  double *slabs_rw;
};

// A_bb for block `blk` of agent `ai` (jobs[z] = (ai, blk)): one workgroup per pose of the block (a block column)
__global__ __launch_bounds__(64) void k_tl_gather(const TLSetupAgent *ags, const int2 *jobs) {
  const int2 jb = jobs[blockIdx.z];
  const TLSetupAgent &g = ags[jb.x];
  const int blk = jb.y, p0 = g.subptr[blk], cnt = g.subptr[blk + 1] - p0;
  const int lj = blockIdx.x;
  if (lj >= cnt) return;
  const int j = g.subposes[p0 + lj], m = 4 * cnt;
  double *A = g.A + g.Doff[blk];
  for (int p = g.rowptr[j] + (int)threadIdx.x / 16; p < g.rowptr[j + 1]; p += 4) {
    const int e = threadIdx.x % 16, cp = e % 4, c = e / 4;
    const int i = g.col[p];
    if (g.blk_of[i] != blk) continue;
    double v = g.qval[(size_t)16 * p + e];
    if (i == j && cp == c) v += g.shift;
    A[(size_t)(4 * lj + c) * m + 4 * g.lidx[i] + cp] = v;
  }
}

// E_i[:, 4k .. 4k+3] = D_i A_i,s  for the k-th separator pose s coupled to subdomain i (jobs[z] = (ai, i), blockIdx.y = k)
__global__ __launch_bounds__(256) void k_tl_E(const TLSetupAgent *ags, const int2 *jobs) {
  const int2 jb = jobs[blockIdx.z];
  const TLSetupAgent &g = ags[jb.x];
  const int i = jb.y, k = blockIdx.y;
  const int a0 = g.adjptr[i], na = g.adjptr[i + 1] - a0;
  if (k >= na) return;
  const int m = 4 * (g.subptr[i + 1] - g.subptr[i]);
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= m) return;
  const int s = g.subposes[g.subptr[g.P] + g.adjlist[a0 + k]];  // the separator pose
  const double *D = g.D + g.Doff[i];
  double acc[4] = {0, 0, 0, 0};
  for (int p = g.rowptr[s]; p < g.rowptr[s + 1]; ++p) {  // row s lists (q, Q_qs): rows of pose q, columns of pose s
    const int q = g.col[p];
    if (g.blk_of[q] != i) continue;
    const int lq = g.lidx[q];
    const double *val = g.qval + (size_t)16 * p;
#pragma unroll
    for (int cp = 0; cp < 4; ++cp) {
      const double d = D[(size_t)(4 * lq + cp) * m + r];
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] += d * val[cp + 4 * c];
    }
  }
  double *E = g.E + g.Eoff[i];
#pragma unroll
  for (int c = 0; c < 4; ++c) E[(size_t)(4 * k + c) * m + r] = acc[c];
}

// Sc[rows of separator pose s', :] -= A_s'i E_i for every subdomain i coupled to s' (one workgroup per s'; the entries
// of row s' are taken in order, a barrier between them: two subdomains may share separator columns)
__global__ __launch_bounds__(256) void k_tl_schur(const TLSetupAgent *ags, const int2 *jobs) {
  const int2 jb = jobs[blockIdx.z];
  const TLSetupAgent &g = ags[jb.x];
  const int ls = blockIdx.x;
  if (ls >= g.ns) return;
  const int NS4 = 4 * g.ns;
  const int sp = g.subposes[g.subptr[g.P] + ls];
  double *Sc = g.A + g.Doff[g.P];
  for (int p = g.rowptr[sp]; p < g.rowptr[sp + 1]; ++p) {
    const int q = g.col[p], i = g.blk_of[q];
    if (i == g.P) continue;
    const int lq = g.lidx[q], m = 4 * (g.subptr[i + 1] - g.subptr[i]);
    const int a0 = g.adjptr[i], ncol = 4 * (g.adjptr[i + 1] - a0);
    const double *E = g.E + g.Eoff[i];
    const double *val = g.qval + (size_t)16 * p;  // Q_{q s'}[cp, c'] = val[cp + 4 c'];  A_{s' q} is its transpose
    for (int kk = threadIdx.x; kk < ncol; kk += 256) {
      const int col = 4 * g.adjlist[a0 + kk / 4] + (kk & 3);
      double e[4];
#pragma unroll
      for (int cp = 0; cp < 4; ++cp) e[cp] = E[(size_t)kk * m + 4 * lq + cp];
#pragma unroll
      for (int cr = 0; cr < 4; ++cr) {
        double s = 0;
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) s += val[cp + 4 * cr] * e[cp];
        Sc[(size_t)col * NS4 + 4 * ls + cr] -= s;
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ int tl_find(const int *list, int count, int key) {
  int lo = 0, hi = count;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (list[mid] < key) lo = mid + 1; else hi = mid; }
  return (lo < count && list[lo] == key) ? lo : -1;
}

// the slab of apply-workgroup b (jobs[z] = (ai, 0), blockIdx.x = b)
__global__ __launch_bounds__(256) void k_tl_pack(const TLSetupAgent *ags, const int2 *jobs) {
  const int2 jb = jobs[blockIdx.z];
  const TLSetupAgent &g = ags[jb.x];
  const int b = blockIdx.x;
  if (b >= g.tl.nwg) return;
  const TLWg w = g.tl.wg[b];
  const int *rp = g.tl.rowpose + (size_t)b * g.tl.rp_stride;
  double *slab = g.slabs_rw + w.slab_off;
  const int npre = 2 * w.pre_cnt, NS4 = 4 * g.ns;
  const bool sepwg = b < g.tl.nA;                     // producer: rows of -E_i
  const bool has_post = !sepwg || g.tl.prod_post;     // ... whose slab carries separator rows only for the one-launch solve
  const double *Sci = g.D + g.Doff[g.P];
  // ---- rows that meet the input vector: D_i (interior workgroup) or -E_i (producer)
  for (int x = threadIdx.x; x < npre * 16; x += 256) {
    const int t = x >> 4, c8 = (x >> 1) & 7, h = x & 1, lp = c8 >> 2, c = c8 & 3;
    const int own = w.own[lp], rpose = rp[t >> 1], rloc = 4 * g.lidx[rpose] + 2 * (t & 1) + h;
    double v = 0;
    if (own >= 0) {
      const int i = g.blk_of[rpose];
      const int m = 4 * (g.subptr[i + 1] - g.subptr[i]);
      if (!sepwg) {
        if (g.blk_of[own] == i) v = g.D[g.Doff[i] + (size_t)(4 * g.lidx[own] + c) * m + rloc];
      } else {
        const int a0 = g.adjptr[i];
        const int k = tl_find(g.adjlist + a0, g.adjptr[i + 1] - a0, g.lidx[own]);
        if (k >= 0) v = -g.E[g.Eoff[i] + (size_t)(4 * k + c) * m + rloc];
      }
    }
    slab[x] = v;
  }
  // ---- separator rows: Sc^-1 (separator workgroup) or W_i = -Sc^-1[:, adj_i] E_i^T (interior workgroup).  One lane per
  // separator row and all 8 columns of the workgroup: an entry of Sc^-1 is fetched once for the (up to) 8 products it
  // enters (the entries of E are the same for every lane: broadcast loads); the sums run in the order of the adjacency
  // list, as before (one lane per (row, column) re-fetched Sc^-1 for every column: 0.8 ms of a 6.6 ms UPDATE_WEIGHT round)
  double *post = slab + (size_t)npre * 16;
  if (!has_post) return;
  int sub[2], m_[2], a0_[2], ncol_[2];
  const double *Eo[2];
  bool septype[2];
#pragma unroll
  for (int lp = 0; lp < 2; ++lp) {
    const int own = w.own[lp];
    sub[lp] = -1; m_[lp] = 0; a0_[lp] = 0; ncol_[lp] = 0; Eo[lp] = g.E; septype[lp] = false;
    if (own < 0) continue;
    if (g.blk_of[own] == g.P) { septype[lp] = true; continue; }
    const int i = g.blk_of[own];
    sub[lp] = i;
    m_[lp] = 4 * (g.subptr[i + 1] - g.subptr[i]);
    a0_[lp] = g.adjptr[i];
    ncol_[lp] = 4 * (g.adjptr[i + 1] - a0_[lp]);
    Eo[lp] = g.E + g.Eoff[i] + 4 * g.lidx[own];
  }
  const bool paired = sub[0] >= 0 && sub[0] == sub[1];  // both poses interior to the same subdomain: one pass over Sc^-1
  for (int srow = threadIdx.x; srow < NS4; srow += 256) {
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (paired) {
      double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int kk = 0; kk < ncol_[0]; ++kk) {
        const double sc = Sci[(size_t)(4 * g.adjlist[a0_[0] + kk / 4] + (kk & 3)) * NS4 + srow];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          s[c] += sc * Eo[0][(size_t)kk * m_[0] + c];
          s[4 + c] += sc * Eo[1][(size_t)kk * m_[0] + c];
        }
      }
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8) v[c8] = -s[c8];
    } else {
#pragma unroll
      for (int lp = 0; lp < 2; ++lp) {
        const int own = w.own[lp];
        if (own < 0) continue;
        if (septype[lp]) {  // a separator pose: its column of Sc^-1
#pragma unroll
          for (int c = 0; c < 4; ++c) v[4 * lp + c] = Sci[(size_t)(4 * g.lidx[own] + c) * NS4 + srow];
        } else {
          double s[4] = {0, 0, 0, 0};
          for (int kk = 0; kk < ncol_[lp]; ++kk) {
            const double sc = Sci[(size_t)(4 * g.adjlist[a0_[lp] + kk / 4] + (kk & 3)) * NS4 + srow];
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] += sc * Eo[lp][(size_t)kk * m_[lp] + c];
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) v[4 * lp + c] = -s[c];
        }
      }
    }
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) post[(size_t)(srow >> 1) * 16 + c8 * 2 + (srow & 1)] = v[c8];
  }
}

}  // namespace dpgo

namespace dpgo_host {

// plan + layout of one agent -> host images of the descriptors (uploaded by tl_build)
struct TLHostLayout {
  std::vector<int> blk_of, lidx, subptr, subposes, adjptr, adjlist, rowpose;
  std::vector<long long> Doff, Eoff;
  std::vector<TLWg> wg;
  size_t d_total = 0, e_total = 0, slab_total = 0;
  int rp_stride = 1, max_cnt = 0, max_adj = 0;
};

static TLHostLayout tl_layout(const TLPlan &pl) {
  TLHostLayout L;
  const int P = (int)pl.sub.size(), n = pl.n;
  L.blk_of.assign(n, P);
  L.lidx.assign(n, 0);
  L.subptr.assign(P + 2, 0);
  for (int i = 0; i < P; ++i) {
    for (size_t q = 0; q < pl.sub[i].size(); ++q) { L.blk_of[pl.sub[i][q]] = i; L.lidx[pl.sub[i][q]] = (int)q; }
    L.subposes.insert(L.subposes.end(), pl.sub[i].begin(), pl.sub[i].end());
    L.subptr[i + 1] = (int)L.subposes.size();
    L.max_cnt = std::max(L.max_cnt, (int)pl.sub[i].size());
  }
  for (int s = 0; s < pl.ns; ++s) L.lidx[pl.sep[s]] = s;
  L.subposes.insert(L.subposes.end(), pl.sep.begin(), pl.sep.end());
  L.subptr[P + 1] = (int)L.subposes.size();
  L.max_cnt = std::max(L.max_cnt, pl.ns);
  L.adjptr.assign(P + 1, 0);
  L.Doff.assign(P + 1, 0);
  L.Eoff.assign(P + 1, 0);
  for (int i = 0; i < P; ++i) {
    L.adjlist.insert(L.adjlist.end(), pl.adj_sep[i].begin(), pl.adj_sep[i].end());
    L.adjptr[i + 1] = (int)L.adjlist.size();
    L.max_adj = std::max(L.max_adj, (int)pl.adj_sep[i].size());
    const size_t m = 4 * pl.sub[i].size();
    L.Doff[i] = (long long)L.d_total; L.d_total += m * m;
    L.Eoff[i] = (long long)L.e_total; L.e_total += m * 4 * pl.adj_sep[i].size();
  }
  L.Doff[P] = (long long)L.d_total;
  L.d_total += (size_t)16 * pl.ns * pl.ns;
  if (L.adjlist.empty()) L.adjlist.push_back(0);
  // workgroups of the apply
  std::vector<std::vector<int>> rows(pl.nwg);
  for (int b = 0; b < pl.nwg; ++b) { rows[b] = tl_pre_rows(pl, b); L.rp_stride = std::max(L.rp_stride, (int)rows[b].size()); }
  L.rowpose.assign((size_t)pl.nwg * L.rp_stride, 0);
  L.wg.resize(pl.nwg);
  for (int b = 0; b < pl.nwg; ++b) {
    TLWg &w = L.wg[b];
    w.own[0] = pl.order[2 * b]; w.own[1] = pl.order[2 * b + 1];
    w.pre_cnt = (int)rows[b].size(); w.pad1 = 0;
    w.sep0 = (b < pl.nA && w.own[0] >= 0) ? pl.sep_index[w.own[0]] : 0;
    w.slab_off = (long long)L.slab_total;
    L.slab_total += (size_t)16 * (2 * rows[b].size() + ((b >= pl.nA || pl.prod_post) ? 2 * pl.ns : 0));
    const int fill = w.own[0] >= 0 ? w.own[0] : 0;
    for (int q = 0; q < L.rp_stride; ++q) L.rowpose[(size_t)b * L.rp_stride + q] = q < (int)rows[b].size() ? rows[b][q] : fill;
  }
  return L;
}

// Does the two-level form pay for this agent?  Measured on MI355X (profiles/experiments/scale.py, sphere2500 split
// 8 / 5 / 4 / 3 / 2 / 1 ways, us per apply, dense | two-level): 312 poses 8.1 | 11.2, 500: 10.2 | 15.6, 625: 16.6 | 20.4,
// 833: 24.0 | 28.7, 1250: 41.9 | 47.1, 2500: 134 | 101.  The exchange between phase A and phase B (publish, counter,
// re-read: ~8 us of dependent round trips across XCDs) costs more than streaming a dense inverse of up to ~200 MB, so
// the automatic mode takes the two-level form only beyond that -- and wherever the dense inverse does not fit at all.
bool tl_worthwhile(const TLPlan &pl) {
  const double dense = 8.0 * 16.0 * (double)pl.n * (double)pl.n;
  return dense > 256.0e6 && pl.bytes < 0.5 * dense && !pl.sub.empty();
}

// device memory the two-level form of an agent holds for good (slabs + tables), and its set-up scratch
double tl_resident_bytes(const TLPlan &pl) { return pl.bytes + 4.0 * pl.nwg * 2.0 * 1024 + 4096; }

int tl_build(dpgo_team *t, const std::vector<Agent *> &agents) {
  if (agents.empty()) return 0;
  hipStream_t s = t->stream;
  const int na = (int)agents.size();
  std::vector<const TLHostLayout *> lay(na);
  std::vector<TLSetupAgent> setup(na);
  size_t scratch = 0;
  for (int k = 0; k < na; ++k) {
    Agent &a = *agents[k];
    // (the layout depends on the dissection alone: kept with the agent across weight updates)
    if (!a.tl_layout_cache || a.tl_layout_serial != a.tl_plan_serial) {
      a.tl_layout_cache = std::make_shared<TLHostLayout>(tl_layout(a.tl_plan));
      a.tl_layout_serial = a.tl_plan_serial;
    }
    lay[k] = static_cast<const TLHostLayout *>(a.tl_layout_cache.get());
    scratch += 3 * lay[k]->d_total + lay[k]->e_total;
  }
  if (t->d_tmp.alloc(scratch)) { set_err("two-level preconditioner: scratch allocation failed"); return DPGO_ERR; }
  HIPC(hipMemsetAsync(t->d_tmp.p, 0, sizeof(double) * scratch, s));
  size_t off = 0;
  std::vector<int2> sub_jobs, sep_jobs, agent_jobs;
  int max_cnt = 0, max_adj = 0, max_ns = 0, max_wg = 0;
  std::vector<double *> invA, invW, invM, invA2, invW2, invM2;
  std::vector<int> invN, invN2;
  for (int k = 0; k < na; ++k) {
    Agent &a = *agents[k];
    const TLPlan &pl = a.tl_plan;
    const TLHostLayout &L = *lay[k];
    const int P = (int)pl.sub.size();
    // the tables depend on the dissection alone: a weight update (same sparsity pattern, same plan) refills the slabs only
    const bool tables_current = a.tl_tables_serial == a.tl_plan_serial && a.d_tl_wg.p;
    if ((!tables_current &&
         (a.d_tl_blk.upload(L.blk_of, s) || a.d_tl_lidx.upload(L.lidx, s) || a.d_tl_subptr.upload(L.subptr, s) ||
          a.d_tl_subposes.upload(L.subposes, s) || a.d_tl_adjptr.upload(L.adjptr, s) || a.d_tl_adjlist.upload(L.adjlist, s) ||
          a.d_tl_doff.upload(L.Doff, s) || a.d_tl_eoff.upload(L.Eoff, s) || a.d_tl_wg.upload(L.wg, s) ||
          a.d_tl_rowpose.upload(L.rowpose, s))) ||
        a.d_tl_slabs.alloc(L.slab_total) || a.d_tl_u.alloc((size_t)std::max(1, 4 * pl.ns) * t->prm.r)) {
      set_err("two-level preconditioner: device allocation / upload failed");
      return DPGO_ERR;
    }
    a.tl_tables_serial = a.tl_plan_serial;
    if (!a.d_tl_flag.p) {
      if (a.d_tl_flag.alloc(TL_FLAG_WORDS)) { set_err("two-level preconditioner: device allocation failed"); return DPGO_ERR; }
      HIPC(hipMemsetAsync(a.d_tl_flag.p, 0, sizeof(unsigned long long) * TL_FLAG_WORDS, s));
    }
    TLSetupAgent &g = setup[k];
    g.rowptr = a.d_rowptr.p; g.col = a.d_col.p; g.qval = a.d_qval.p;
    g.blk_of = a.d_tl_blk.p; g.lidx = a.d_tl_lidx.p; g.subptr = a.d_tl_subptr.p; g.subposes = a.d_tl_subposes.p;
    g.adjptr = a.d_tl_adjptr.p; g.adjlist = a.d_tl_adjlist.p; g.Doff = a.d_tl_doff.p; g.Eoff = a.d_tl_eoff.p;
    double *base = t->d_tmp.p + off;
    g.A = base; double *Wk = base + L.d_total; g.D = base + 2 * L.d_total; g.E = base + 3 * L.d_total;
    off += 3 * L.d_total + L.e_total;
    g.P = P; g.ns = pl.ns; g.n = pl.n; g.pad = 0; g.shift = t->prm.precond_shift;
    TLDev &tl = a.dev.tl;
    tl.ns = pl.ns; tl.nwg = pl.nwg; tl.nA = pl.nA; tl.rp_stride = L.rp_stride; tl.nS2 = pl.nS2; tl.prod_post = pl.prod_post ? 1 : 0;
    tl.wg = a.d_tl_wg.p; tl.rowpose = a.d_tl_rowpose.p; tl.slabs = a.d_tl_slabs.p; tl.u = a.d_tl_u.p;
    tl.flag = a.d_tl_flag.p; tl.err = t->h_bar_err;
    g.tl = tl; g.slabs_rw = a.d_tl_slabs.p;
    for (int i = 0; i < P; ++i) {
      sub_jobs.push_back(make_int2(k, i));
      const size_t o = (size_t)L.Doff[i];
      invA.push_back(g.A + o); invW.push_back(Wk + o); invM.push_back(g.D + o); invN.push_back(4 * (int)pl.sub[i].size());
    }
    if (pl.ns > 0) {
      sep_jobs.push_back(make_int2(k, P));
      const size_t o = (size_t)L.Doff[P];
      invA2.push_back(g.A + o); invW2.push_back(Wk + o); invM2.push_back(g.D + o); invN2.push_back(4 * pl.ns);
    }
    agent_jobs.push_back(make_int2(k, 0));
    max_cnt = std::max(max_cnt, L.max_cnt); max_adj = std::max(max_adj, L.max_adj);
    max_ns = std::max(max_ns, pl.ns); max_wg = std::max(max_wg, pl.nwg);
  }
  DevBuf<TLSetupAgent> d_setup;
  DevBuf<int2> d_sub_jobs, d_sep_jobs, d_agent_jobs;
  if (d_setup.upload(setup, s) || d_sub_jobs.upload(sub_jobs, s) || d_sep_jobs.upload(sep_jobs, s) || d_agent_jobs.upload(agent_jobs, s)) {
    set_err("two-level preconditioner: job upload failed");
    return DPGO_ERR;
  }
  auto fail_msg = [&](const char *what, int fail) {
    set_err(std::string("two-level preconditioner: Cholesky of ") + what + " failed at pivot " + std::to_string(fail & 0xffffff) +
            " (matrix " + std::to_string(fail >> 24) + " of the batch)");
    return DPGO_ERR;
  };
  // 1. + 2.  subdomain blocks and their inverses (z <= 65535 per launch)
  for (size_t j0 = 0; j0 < sub_jobs.size(); j0 += 32768) {
    const unsigned nz = (unsigned)std::min<size_t>(32768, sub_jobs.size() - j0);
    hipLaunchKernelGGL(k_tl_gather, dim3(max_cnt, 1, nz), dim3(64), 0, s, d_setup.p, d_sub_jobs.p + j0);
  }
  if (!sep_jobs.empty())
    hipLaunchKernelGGL(k_tl_gather, dim3(max_cnt, 1, (unsigned)sep_jobs.size()), dim3(64), 0, s, d_setup.p, d_sep_jobs.p);
  for (size_t j0 = 0; j0 < invN.size(); j0 += 32768) {
    const int cnt = (int)std::min<size_t>(32768, invN.size() - j0);
    const int fail = dense_spd_inverse_batched(s, cnt, invA.data() + j0, invW.data() + j0, invM.data() + j0, invN.data() + j0, true);
    if (fail) return fail_msg("a subdomain block of Q + shift I", fail);
  }
  if (!sep_jobs.empty()) {
    // 3. + 4.  coupling blocks and the Schur complement
    if (max_adj > 0)
      for (size_t j0 = 0; j0 < sub_jobs.size(); j0 += 32768) {
        const unsigned nz = (unsigned)std::min<size_t>(32768, sub_jobs.size() - j0);
        hipLaunchKernelGGL(k_tl_E, dim3((4 * max_cnt + 255) / 256, max_adj, nz), dim3(256), 0, s, d_setup.p, d_sub_jobs.p + j0);
      }
    hipLaunchKernelGGL(k_tl_schur, dim3(max_ns, 1, (unsigned)sep_jobs.size()), dim3(256), 0, s, d_setup.p, d_sep_jobs.p);
    // 5.
    const int fail = dense_spd_inverse_batched(s, (int)invN2.size(), invA2.data(), invW2.data(), invM2.data(), invN2.data(), true);
    if (fail) return fail_msg("the Schur complement", fail);
  }
  // 6.
  hipLaunchKernelGGL(k_tl_pack, dim3(max_wg, 1, (unsigned)na), dim3(256), 0, s, d_setup.p, d_agent_jobs.p);
  HIPC(hipStreamSynchronize(s));
  HIPC(hipGetLastError());
  return 0;
}

}  // namespace dpgo_host
