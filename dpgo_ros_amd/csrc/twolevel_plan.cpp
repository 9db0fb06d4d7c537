// twolevel_plan.cpp -- the dissection behind the two-level preconditioner (twolevel.h): subdomains + vertex separator of
// an agent's private pose graph, chosen so that one apply streams as few bytes as possible.  Host code, set-up only
// (once per pose-graph STRUCTURE: weight updates keep the plan).
//
// Method: recursive bisection by breadth-first level sets (George's automatic nested dissection): from a pseudo-
// peripheral pose, the level structure of a component is cut at the smallest level that leaves at least a quarter of
// the component on either side; the level is the separator, the parts are dissected again until they hold at most
// `max_sub` poses.  Separator poses whose neighbours lie in at most one subdomain are then handed back to it.
// Everything is deterministic (index order), so every rank / run derives the same plan from the same graph.
#include "twolevel.h"

#include <algorithm>
#include <cstdint>
#include <queue>

namespace dpgo_host {

namespace {

struct Graph {
  int n;
  const std::vector<int> &rowptr, &col;
};

// breadth-first levels of the component of `start` among the poses with mark[v] == tag; returns the visit order
std::vector<int> bfs(const Graph &g, const std::vector<int> &mark, int tag, int start, std::vector<int> &level) {
  std::vector<int> order;
  order.push_back(start);
  level[start] = 0;
  for (size_t h = 0; h < order.size(); ++h) {
    const int u = order[h];
    for (int p = g.rowptr[u]; p < g.rowptr[u + 1]; ++p) {
      const int v = g.col[p];
      if (v != u && mark[v] == tag && level[v] < 0) { level[v] = level[u] + 1; order.push_back(v); }
    }
  }
  return order;
}

struct Dissection {
  std::vector<std::vector<int>> sub;
  std::vector<int> sep;
};

Dissection dissect(const Graph &g, int max_sub) {
  const int n = g.n;
  Dissection out;
  // mark: -1 separator, otherwise the id of the (current) part a pose belongs to
  std::vector<int> mark(n, 0), level(n, -1);
  int next_tag = 1;
  std::vector<std::vector<int>> work;
  auto split_components = [&](const std::vector<int> &verts, int tag) {
    // connected components of the poses `verts` (all carrying `tag`): each gets a fresh tag and joins the work list
    for (int v : verts) level[v] = -1;
    for (int s : verts) {
      if (mark[s] != tag || level[s] >= 0) continue;
      std::vector<int> comp = bfs(g, mark, tag, s, level);
      const int t = next_tag++;
      for (int v : comp) mark[v] = t;
      std::sort(comp.begin(), comp.end());
      work.push_back(std::move(comp));
    }
  };
  {
    std::vector<int> all(n);
    for (int i = 0; i < n; ++i) all[i] = i;
    split_components(all, 0);
  }
  while (!work.empty()) {
    std::vector<int> comp = std::move(work.back());
    work.pop_back();
    if ((int)comp.size() <= max_sub) { out.sub.push_back(std::move(comp)); continue; }
    const int tag = mark[comp[0]];
    // pseudo-peripheral start: repeat BFS from the last pose reached
    int s = comp[0];
    std::vector<int> order;
    for (int rep = 0; rep < 4; ++rep) {
      for (int v : comp) level[v] = -1;
      order = bfs(g, mark, tag, s, level);
      const int far = order.back();
      if (far == s || rep == 3) break;
      s = far;
    }
    const int L = level[order.back()] + 1;
    std::vector<int> cnt(L, 0);
    for (int v : comp) cnt[level[v]] += 1;
    const int tot = (int)comp.size();
    int best = -1, cum = 0;
    for (int l = 0; l < L; ++l) {
      const int below = cum, above = tot - cum - cnt[l];
      cum += cnt[l];
      if (l == 0 || l == L - 1) continue;
      if (4 * std::min(below, above) < tot) continue;
      if (best < 0 || cnt[l] < cnt[best]) best = l;
    }
    if (best < 0) {
      // no balanced cut (a very short level structure): the level that holds the median pose
      int c2 = 0;
      for (int l = 0; l < L; ++l) { c2 += cnt[l]; if (2 * c2 >= tot) { best = l; break; } }
    }
    std::vector<int> rest;
    for (int v : comp) {
      if (level[v] == best) { mark[v] = -1; out.sep.push_back(v); }
      else rest.push_back(v);
    }
    if (rest.empty()) continue;  // (the whole component was one level: all of it is separator)
    split_components(rest, tag);
  }
  std::sort(out.sep.begin(), out.sep.end());
  return out;
}

// hand separator poses whose neighbours lie in at most one subdomain back to that subdomain
void thin(const Graph &g, Dissection &d, int max_sub) {
  const int n = g.n;
  std::vector<int> owner(n, -1);
  std::vector<int> size(d.sub.size());
  for (size_t i = 0; i < d.sub.size(); ++i) { size[i] = (int)d.sub[i].size(); for (int v : d.sub[i]) owner[v] = (int)i; }
  std::vector<char> is_sep(n, 0);
  for (int v : d.sep) is_sep[v] = 1;
  const int cap = max_sub + max_sub / 5;
  bool changed = true;
  while (changed) {
    changed = false;
    for (int v = 0; v < n; ++v) {
      if (!is_sep[v]) continue;
      int only = -1;
      bool many = false;
      for (int p = g.rowptr[v]; p < g.rowptr[v + 1]; ++p) {
        const int o = owner[g.col[p]];
        if (o < 0) continue;
        if (only < 0) only = o; else if (o != only) { many = true; break; }
      }
      if (many || only < 0 || size[only] >= cap) continue;
      owner[v] = only; size[only] += 1; is_sep[v] = 0; changed = true;
    }
  }
  for (auto &s : d.sub) s.clear();
  d.sep.clear();
  for (int v = 0; v < n; ++v) { if (owner[v] >= 0) d.sub[owner[v]].push_back(v); else d.sep.push_back(v); }
  d.sub.erase(std::remove_if(d.sub.begin(), d.sub.end(), [](const std::vector<int> &s) { return s.empty(); }), d.sub.end());
}

TLPlan finish_plan(const Graph &g, Dissection d) {
  TLPlan pl;
  const int n = g.n;
  pl.n = n;
  // subdomains in order of their first pose: neighbouring poses of the trajectory end up in neighbouring workgroups
  std::sort(d.sub.begin(), d.sub.end(), [](const std::vector<int> &a, const std::vector<int> &b) { return a[0] < b[0]; });
  pl.sub = std::move(d.sub);
  pl.sep = std::move(d.sep);
  pl.ns = (int)pl.sep.size();
  pl.sub_of.assign(n, -1);
  pl.sep_index.assign(n, -1);
  for (size_t i = 0; i < pl.sub.size(); ++i) for (int v : pl.sub[i]) pl.sub_of[v] = (int)i;
  for (int s = 0; s < pl.ns; ++s) pl.sep_index[pl.sep[s]] = s;
  pl.adj_sep.assign(pl.sub.size(), {});
  pl.adj_sub.assign(pl.ns, {});
  for (size_t i = 0; i < pl.sub.size(); ++i) {
    std::vector<int> &a = pl.adj_sep[i];
    for (int v : pl.sub[i])
      for (int p = g.rowptr[v]; p < g.rowptr[v + 1]; ++p) { const int s = pl.sep_index[g.col[p]]; if (s >= 0) a.push_back(s); }
    std::sort(a.begin(), a.end());
    a.erase(std::unique(a.begin(), a.end()), a.end());
    for (int s : a) pl.adj_sub[s].push_back((int)i);
  }
  // Workgroups of an apply, in launch order (workgroup b owns slots 2b, 2b+1 of `order`):
  //   [0, nA)          PRODUCERS, one per separator pose: they form that pose's entry of u from the subdomains adjacent
  //                    to it, publish it and leave -- they never wait for anybody, so whatever part of the grid is
  //                    resident the exchange completes (a producer that waited for the other producers could starve
  //                    them of the slots they need: 636 producers on cubicle as one agent, ~512 resident);
  //   [nA, nA + nS2)   the separator poses again, two per workgroup: their columns of Sc^-1 against u (no rows before
  //                    the exchange);
  //   the rest         the subdomains, each padded to an even number of slots (no workgroup straddles two of them).
  for (int v : pl.sep) { pl.order.push_back(v); pl.order.push_back(-1); }
  pl.nA = (int)pl.order.size() / 2;
  for (int v : pl.sep) pl.order.push_back(v);
  if (pl.order.size() & 1) pl.order.push_back(-1);
  pl.nS2 = (int)pl.order.size() / 2 - pl.nA;
  for (auto &s : pl.sub) {
    for (int v : s) pl.order.push_back(v);
    if (pl.order.size() & 1) pl.order.push_back(-1);
  }
  pl.nwg = (int)pl.order.size() / 2;
  // the one-launch RTR solve (rtr_fused.hip) runs producers that also own their pose's column of Sc^-1: their slabs
  // carry it where such a solve is possible at all (<= 512 workgroups, <= 512 separator poses)
  pl.prod_post = pl.ns > 0 && pl.ns <= 512 && pl.nwg - pl.nS2 <= 512;
  // bytes of one apply: per workgroup (rows before the exchange [+ separator rows]) x 8 columns x 8 bytes
  double rows = 0;
  for (int b = 0; b < pl.nwg; ++b) rows += 4.0 * (tl_pre_rows(pl, b).size() + (b < pl.nA ? 0 : pl.ns));
  pl.bytes = rows * 8 * 8;
  return pl;
}

}  // namespace

// poses whose input-vector rows enter workgroup b's product BEFORE the exchange: for a separator workgroup the poses
// of every subdomain coupled to one of its poses (phase A), for an interior workgroup the poses of the subdomain(s)
// of its own poses
std::vector<int> tl_pre_rows(const TLPlan &pl, int b) {
  std::vector<int> subs;
  if (b >= pl.nA && b < pl.nA + pl.nS2) return {};  // separator poses, consumer side: u rows only
  for (int q = 0; q < 2; ++q) {
    const int v = pl.order[2 * b + q];
    if (v < 0) continue;
    if (pl.sub_of[v] >= 0) subs.push_back(pl.sub_of[v]);
    else for (int i : pl.adj_sub[pl.sep_index[v]]) subs.push_back(i);
  }
  std::sort(subs.begin(), subs.end());
  subs.erase(std::unique(subs.begin(), subs.end()), subs.end());
  std::vector<int> rows;
  for (int i : subs) rows.insert(rows.end(), pl.sub[i].begin(), pl.sub[i].end());
  return rows;
}

int tl_max_pre_poses(const TLPlan &pl) {
  // (no workgroup straddles two subdomains: an interior workgroup meets its subdomain, a separator workgroup the
  // subdomains adjacent to its two poses)
  size_t mx = 0;
  for (const auto &s : pl.sub) mx = std::max(mx, s.size());
  for (int b = 0; b < pl.nA; ++b) mx = std::max(mx, tl_pre_rows(pl, b).size());
  return (int)mx;
}

TLPlan tl_make_plan(int n, const std::vector<int> &rowptr, const std::vector<int> &col, int max_sub, int fit_pairs, int fit_wg) {
  const Graph g{n, rowptr, col};
  // fit_pairs / fit_wg > 0: among the ladder's plans prefer those whose largest slab (row pairs) and solve grid fit the
  // one-launch RTR solve (two workgroups per CU), fewest bytes among them; none fits: fewest bytes overall
  auto fits = [&](const TLPlan &pl) {
    return fit_pairs > 0 && pl.prod_post && 2 * (tl_max_pre_poses(pl) + pl.ns) <= fit_pairs && pl.nwg - pl.nS2 <= fit_wg;
  };
  if (max_sub > 0) {
    Dissection d = dissect(g, max_sub);
    thin(g, d, max_sub);
    return finish_plan(g, std::move(d));
  }
  static const int ladder[] = {2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024};
  TLPlan best;
  bool have = false;
  int last_ms = -1;
  for (int div : ladder) {
    const int ms = std::max(4, n / div);
    if (ms == last_ms) break;
    last_ms = ms;
    Dissection d = dissect(g, ms);
    thin(g, d, ms);
    TLPlan pl = finish_plan(g, std::move(d));
    const bool f_new = fits(pl), f_old = have && fits(best);
    if (!have || (f_new && !f_old) || (f_new == f_old && pl.bytes < best.bytes)) { best = std::move(pl); have = true; }
    if (ms <= 4) break;
  }
  return best;
}

}  // namespace dpgo_host
