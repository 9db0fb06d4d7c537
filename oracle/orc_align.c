/* orc_align.c -- TEST INFRASTRUCTURE ONLY (see dpgo_oracle.h).  CPU restatement of the robust inter-robot
 * frame alignment used when an agent first receives an initialised neighbour's public poses
 * (updateNeighborPoses -> initializeInGlobalFrame, src/PGOAgentROS.cpp:1276, 353-358; parameter
 * robustInitMinInliers, src/PGOAgentROSNode.cpp:150).  PARITY UNPINNED: dpgo's own routine is not in the tree;
 * this follows the published two-stage scheme -- GNC-TLS single-rotation averaging under the chordal metric,
 * then GNC-TLS translation averaging on the rotation inliers (Yang et al., RA-L 2020, Alg. 1 with the TLS
 * weight update (13) and mu initialisation of Remark 5). */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "dpgo_oracle.h"

static double weight_tls(double r2, double mu, double c2) {
  double hi = (mu + 1.0) / mu * c2, lo = mu / (mu + 1.0) * c2;
  if (r2 >= hi) return 0.0;
  if (r2 <= lo) return 1.0;
  return sqrt(c2 * mu * (mu + 1.0) / r2) - mu;
}

/* kind 0: rotation part (9 doubles at offset 0, projected to SO(3)); kind 1: translation (3 at offset 9) */
static int weighted_mean(const double *Tc, int n, const double *w, int kind, double *est) {
  int len = kind == 0 ? 9 : 3, off = kind == 0 ? 0 : 9;
  double acc[9] = {0}, sw = 0;
  for (int i = 0; i < n; ++i) {
    if (w[i] <= 0) continue;
    sw += w[i];
    for (int e = 0; e < len; ++e) acc[e] += w[i] * Tc[12 * (size_t)i + off + e];
  }
  if (!(sw > 0)) return 0;
  if (kind == 0) orc_project_rotation(acc, est);
  else for (int e = 0; e < 3; ++e) est[e] = acc[e] / sw;
  return 1;
}

static double sqdist(const double *Tc, int i, int kind, const double *est) {
  int len = kind == 0 ? 9 : 3, off = kind == 0 ? 0 : 9;
  double s = 0;
  for (int e = 0; e < len; ++e) { double d = est[e] - Tc[12 * (size_t)i + off + e]; s += d * d; }
  return s;
}

/* w[i] < 0 marks an excluded candidate.  Returns 0 when the weights vanish. */
static int gnc(const double *Tc, int n, int kind, double barc, double *w, double *est) {
  double c2 = barc * barc;
  if (!weighted_mean(Tc, n, w, kind, est)) return 0;
  double rmax2 = 0;
  for (int i = 0; i < n; ++i) if (w[i] > 0) { double r2 = sqdist(Tc, i, kind, est); if (r2 > rmax2) rmax2 = r2; }
  double mu = c2 / (2.0 * rmax2 - c2);
  if (!(mu > 0)) return 1;
  double *wn = (double *)malloc(sizeof(double) * n);
  int ok = 1;
  for (int it = 0; it < 1000; ++it) {
    int binary = 1;
    for (int i = 0; i < n; ++i) {
      if (w[i] < 0) { wn[i] = -1; continue; }
      wn[i] = weight_tls(sqdist(Tc, i, kind, est), mu, c2);
      if (wn[i] > 0 && wn[i] < 1) binary = 0;
    }
    if (!weighted_mean(Tc, n, wn, kind, est)) { ok = 0; break; }
    memcpy(w, wn, sizeof(double) * n);
    if (binary) break;
    mu *= 1.4;
  }
  free(wn);
  return ok;
}

int orc_robust_frame_alignment(const double *Tc, int n, double max_rot_rad, double max_trans, int min_inliers,
                               double *T_out, int *inlier) {
  if (n <= 0) return 1;
  double *w = (double *)malloc(sizeof(double) * n);
  double R[9], t[3];
  int rc = 1, cnt = 0;
  for (int i = 0; i < n; ++i) w[i] = 1.0;
  if (!gnc(Tc, n, 0, 2.0 * sqrt(2.0) * sin(0.5 * max_rot_rad), w, R)) goto done;
  for (int i = 0; i < n; ++i) { if (w[i] > 0.5) { w[i] = 1.0; ++cnt; } else w[i] = -1.0; }
  if (cnt < min_inliers) goto done;
  if (!gnc(Tc, n, 1, max_trans, w, t)) goto done;
  cnt = 0;
  for (int i = 0; i < n; ++i) {
    int in = w[i] > 0.5;
    if (inlier) inlier[i] = in;
    cnt += in;
    w[i] = in ? 1.0 : -1.0;
  }
  if (cnt < min_inliers) goto done;
  weighted_mean(Tc, n, w, 1, t);
  weighted_mean(Tc, n, w, 0, R);
  memcpy(T_out, R, sizeof R);
  memcpy(T_out + 9, t, sizeof t);
  rc = 0;
done:
  free(w);
  return rc;
}
