/* orc_io.c -- dataset input of the CPU oracle (TEST INFRASTRUCTURE ONLY; see dpgo_oracle.h).
 *
 * Restates, from the reference tree:
 *   - g2o EDGE_SE3:QUAT layout (data/sphere2500.g2o:2501): i j x y z qx qy qz qw + 21
 *     upper-triangular information entries, translation block first.
 *   - contiguous-block partition + edge classification (src/PGODatasetPublisherNode.cpp:84-135).
 *   - wrapper weighting kappa=1e4, tau=1e2 and odometry => fixedWeight (src/utils.cpp:141-149).
 *   - tunnels CSV header (data/tunnels/robot0/measurements.csv:1).
 * Library weighting (kappa, tau from the information matrix) is the SE-Sync convention used by
 * dpgo's own read_g2o_file [UPSTREAM-RECALL]: tau = 3/tr(I_t^-1), kappa = 3/(2 tr(I_R^-1)).
 */
#include "orc_internal.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void orc_free(void *p) { free(p); }

void orc_quat_to_rot(double qx, double qy, double qz, double qw, double R[9]) {
  double nrm = sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
  qx /= nrm; qy /= nrm; qz /= nrm; qw /= nrm;
  R[0] = 1 - 2 * (qy * qy + qz * qz); R[1] = 2 * (qx * qy - qz * qw);     R[2] = 2 * (qx * qz + qy * qw);
  R[3] = 2 * (qx * qy + qz * qw);     R[4] = 1 - 2 * (qx * qx + qz * qz); R[5] = 2 * (qy * qz - qx * qw);
  R[6] = 2 * (qx * qz - qy * qw);     R[7] = 2 * (qy * qz + qx * qw);     R[8] = 1 - 2 * (qx * qx + qy * qy);
}

/* trace of the inverse of a symmetric 3x3 given as a,b,c / d,e / f (upper triangle) */
static double inv_trace_sym3(double a, double b, double c, double d, double e, double f) {
  double c00 = d * f - e * e, c11 = a * f - c * c, c22 = a * d - b * b;
  double det = a * c00 - b * (b * f - c * e) + c * (b * e - c * d);
  return (c00 + c11 + c22) / det;
}

int orc_read_g2o(const char *path, int weight_mode, orc_meas_t **out, int *num_poses) {
  FILE *fp = fopen(path, "r");
  if (!fp) return -1;
  int cap = 1024, nm = 0, maxid = -1;
  orc_meas_t *m = (orc_meas_t *)malloc(sizeof(orc_meas_t) * cap);
  char line[4096];
  while (fgets(line, sizeof line, fp)) {
    if (strncmp(line, "EDGE_SE3:QUAT", 13) != 0) continue;
    int i, j, off = 0;
    double v[7], I[21];
    if (sscanf(line + 13, "%d %d%n", &i, &j, &off) != 2) continue;
    char *p = line + 13 + off;
    int ok = 1;
    for (int q = 0; q < 7 && ok; ++q) { char *e; v[q] = strtod(p, &e); if (e == p) ok = 0; p = e; }
    for (int q = 0; q < 21 && ok; ++q) { char *e; I[q] = strtod(p, &e); if (e == p) ok = 0; p = e; }
    if (!ok) continue;
    if (nm == cap) { cap *= 2; m = (orc_meas_t *)realloc(m, sizeof(orc_meas_t) * cap); }
    orc_meas_t *e = &m[nm++];
    memset(e, 0, sizeof *e);
    e->r1 = 0; e->r2 = 0; e->p1 = i; e->p2 = j;
    e->t[0] = v[0]; e->t[1] = v[1]; e->t[2] = v[2];
    orc_quat_to_rot(v[3], v[4], v[5], v[6], e->R);
    if (weight_mode == ORC_WEIGHT_WRAPPER) {
      e->kappa = 10000.0; e->tau = 100.0;  /* src/utils.cpp:141-142 */
    } else {
      /* upper-tri 6x6 row-major: row0: I[0..5], row1: I[6..10], row2: I[11..14], row3: I[15..17],
       * row4: I[18..19], row5: I[20] */
      e->tau = 3.0 / inv_trace_sym3(I[0], I[1], I[2], I[6], I[7], I[11]);
      e->kappa = 3.0 / (2.0 * inv_trace_sym3(I[15], I[16], I[17], I[18], I[19], I[20]));
    }
    e->weight = 1.0; e->fixed_weight = 0; e->is_known_inlier = 0;
    if (i > maxid) maxid = i;
    if (j > maxid) maxid = j;
  }
  fclose(fp);
  *out = m;
  *num_poses = maxid + 1;
  return nm;
}

int orc_read_measurements_csv(const char *path, int weight_mode, orc_meas_t **out) {
  FILE *fp = fopen(path, "r");
  if (!fp) return -1;
  int cap = 1024, nm = 0;
  orc_meas_t *m = (orc_meas_t *)malloc(sizeof(orc_meas_t) * cap);
  char line[4096];
  if (!fgets(line, sizeof line, fp)) { fclose(fp); *out = m; return 0; } /* header */
  while (fgets(line, sizeof line, fp)) {
    double v[15];
    char *p = line;
    int ok = 1;
    for (int q = 0; q < 15 && ok; ++q) {
      char *e; v[q] = strtod(p, &e);
      if (e == p) { ok = 0; break; }
      p = e; while (*p == ',' || *p == ' ') ++p;
    }
    if (!ok) continue;
    if (nm == cap) { cap *= 2; m = (orc_meas_t *)realloc(m, sizeof(orc_meas_t) * cap); }
    orc_meas_t *e = &m[nm++];
    memset(e, 0, sizeof *e);
    e->r1 = (int)v[0]; e->p1 = (int)v[1]; e->r2 = (int)v[2]; e->p2 = (int)v[3];
    orc_quat_to_rot(v[4], v[5], v[6], v[7], e->R);
    e->t[0] = v[8]; e->t[1] = v[9]; e->t[2] = v[10];
    if (weight_mode == ORC_WEIGHT_WRAPPER) {
      /* msg codec drops kappa/tau/weight/inlier and re-derives them (src/utils.cpp:108-152) */
      e->kappa = 10000.0; e->tau = 100.0; e->weight = 1.0;
      e->fixed_weight = (e->r1 == e->r2 && e->p1 + 1 == e->p2);
      e->is_known_inlier = 0;
    } else {
      e->kappa = v[11]; e->tau = v[12];
      e->is_known_inlier = (int)v[13]; e->weight = v[14];
      e->fixed_weight = e->is_known_inlier;
    }
  }
  fclose(fp);
  *out = m;
  return nm;
}

void orc_partition(orc_meas_t *m, int nm, int num_poses, int num_robots, int weight_mode) {
  int per = num_poses / num_robots; /* PGODatasetPublisherNode.cpp:85 */
  for (int k = 0; k < nm; ++k) {
    int g1 = m[k].p1, g2 = m[k].p2;
    /* per == 0 (more robots than poses): every range but the last is empty (:85-103), all poses go to the last robot */
    int ra = per > 0 ? g1 / per : num_robots - 1; if (ra >= num_robots) ra = num_robots - 1; /* last robot takes remainder :95 */
    int rb = per > 0 ? g2 / per : num_robots - 1; if (rb >= num_robots) rb = num_robots - 1;
    m[k].r1 = ra; m[k].p1 = g1 - ra * per;
    m[k].r2 = rb; m[k].p2 = g2 - rb * per;
    if (weight_mode == ORC_WEIGHT_WRAPPER)
      m[k].fixed_weight = (m[k].r1 == m[k].r2 && m[k].p1 + 1 == m[k].p2); /* utils.cpp:147-149 */
  }
}
