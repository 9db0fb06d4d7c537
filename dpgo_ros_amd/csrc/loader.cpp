// loader.cpp -- host-side dataset input for the drop-in boundary (plain C++17, no Eigen).
//
// Replaces, for the hot path's inputs:
//   read_g2o_file(filename, num_poses)           src/PGODatasetPublisherNode.cpp:80
//   PGOLogger::loadMeasurements(file, false)     src/PGODatasetPublisherNode.cpp:168
//   contiguous-block partition + classification  src/PGODatasetPublisherNode.cpp:84-135
//   wrapper weighting kappa=1e4 / tau=1e2, odometry => fixedWeight   src/utils.cpp:141-149
// and writes the same formats back (PGOLogger::logMeasurements / logTrajectory analogue, so that
// GNC weights and rounded trajectories round-trip through loadMeasurements / read_g2o_file).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/dpgo_hip.h"

namespace {

void quat_to_rot(double qx, double qy, double qz, double qw, double R[9]) {
  const double s = std::sqrt(qx * qx + qy * qy + qz * qz + qw * qw);
  const double x = qx / s, y = qy / s, z = qz / s, w = qw / s;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

// unit quaternion (x, y, z, w), w >= 0, of a row-major rotation (Shepperd's branch on the largest pivot)
void rot_to_quat(const double R[9], double q[4]) {
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    const double s = 2.0 * std::sqrt(1.0 + tr);
    q[3] = 0.25 * s; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double s = 2.0 * std::sqrt(1.0 + R[0] - R[4] - R[8]);
    q[3] = (R[7] - R[5]) / s; q[0] = 0.25 * s; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const double s = 2.0 * std::sqrt(1.0 + R[4] - R[0] - R[8]);
    q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = 0.25 * s; q[2] = (R[5] + R[7]) / s;
  } else {
    const double s = 2.0 * std::sqrt(1.0 + R[8] - R[0] - R[4]);
    q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = 0.25 * s;
  }
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double sg = q[3] < 0 ? -1.0 : 1.0;
  for (int k = 0; k < 4; ++k) q[k] = sg * q[k] / n;
}

// tr(S^-1) of the symmetric 3x3 [a b c; b d e; c e f]
double trace_inverse_sym3(double a, double b, double c, double d, double e, double f) {
  const double m00 = d * f - e * e, m11 = a * f - c * c, m22 = a * d - b * b;
  const double det = a * m00 - b * (b * f - c * e) + c * (b * e - c * d);
  return (m00 + m11 + m22) / det;
}

dpgo_measurement_t *to_c_array(const std::vector<dpgo_measurement_t> &v) {
  auto *out = static_cast<dpgo_measurement_t *>(std::malloc(sizeof(dpgo_measurement_t) * (v.empty() ? 1 : v.size())));
  if (!v.empty()) std::memcpy(out, v.data(), sizeof(dpgo_measurement_t) * v.size());
  return out;
}

}  // namespace

extern "C" {

void dpgo_free(void *p) { std::free(p); }

int dpgo_read_g2o(const char *path, int weight_mode, dpgo_measurement_t **out, int *num_poses) {
  std::ifstream in(path);
  if (!in) return -1;
  std::vector<dpgo_measurement_t> ms;
  std::string line, tag;
  int max_id = -1;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    if (!(ss >> tag) || tag != "EDGE_SE3:QUAT") continue;
    int i, j;
    double v[7], I[21];
    if (!(ss >> i >> j)) continue;
    bool ok = true;
    for (double &x : v) ok = ok && static_cast<bool>(ss >> x);
    for (double &x : I) ok = ok && static_cast<bool>(ss >> x);
    if (!ok) continue;
    dpgo_measurement_t m{};
    m.r1 = 0; m.r2 = 0; m.p1 = i; m.p2 = j;
    m.t[0] = v[0]; m.t[1] = v[1]; m.t[2] = v[2];
    quat_to_rot(v[3], v[4], v[5], v[6], m.R);
    if (weight_mode == DPGO_WEIGHT_WRAPPER) {
      m.kappa = 10000.0; m.tau = 100.0;
    } else {
      // upper-triangular 6x6, translation block first: rows start at I[0], I[6], I[11], I[15], I[18], I[20]
      m.tau = 3.0 / trace_inverse_sym3(I[0], I[1], I[2], I[6], I[7], I[11]);
      m.kappa = 3.0 / (2.0 * trace_inverse_sym3(I[15], I[16], I[17], I[18], I[19], I[20]));
    }
    m.weight = 1.0;
    ms.push_back(m);
    max_id = std::max(max_id, std::max(i, j));
  }
  *out = to_c_array(ms);
  *num_poses = max_id + 1;
  return static_cast<int>(ms.size());
}

int dpgo_read_measurements_csv(const char *path, int weight_mode, dpgo_measurement_t **out) {
  std::ifstream in(path);
  if (!in) return -1;
  std::vector<dpgo_measurement_t> ms;
  std::string line;
  std::getline(in, line);  // header
  while (std::getline(in, line)) {
    for (char &ch : line) if (ch == ',') ch = ' ';
    std::istringstream ss(line);
    double v[15];
    bool ok = true;
    for (double &x : v) ok = ok && static_cast<bool>(ss >> x);
    if (!ok) continue;
    dpgo_measurement_t m{};
    m.r1 = (int)v[0]; m.p1 = (int)v[1]; m.r2 = (int)v[2]; m.p2 = (int)v[3];
    quat_to_rot(v[4], v[5], v[6], v[7], m.R);
    m.t[0] = v[8]; m.t[1] = v[9]; m.t[2] = v[10];
    if (weight_mode == DPGO_WEIGHT_WRAPPER) {
      m.kappa = 10000.0; m.tau = 100.0; m.weight = 1.0;
      m.fixed_weight = (m.r1 == m.r2 && m.p1 + 1 == m.p2);
    } else {
      m.kappa = v[11]; m.tau = v[12];
      m.is_known_inlier = (int)v[13];
      m.weight = v[14];
      m.fixed_weight = m.is_known_inlier;
    }
    ms.push_back(m);
  }
  *out = to_c_array(ms);
  return static_cast<int>(ms.size());
}

int dpgo_write_measurements_csv(const char *path, const dpgo_measurement_t *m, int nm) {
  FILE *f = std::fopen(path, "w");
  if (!f) return -1;
  std::fprintf(f, "robot_src,pose_src,robot_dst,pose_dst,qx,qy,qz,qw,tx,ty,tz,kappa,tau,is_known_inlier,weight\n");
  for (int k = 0; k < nm; ++k) {
    double q[4];
    rot_to_quat(m[k].R, q);
    std::fprintf(f, "%d,%d,%d,%d,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%d,%.17g\n", m[k].r1, m[k].p1,
                 m[k].r2, m[k].p2, q[0], q[1], q[2], q[3], m[k].t[0], m[k].t[1], m[k].t[2], m[k].kappa, m[k].tau,
                 m[k].is_known_inlier ? 1 : 0, m[k].weight);
  }
  return std::fclose(f) == 0 ? nm : -1;
}

int dpgo_write_g2o(const char *path, const dpgo_measurement_t *m, int nm, const double *T, int num_poses,
                   const int *robot_offsets) {
  FILE *f = std::fopen(path, "w");
  if (!f) return -1;
  for (int i = 0; T && i < num_poses; ++i) {
    const double *Ti = T + (size_t)12 * i;  // column-major 3x4
    const double Rm[9] = {Ti[0], Ti[3], Ti[6], Ti[1], Ti[4], Ti[7], Ti[2], Ti[5], Ti[8]};
    double q[4];
    rot_to_quat(Rm, q);
    std::fprintf(f, "VERTEX_SE3:QUAT %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", i, Ti[9], Ti[10], Ti[11], q[0], q[1],
                 q[2], q[3]);
  }
  for (int k = 0; k < nm; ++k) {
    const int i = m[k].p1 + (robot_offsets ? robot_offsets[m[k].r1] : 0);
    const int j = m[k].p2 + (robot_offsets ? robot_offsets[m[k].r2] : 0);
    double q[4];
    rot_to_quat(m[k].R, q);
    std::fprintf(f, "EDGE_SE3:QUAT %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g", i, j, m[k].t[0], m[k].t[1], m[k].t[2],
                 q[0], q[1], q[2], q[3]);
    // isotropic information: translation block tau I, rotation block 2 kappa I (inverse of the reader's
    // tau = 3 / tr(I_t^-1), kappa = 3 / (2 tr(I_R^-1)))
    const double d[6] = {m[k].tau, m[k].tau, m[k].tau, 2 * m[k].kappa, 2 * m[k].kappa, 2 * m[k].kappa};
    for (int a = 0; a < 6; ++a)
      for (int b = a; b < 6; ++b) std::fprintf(f, " %.17g", a == b ? d[a] : 0.0);
    std::fprintf(f, "\n");
  }
  return std::fclose(f) == 0 ? nm : -1;
}

int dpgo_write_trajectory_csv(const char *path, const double *T, int num_poses) {
  FILE *f = std::fopen(path, "w");
  if (!f) return -1;
  std::fprintf(f, "pose_index,qx,qy,qz,qw,tx,ty,tz\n");
  for (int i = 0; i < num_poses; ++i) {
    const double *Ti = T + (size_t)12 * i;
    const double Rm[9] = {Ti[0], Ti[3], Ti[6], Ti[1], Ti[4], Ti[7], Ti[2], Ti[5], Ti[8]};
    double q[4];
    rot_to_quat(Rm, q);
    std::fprintf(f, "%d,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n", i, q[0], q[1], q[2], q[3], Ti[9], Ti[10], Ti[11]);
  }
  return std::fclose(f) == 0 ? num_poses : -1;
}

void dpgo_partition(dpgo_measurement_t *m, int nm, int num_poses, int num_robots, int weight_mode) {
  if (num_robots <= 0 || nm <= 0 || !m) return;
  const int per = num_poses / num_robots;
  for (int k = 0; k < nm; ++k) {
    const int g1 = m[k].p1, g2 = m[k].p2;
    // more robots than poses (per == 0): the reference logs an error and its index map then gives every pose to the
    // last robot (src/PGODatasetPublisherNode.cpp:85-103: all ranges but the last are empty); same here, no division
    const int ra = per > 0 ? std::min(g1 / per, num_robots - 1) : num_robots - 1;
    const int rb = per > 0 ? std::min(g2 / per, num_robots - 1) : num_robots - 1;
    m[k].r1 = ra; m[k].p1 = g1 - ra * per;
    m[k].r2 = rb; m[k].p2 = g2 - rb * per;
    if (weight_mode == DPGO_WEIGHT_WRAPPER) m[k].fixed_weight = (ra == rb && m[k].p1 + 1 == m[k].p2);
  }
}

void dpgo_odometry_init(const dpgo_measurement_t *m, int nm, int num_poses, double *T) {
  std::vector<const dpgo_measurement_t *> odo(num_poses, nullptr);
  for (int e = 0; e < nm; ++e)
    if (m[e].r1 == m[e].r2 && m[e].p2 == m[e].p1 + 1 && m[e].p1 >= 0 && m[e].p2 < num_poses && !odo[m[e].p1]) odo[m[e].p1] = &m[e];
  std::memset(T, 0, sizeof(double) * 12 * (size_t)num_poses);
  T[0] = T[4] = T[8] = 1.0;
  for (int i = 0; i + 1 < num_poses; ++i) {
    const double *Ti = T + (size_t)12 * i;
    double *Tn = T + (size_t)12 * (i + 1);
    if (!odo[i]) { std::memcpy(Tn, Ti, sizeof(double) * 12); continue; }
    const dpgo_measurement_t &e = *odo[i];
    for (int c = 0; c < 3; ++c)
      for (int a = 0; a < 3; ++a) {
        double s = 0;
        for (int b = 0; b < 3; ++b) s += Ti[3 * b + a] * e.R[3 * b + c];
        Tn[3 * c + a] = s;
      }
    for (int a = 0; a < 3; ++a) {
      double s = Ti[9 + a];
      for (int b = 0; b < 3; ++b) s += Ti[3 * b + a] * e.t[b];
      Tn[9 + a] = s;
    }
  }
}

void dpgo_fixed_stiefel(int r, double *YLift) {
  std::memset(YLift, 0, sizeof(double) * 3 * r);
  for (int c = 0; c < 3; ++c) YLift[c * r + c] = 1.0;
}

void dpgo_lift(const double *T, int num_poses, const double *YLift, int r, double *X) {
  for (int i = 0; i < num_poses; ++i)
    for (int c = 0; c < 4; ++c)
      for (int a = 0; a < r; ++a) {
        double s = 0;
        for (int b = 0; b < 3; ++b) s += YLift[b * r + a] * T[(size_t)12 * i + 3 * c + b];
        X[((size_t)4 * i + c) * r + a] = s;
      }
}

}  // extern "C"
