#pragma once
#include <geometry_msgs/Pose.h>
namespace tf {
typedef double tfScalar;
struct Vector3 { double v[3] = {0, 0, 0}; Vector3() {} Vector3(double a, double b, double c) { v[0] = a; v[1] = b; v[2] = c; }
  double x() const { return v[0]; } double y() const { return v[1]; } double z() const { return v[2]; }
  double getX() const { return v[0]; } double getY() const { return v[1]; } double getZ() const { return v[2]; }
  double &operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } };
typedef Vector3 Point;
struct Quaternion { double q[4] = {0, 0, 0, 1}; Quaternion() {} Quaternion(double a, double b, double c, double d) { q[0] = a; q[1] = b; q[2] = c; q[3] = d; }
  double x() const { return q[0]; } double y() const { return q[1]; } double z() const { return q[2]; } double w() const { return q[3]; }
  Quaternion &normalize() { return *this; } Quaternion normalized() const { return *this; } };
struct Matrix3x3 {
  double m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  Matrix3x3() {}
  explicit Matrix3x3(const Quaternion &) {}
  Matrix3x3(double, double, double, double, double, double, double, double, double) {}
  void getRotation(Quaternion &) const {}
  void setRotation(const Quaternion &) {}
  void setValue(double, double, double, double, double, double, double, double, double) {}
  const Vector3 getRow(int) const { return Vector3(); }
  const Vector3 getColumn(int) const { return Vector3(); }
  struct RowProxy { double r[3]; double &operator[](int j) { return r[j]; } double operator[](int j) const { return r[j]; } };
  Vector3 operator[](int) const { return Vector3(); }
};
inline void quaternionTFToMsg(const Quaternion &, geometry_msgs::Quaternion &) {}
inline void quaternionMsgToTF(const geometry_msgs::Quaternion &, Quaternion &) {}
inline void pointTFToMsg(const Point &, geometry_msgs::Point &) {}
inline void pointMsgToTF(const geometry_msgs::Point &, Point &) {}
}
