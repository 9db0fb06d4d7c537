// DPGO_types.h -- types of `namespace DPGO` consumed by the ROS wrapper (SURVEY App. A "Types").
// Header-only facade over the C-ABI (include/dpgo_hip.h).  `DPGO::Matrix` is Eigen::MatrixXd when
// Eigen is installed (the ROS box: include/dpgo_ros/utils.h:35-53 uses it as such); without Eigen a
// minimal column-major matrix with the same storage layout stands in so that the facade and its mock
// wrapper test still build.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <optional>
#include <set>
#include <string>
#include <vector>

// The wrapper's translation units rely on the library's headers for glog's CHECK macros (src/utils.cpp:284 uses CHECK
// and includes nothing else that declares it), for <fstream> (include/dpgo_ros/PGOAgentROS.h:153 std::ofstream) and for
// unqualified `vector` (src/PGOAgentROS.cpp:288).  tests/test_wrapper_sources.py type-checks those sources against
// these headers.
#if __has_include(<glog/logging.h>)
#include <glog/logging.h>
#else
namespace DPGO { namespace detail {
struct CheckSink {
  bool fatal;
  explicit CheckSink(bool f) : fatal(f) {}
  ~CheckSink() { if (fatal) { std::cerr << std::endl; std::abort(); } }
  template <class T> CheckSink &operator<<(const T &v) { if (fatal) std::cerr << v; return *this; }
  CheckSink &operator<<(std::ostream &(*f)(std::ostream &)) { if (fatal) std::cerr << f; return *this; }
};
} }
#define CHECK(cond) if (cond) {} else ::DPGO::detail::CheckSink(true) << "CHECK failed: " #cond " "
#define CHECK_EQ(a, b) CHECK((a) == (b))
#define CHECK_NE(a, b) CHECK((a) != (b))
#define CHECK_LT(a, b) CHECK((a) < (b))
#define CHECK_LE(a, b) CHECK((a) <= (b))
#define CHECK_GT(a, b) CHECK((a) > (b))
#define CHECK_GE(a, b) CHECK((a) >= (b))
#define LOG(severity) ::DPGO::detail::CheckSink(false)
#define LOG_IF(severity, cond) ::DPGO::detail::CheckSink(false)
#endif

namespace DPGO {
using std::map;
using std::set;
using std::string;
using std::vector;
}

#if __has_include(<Eigen/Dense>) && !defined(DPGO_FACADE_NO_EIGEN)
#include <Eigen/Dense>
#define DPGO_FACADE_HAS_EIGEN 1
namespace DPGO {
typedef Eigen::MatrixXd Matrix;
typedef Eigen::VectorXd Vector;
typedef Eigen::Block<Matrix> MatrixBlockRef;
inline MatrixBlockRef mat_block_ref(Matrix &M, size_t i, size_t j, size_t p, size_t q) { return M.block(i, j, p, q); }
}  // namespace DPGO
#else
#define DPGO_FACADE_HAS_EIGEN 0
namespace DPGO {
class Matrix {
 public:
  Matrix() : r_(0), c_(0) {}
  Matrix(size_t r, size_t c) : r_(r), c_(c), d_(r * c, 0.0) {}
  static Matrix Zero(size_t r, size_t c) { return Matrix(r, c); }
  static Matrix Zero(size_t n) { return Matrix(n, 1); }  // Vector::Zero(n) (src/PGOAgentROS.cpp:1465)
  // Eigen's comma initialiser, row by row (src/utils.cpp:69-71,80:  R << a, b, c, ...;)
  struct CommaInit {
    Matrix *m; size_t k;
    CommaInit &operator,(double v) { m->d_[(k % m->c_) * m->r_ + k / m->c_] = v; ++k; return *this; }
  };
  CommaInit operator<<(double v) { CommaInit ci{this, 0}; ci, v; return ci; }
  static Matrix Identity(size_t r, size_t c) { Matrix m(r, c); for (size_t i = 0; i < r && i < c; ++i) m(i, i) = 1.0; return m; }
  size_t rows() const { return r_; }
  size_t cols() const { return c_; }
  size_t size() const { return d_.size(); }
  double &operator()(size_t i, size_t j) { return d_[j * r_ + i]; }
  double operator()(size_t i, size_t j) const { return d_[j * r_ + i]; }
  double &operator()(size_t i) { return d_[i]; }
  double operator()(size_t i) const { return d_[i]; }
  double *data() { return d_.data(); }
  const double *data() const { return d_.data(); }
  Matrix block(size_t i, size_t j, size_t p, size_t q) const {
    Matrix b(p, q);
    for (size_t y = 0; y < q; ++y) for (size_t x = 0; x < p; ++x) b(x, y) = (*this)(i + x, j + y);
    return b;
  }
  void setBlock(size_t i, size_t j, const Matrix &b) {
    for (size_t y = 0; y < b.cols(); ++y) for (size_t x = 0; x < b.rows(); ++x) (*this)(i + x, j + y) = b(x, y);
  }
  Matrix transpose() const { Matrix t(c_, r_); for (size_t j = 0; j < c_; ++j) for (size_t i = 0; i < r_; ++i) t(j, i) = (*this)(i, j); return t; }
  double norm() const { double s = 0; for (double v : d_) s += v * v; return std::sqrt(s); }
  Matrix operator*(const Matrix &o) const {
    Matrix m(r_, o.c_);
    for (size_t j = 0; j < o.c_; ++j) for (size_t k = 0; k < c_; ++k) for (size_t i = 0; i < r_; ++i) m(i, j) += (*this)(i, k) * o(k, j);
    return m;
  }
  Matrix operator-(const Matrix &o) const { Matrix m(*this); for (size_t i = 0; i < d_.size(); ++i) m.d_[i] -= o.d_[i]; return m; }
  Matrix operator+(const Matrix &o) const { Matrix m(*this); for (size_t i = 0; i < d_.size(); ++i) m.d_[i] += o.d_[i]; return m; }
 private:
  size_t r_, c_;
  std::vector<double> d_;
};
typedef Matrix Vector;
// assignable view of a block, what Eigen's M.block(i, j, p, q) is on the ROS box: the wrapper writes
// `X.rotation() = YLift.value()` (src/PGOAgentROS.cpp:1464) and `poses.rotation(i) = R` (:299-300)
class MatrixBlock {
 public:
  MatrixBlock(Matrix &m, size_t i, size_t j, size_t p, size_t q) : m_(m), i_(i), j_(j), p_(p), q_(q) {}
  MatrixBlock &operator=(const Matrix &b) { m_.setBlock(i_, j_, b); return *this; }
  MatrixBlock &operator=(const MatrixBlock &b) { m_.setBlock(i_, j_, (Matrix)b); return *this; }
  operator Matrix() const { return m_.block(i_, j_, p_, q_); }
  size_t rows() const { return p_; }
  size_t cols() const { return q_; }
  double &operator()(size_t x, size_t y) { return m_(i_ + x, j_ + y); }
  double operator()(size_t x, size_t y) const { return m_(i_ + x, j_ + y); }
  Matrix transpose() const { return ((Matrix)*this).transpose(); }
  double norm() const { return ((Matrix)*this).norm(); }
  Matrix operator*(const Matrix &o) const { return (Matrix)*this * o; }
 private:
  Matrix &m_;
  size_t i_, j_, p_, q_;
};
typedef MatrixBlock MatrixBlockRef;
inline MatrixBlockRef mat_block_ref(Matrix &M, size_t i, size_t j, size_t p, size_t q) { return MatrixBlock(M, i, j, p, q); }
}  // namespace DPGO
#endif

namespace DPGO {

// neutral element access helpers (work for both Matrix flavours)
inline Matrix mat_block(const Matrix &M, size_t i, size_t j, size_t p, size_t q) {
  Matrix b = Matrix::Zero(p, q);
  for (size_t y = 0; y < q; ++y) for (size_t x = 0; x < p; ++x) b(x, y) = M(i + x, j + y);
  return b;
}
inline void mat_set_block(Matrix &M, size_t i, size_t j, const Matrix &b) {
  for (size_t y = 0; y < (size_t)b.cols(); ++y) for (size_t x = 0; x < (size_t)b.rows(); ++x) M(i + x, j + y) = b(x, y);
}

// PoseID(robot, frame)  (src/PGOAgentROS.cpp:271,685,1271,1396)
struct PoseID {
  unsigned int robot_id, frame_id;
  PoseID(unsigned int rid = 0, unsigned int fid = 0) : robot_id(rid), frame_id(fid) {}
  bool operator==(const PoseID &o) const { return robot_id == o.robot_id && frame_id == o.frame_id; }
};
struct ComparePoseID {  // include/dpgo_ros/PGOAgentROS.h:189
  bool operator()(const PoseID &a, const PoseID &b) const {
    return a.robot_id < b.robot_id || (a.robot_id == b.robot_id && a.frame_id < b.frame_id);
  }
};
// EdgeID(src, dst) + isSharedLoopClosure + HashEdgeID  (src/PGOAgentROS.cpp:1434-1435; PGOAgentROS.h:192)
struct EdgeID {
  PoseID src_pose_id, dst_pose_id;
  EdgeID(const PoseID &s, const PoseID &d) : src_pose_id(s), dst_pose_id(d) {}
  bool isOdometry() const { return src_pose_id.robot_id == dst_pose_id.robot_id && src_pose_id.frame_id + 1 == dst_pose_id.frame_id; }
  bool isPrivateLoopClosure() const { return src_pose_id.robot_id == dst_pose_id.robot_id && !isOdometry(); }
  bool isSharedLoopClosure() const { return src_pose_id.robot_id != dst_pose_id.robot_id; }
  bool operator==(const EdgeID &o) const { return src_pose_id == o.src_pose_id && dst_pose_id == o.dst_pose_id; }
};
struct HashEdgeID {
  size_t operator()(const EdgeID &e) const {
    size_t h = 1469598103934665603ull;
    for (unsigned v : {e.src_pose_id.robot_id, e.src_pose_id.frame_id, e.dst_pose_id.robot_id, e.dst_pose_id.frame_id}) {
      h ^= v; h *= 1099511628211ull;
    }
    return h;
  }
};

// LiftedPose(r, d): r x (d+1), .rotation() .translation() .setData() .getData()  (:1420-1422,1463-1466)
class LiftedPose {
 public:
  LiftedPose() : r_(0), d_(0) {}
  LiftedPose(unsigned r, unsigned d) : r_(r), d_(d), X_(Matrix::Zero(r, d + 1)) { for (unsigned i = 0; i < d; ++i) X_(i, i) = 1.0; }
  explicit LiftedPose(const Matrix &X) : r_(X.rows()), d_(X.cols() - 1), X_(X) {}
  unsigned r() const { return r_; }
  unsigned d() const { return d_; }
  const Matrix &pose() const { return X_; }
  const Matrix &getData() const { return X_; }
  void setData(const Matrix &X) { X_ = X; }
  Matrix rotation() const { return mat_block(X_, 0, 0, r_, d_); }
  Matrix translation() const { return mat_block(X_, 0, d_, r_, 1); }
  // assignable views (src/PGOAgentROS.cpp:1464-1465: X.rotation() = ..., X.translation() = ...)
  MatrixBlockRef rotation() { return mat_block_ref(X_, 0, 0, r_, d_); }
  MatrixBlockRef translation() { return mat_block_ref(X_, 0, d_, r_, 1); }
  void setRotation(const Matrix &Y) { mat_set_block(X_, 0, 0, Y); }
  void setTranslation(const Matrix &p) { mat_set_block(X_, 0, d_, p); }
 protected:
  unsigned r_, d_;
  Matrix X_;
};
// Pose(d), Pose(Matrix)  (:353,357,1398-1399)
class Pose : public LiftedPose {
 public:
  Pose() : LiftedPose() {}  // containers / std::optional of Pose in the wrapper need it
  explicit Pose(unsigned d) : LiftedPose(d, d) {}
  explicit Pose(const Matrix &T) : LiftedPose(T) {}
  Pose inverse() const {
    Matrix Rt = rotation().transpose();
    Matrix t = Rt * translation();
    Pose out(d_);
    out.setRotation(Rt);
    for (unsigned i = 0; i < d_; ++i) out.X_(i, d_) = -t(i, 0);
    return out;
  }
  Pose operator*(const Pose &o) const {
    Pose out(d_);
    out.setRotation(rotation() * o.rotation());
    Matrix t = rotation() * o.translation() + translation();
    out.setTranslation(t);
    return out;
  }
};
// PoseArray(d, n): .rotation(i) .translation(i) .pose(i) .d() .n() .getData()  (:285,299-300,357,623,632)
class PoseArray {
 public:
  PoseArray(unsigned d, unsigned n) : d_(d), n_(n), X_(Matrix::Zero(d, (d + 1) * n)) {}
  unsigned d() const { return d_; }
  unsigned n() const { return n_; }
  const Matrix &getData() const { return X_; }
  void setData(const Matrix &X) { X_ = X; }
  Matrix pose(unsigned i) const { return mat_block(X_, 0, i * (d_ + 1), d_, d_ + 1); }
  Matrix rotation(unsigned i) const { return mat_block(X_, 0, i * (d_ + 1), d_, d_); }
  Matrix translation(unsigned i) const { return mat_block(X_, 0, i * (d_ + 1) + d_, d_, 1); }
  // assignable views (src/PGOAgentROS.cpp:299-300: poses.rotation(i) = R; poses.translation(i) = t)
  MatrixBlockRef rotation(unsigned i) { return mat_block_ref(X_, 0, i * (d_ + 1), d_, d_); }
  MatrixBlockRef translation(unsigned i) { return mat_block_ref(X_, 0, i * (d_ + 1) + d_, d_, 1); }
  void setPose(unsigned i, const Matrix &T) { mat_set_block(X_, 0, i * (d_ + 1), T); }
 private:
  unsigned d_, n_;
  Matrix X_;
};
typedef std::map<PoseID, LiftedPose, ComparePoseID> PoseDict;

// msg/Status.msg:1-3, tests/testUtils.cpp:67-69
// unscoped: the wrapper assigns it to a uint8 message field without a cast (src/utils.cpp:265) and also spells the
// enumerators PGOAgentState::X
enum PGOAgentState { WAIT_FOR_DATA = 0, WAIT_FOR_INITIALIZATION = 1, INITIALIZED = 2 };
// PGOAgentStatus(agentID, state, instanceNumber, iterationNumber, readyToTerminate, relativeChange)
struct PGOAgentStatus {
  unsigned agentID;
  PGOAgentState state;
  unsigned instanceNumber, iterationNumber;
  bool readyToTerminate;
  double relativeChange;
  explicit PGOAgentStatus(unsigned id = 0, PGOAgentState s = PGOAgentState::WAIT_FOR_DATA, unsigned instance = 0,
                          unsigned iteration = 0, bool ready = false, double change = 0)
      : agentID(id), state(s), instanceNumber(instance), iterationNumber(iteration), readyToTerminate(ready),
        relativeChange(change) {}
};

enum class InitializationMethod { Odometry, Chordal, GNC_TLS };  // src/PGOAgentROSNode.cpp:106-112
struct ROptParameters {                                          // :85,90,96-100
  enum class ROptMethod { RTR, RGD };
  ROptMethod method = ROptMethod::RTR;
  bool verbose = false;
  double gradnorm_tol = 1e-2;
  double RGD_stepsize = 1e-3;
  bool RGD_use_preconditioner = true;
  double RTR_initial_radius = 100;
  unsigned RTR_iterations = 3, RTR_tCG_iterations = 50;
  // RGD with a backtracking (Armijo) line search from RGD_stepsize (dpgo_params_t::rgd_line_search; no wrapper call site
  // writes these: src/PGOAgentROSNode.cpp:86-97 sets method, RGD_stepsize, RGD_use_preconditioner only)
  bool RGD_line_search = false;
  unsigned RGD_ls_max_backoffs = 7;
  double RGD_ls_shrink = 0.5, RGD_ls_sigma = 1e-4;
};
struct ROPTResult {  // mLocalOptResult (:169-172)
  bool success = false;
  double fInit = 0, fOpt = 0, gradNormInit = 0, gradNormOpt = 0;
};
struct RobustCostParameters {  // :178-188, 196-210
  enum class Type { L2, L1, Huber, TLS, GM, GNC_TLS };
  Type costType = Type::L2;
  unsigned GNCMaxNumIters = 10000;
  double GNCBarc = 5.0, GNCMuStep = 1.4, GNCInitMu = 1e-4;
  double TLSThreshold = 10.0, HuberThreshold = 3.0;  // [UPSTREAM-RECALL]; no wrapper call site writes them
};

}  // namespace DPGO
