"""API-conformance check of the facade: the reference wrapper's OWN translation units
(/root/reference/src/*.cpp, read where they lie, never copied) must type-check against include/DPGO/* with the
stand-in `DPGO::Matrix`.  ROS, tf, glog and pose_graph_tools are absent from this image, so minimal declarations stand in
for them (tests/cpp/ros_stubs/README.md) and the message headers are generated from the reference's .msg / .srv data
files at test time.  `g++ -fsyntax-only`: nothing is built, linked or run, and this pins no numbers -- it finds names,
signatures and Matrix operations the wrapper needs and the facade lacks.  Skipped where /root/reference does not exist
(the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference wrapper sources not present")


@pytest.fixture(scope="module")
def msg_headers(tmp_path_factory):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gen_ros_msgs
    out = str(tmp_path_factory.mktemp("rosgen"))
    gen_ros_msgs.generate(REF, out)
    return out


@pytest.mark.parametrize("tu", ["utils.cpp", "PGODatasetPublisherNode.cpp", "PGOAgentROS.cpp", "PGOAgentROSNode.cpp"])
def test_wrapper_translation_unit_type_checks_against_the_facade(tu, msg_headers):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-DDPGO_FACADE_NO_EIGEN", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "tests", "cpp", "ros_stubs"), "-I" + msg_headers, "-I" + os.path.join(REF, "include"),
           os.path.join(REF, "src", tu)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    errors = [l for l in res.stderr.splitlines() if "error" in l]
    assert res.returncode == 0, "\n".join(errors[:20])


def test_assignable_pose_blocks_have_effect():
    """the wrapper writes `X.rotation() = YLift` and `poses.rotation(i) = R` (src/PGOAgentROS.cpp:299-300,1464-1465):
    with the stand-in Matrix these must modify the pose, not a temporary"""
    src = r'''
#include <DPGO/DPGO_types.h>
#include <cstdio>
using namespace DPGO;
int main() {
  LiftedPose X(5, 3);
  Matrix Y = Matrix::Zero(5, 3);
  Y(4, 2) = 7.0;
  X.rotation() = Y;
  X.translation() = Vector::Zero(5);
  PoseArray P(3, 2);
  Matrix R(3, 3);
  R << 1, 2, 3, 4, 5, 6, 7, 8, 9;
  P.rotation(1) = R;
  Matrix t(3, 1);
  t << 10, 11, 12;
  P.translation(1) = t;
  const Matrix &D = P.getData();
  const bool ok = X.getData()(4, 2) == 7.0 && X.getData()(0, 0) == 0.0 && D(0, 4 + 1) == 2.0 && D(1, 4 + 0) == 4.0 && D(2, 7) == 12.0;
  std::printf("%d\n", ok ? 1 : 0);
  return ok ? 0 : 1;
}
'''
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        cpp = os.path.join(td, "t.cpp")
        open(cpp, "w").write(src)
        exe = os.path.join(td, "t")
        subprocess.check_call(["g++", "-std=c++17", "-DDPGO_FACADE_NO_EIGEN", "-I" + os.path.join(ROOT, "include"), cpp, "-o", exe])
        assert subprocess.run([exe]).returncode == 0
