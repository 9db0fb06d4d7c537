#!/bin/bash
# build_variant.sh NAME "EXTRA_FLAGS" -- an instrumented libdpgo_hip.so under profiles/experiments/build/NAME/ (trace
# builds of the step / solve kernels); use it with DPGO_HIP_LIB=profiles/experiments/build/NAME/libdpgo_hip.so
set -e
cd "$(dirname "$0")/../.."
name=$1; flags=$2
out=profiles/experiments/build/$name
mkdir -p $out
src=dpgo_ros_amd/csrc
objs=""
for f in spmm precond step_fused step_deep step_persist rtr_fused linesearch pose_ops dense_inverse twolevel assembly solve capi chordal; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on $flags -c $src/$f.hip -o $out/$f.o &
  objs="$objs $out/$f.o"
done
for f in loader frame_align twolevel_plan rank_exchange; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on $flags -x hip -c $src/$f.cpp -o $out/$f.o &
  objs="$objs $out/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libdpgo_hip.so $objs -ldl
rm -f $out/*.o
echo $out/libdpgo_hip.so
