"""lockstep ASAPP ticks on tunnels alone (for rocprofv3 --kernel-trace): the launches of a tick"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from dpgo_ros_amd import capi
r = bench.asapp_leg(capi)
print(r["ms_per_tick"])
