import sys, os, ctypes, importlib
sys.path.insert(0,'/root/repo')
import numpy as np
import time
for tag in ("o2_nt1","o0_nt1"):
    from dpgo_ros_amd import capi
    capi._LIB=None; capi.LIB_PATH=os.path.join('/root/repo/dpgo_ros_amd','libdpgo_hip_%s.so'%tag)
    m,n=capi.read_g2o('/root/repo/data/sphere2500.g2o')
    mp=capi.partition(m,n,5); T=capi.odometry_init(m,n); Y=capi.fixed_stiefel(5)
    t=capi.Team.from_measurements(mp, capi.default_params(r=5,num_robots=5,method=1,acceleration=1,rgd_stepsize=0.1))
    t.set_initial(T,Y)
    ms,b=t.time_kernel(1,0,reps=300)
    t.run(100); t.synchronize(); t0=time.perf_counter(); t.run(2000); t.synchronize(); dt=time.perf_counter()-t0
    print(tag, "precond %.2f us"%(ms*1e3), "iter %.2f us"%(dt/2000*1e6))
    t.close()
