import sys, os, time
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', '..'))
from dpgo_ros_amd import capi
ROOT=os.path.join(os.path.dirname(os.path.abspath(__file__)),'..','..')
m,n=capi.read_g2o(os.path.join(ROOT,'data/sphere2500.g2o'))
T=capi.odometry_init(m,n); Y=capi.fixed_stiefel(5)
for N in (5,2,1):
    mp=capi.partition(m,n,N) if N>1 else m
    t=capi.Team.from_measurements(mp, capi.default_params(r=5,num_robots=N,method=1))
    t0=time.perf_counter(); t.set_initial(T,Y); t.synchronize(); dt=time.perf_counter()-t0
    print("agents %d poses %d: finalize (Q assembly, upload, dense inverse) %.3f s total, %.3f s per agent"%(N,t.agents[0].n,dt,dt/N))
    t.close()
