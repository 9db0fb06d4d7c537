import sys, os, time
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', '..'))
os.chdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from dpgo_ros_amd import capi
m,n=capi.read_g2o('data/sphere2500.g2o')
mp=capi.partition(m,n,5); T=capi.odometry_init(m,n); Y=capi.fixed_stiefel(5)
t=capi.Team.from_measurements(mp, capi.default_params(r=5,num_robots=5,method=0,acceleration=1,rtr_iterations=3,rtr_tcg_iterations=50,gradnorm_tol=1e-2,restart_interval=50))
t.set_initial(T,Y)
t.run(int(sys.argv[1]) if len(sys.argv)>1 else 100); t.synchronize()
t0=time.perf_counter(); t.run(100); t.synchronize(); print("ms/iter", (time.perf_counter()-t0)*10)
for rep in range(3):
    t0=time.perf_counter(); t.run(200); t1=time.perf_counter(); t.synchronize(); t2=time.perf_counter()
    print("rep", rep, "ms/iter", (t2-t0)*5, " host enqueue ms/iter", (t1-t0)*5)
