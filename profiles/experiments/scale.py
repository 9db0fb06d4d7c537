import sys, os, time
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', '..'))
from dpgo_ros_amd import capi
m,n=capi.read_g2o('data/sphere2500.g2o')
T=capi.odometry_init(m,n); Y=capi.fixed_stiefel(5)
print("| agents | poses/agent | M bytes | precond us/launch | GB/s | frac of 8 TB/s | eval us | eval GB/s |")
print("|---|---|---|---|---|---|---|---|")
MODE=int(os.environ.get('PRECOND_MODE','0'))
for N in (8,5,4,3,2,1):
    mp=capi.partition(m,n,N) if N>1 else m
    t=capi.Team.from_measurements(mp, capi.default_params(r=5,num_robots=N,method=1,acceleration=0,rgd_stepsize=0.1,precond_mode=MODE))
    t.set_initial(T,Y)
    a=0
    ms,b=t.time_kernel(a,0,reps=100)
    ms2,b2=t.time_kernel(a,1,reps=200)
    print("| %d | %d | %.1f MB | %.2f | %.0f | %.3f | %.2f | %.0f |"%(N, t.agents[a].n, b/1e6, ms*1e3, b/(ms*1e-3)/1e9, b/(ms*1e-3)/8e12, ms2*1e3, b2/(ms2*1e-3)/1e9))
    t.close()
