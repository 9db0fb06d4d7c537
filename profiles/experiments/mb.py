import sys, os
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', '..'))
from dpgo_ros_amd import capi
import numpy as np
m,n=capi.read_g2o('data/sphere2500.g2o')
mp=capi.partition(m,n,5); T=capi.odometry_init(m,n); Y=capi.fixed_stiefel(5)
t=capi.Team.from_measurements(mp, capi.default_params(r=5,num_robots=5,method=1,acceleration=1,rgd_stepsize=0.1))
t.set_initial(T,Y)
names={0:'precond',1:'eval',2:'hess',3:'retract',4:'nest_pre(all)',5:'noop 1x64',6:'noop 256x256',7:'status(all)',8:'copy(all)'}
for rep in range(2):
  for w in range(9):
    ms,b=t.time_kernel(1,w,reps=300)
    print(names[w], "%.2f us"%(ms*1e3))
os.system("rocm-smi --showclocks | grep -E 'sclk|mclk' | head -4")
