python - <<'P'
import numpy as np, os, tempfile
from dpgo_ros_amd import capi
from oracle import oracle as O
from tests.util import load, add_outliers, params_pair
N=3
m,_,n=load("smallGrid3D",1)
mo=add_outliers(m,n,frac=0.1,seed=0)
mp=O.partition(mo,n,N)
T,Y=O.odometry_init(mo,n),O.fixed_stiefel(5)
kw=dict(r=5,num_robots=N,method=capi.METHOD_RTR,gradnorm_tol=1e-2,robust_cost_type=capi.COST_GNC_TLS,gnc_barc=3.0,gnc_mu_step=2.0,gnc_init_mu=1e-2,robust_opt_num_weight_updates=3,robust_opt_inner_iters=2*N,robust_opt_min_convergence_ratio=0.97,rel_change_tol=0.05,max_num_iters=200)
ph,po=params_pair(**kw)
def mk():
    t=capi.Team.from_measurements(mp.view(capi.MEAS_DTYPE),ph); t.set_initial(T,Y); return t
for total in (3, 6, 7, 9, 12, 13, 18, 19, 30, 300):
    a=mk(); b=mk()
    d=tempfile.mkdtemp()
    b.set_iteration_log(d)
    ra=a.run_schedule(total); rb=b.run_schedule(total)
    print(total, ra, rb, np.abs(a.global_X()-b.global_X()).max(), [np.abs(a.agents[i].measurements()["weight"]-b.agents[i].measurements()["weight"]).max() for i in range(N)])
    a.close(); b.close()
P
