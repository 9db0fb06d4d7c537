"""spread of the free-running asynchronous run across two processes (tests/test_gpu_distributed.py::test_peer_access_free_running_asynchronous_mode_descends):
final cost over initial cost and over the synchronous reference, run after run"""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tests.test_gpu_distributed as T
from oracle import oracle as O
def main():
    N, mp, n, Tm, kw = T._problem("peer_free")
    ref = O.Team(mp, n, O.default_params(r=5, num_robots=N, **kw))
    ref.set_initial(Tm, O.fixed_stiefel(5))
    for _ in range(300):
        ref.exchange_all()
        for a in ref.agents:
            a.iterate(True)
    ref.exchange_all()
    rc = ref.cost()
    for k in range(12):
        try:
            outs = T._spawn("peer_free")
            c0, c = float(outs[0]["cost0"]), float(outs[0]["cost"])
            print("run %d: c/c0 %.4f  c/ref %.4f" % (k, c / c0, c / rc), flush=True)
        except Exception as e:
            print("run %d: EXC %r" % (k, e), flush=True)


if __name__ == "__main__":
    main()
