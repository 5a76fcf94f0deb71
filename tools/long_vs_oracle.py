"""the falling blob of inputs_bubble_3d on one 32^3 level until it has hit the floor: HIP path against the CPU oracle over a long run,
through the phase in which the density leaves its initial bounds (python tools/long_vs_oracle.py [nsteps])"""
import sys
sys.path.insert(0, ".")
import numpy as np
from oracle import voracle as vo
from varden_amd import driver
from varden_amd.capi import default_params
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 100
phys = [[15, 15]] * 3
kw = dict(prob_type=1, grav=-9.8, init_shrink=0.1, init_iter=4)
prm = lambda: default_params(cflfac=0.9, visc_coef=0.001, prob_type=1)   # noqa: E731
G = driver.Varden(32, phys, prm(), do_initial_projection=1, **kw)
O = vo.Sim(32, phys, prm(), **kw)
for it in range(1, ns + 1):
    G.step(); O.step()
    if it % 25 == 0:
        g = G.snew[0].to_numpy(0)[3:-3, 3:-3, 3:-3, 0]
        o = O.snew.a[3:-3, 3:-3, 3:-3, 0]
        d = np.abs(g - o).max()
        print("step %3d  t %.4f / %.4f  rho gpu %.4f..%.4f  oracle %.4f..%.4f  max|diff| %.2e" % (it, G.time, O.time, g.min(), g.max(), o.min(), o.max(), d), flush=True)
