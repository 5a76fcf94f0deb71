"""dm = 2 bubble (BASELINE.json configs[0]): GPU path vs the CPU oracle"""
import sys
sys.path.insert(0, ".")
import numpy as np
from varden_amd import boxlib as bl, capi
from varden_amd.driver import Varden
from oracle import voracle as vo
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W = [[15, 15], [15, 15]]
prm = capi.default_params(dm=2)
G = Varden(n, W, params=prm, init_iter=2, init_shrink=0.1)
S = vo.Sim(n, W, prm=capi.default_params(dm=2), dm=2, init_iter=2, init_shrink=0.1)
print("dt gpu %.17g cpu %.17g" % (G.dt, S.dt))
for it in range(nsteps):
    G.step(); S.step()
    ug = G.gather_valid(G.unew[0]); uc = S.unew.valid()
    sg = G.gather_valid(G.snew[0]); sc = S.snew.valid()
    from varden_amd import advance as adv
    print(it, "dt %.6e %.6e" % (G.dt, S.dt), "cyc gpu", adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0], "cpu", S.mgstat[0].cycles, S.mgstat[1].cycles,
          "du %.3e ds %.3e umax %.3e" % (np.abs(ug - uc).max(), np.abs(sg - sc).max(), np.abs(uc).max()))
