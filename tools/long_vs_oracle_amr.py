"""two-level fixed hierarchy (16^3 base, one 16^3 fine box over the bubble, viscous) over a long run: HIP path against the CPU oracle
(python tools/long_vs_oracle_amr.py [nsteps])"""
import sys
sys.path.insert(0, ".")
import numpy as np
from oracle import voracle as vo
from varden_amd import driver
from varden_amd.capi import default_params
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 120
walls = [[15, 15]] * 3
flo, fhi = (8, 8, 8), (23, 23, 23)
mk = lambda: default_params(cflfac=0.9, visc_coef=0.001)   # noqa: E731
kw = dict(init_iter=2, do_initial_projection=1)
O = vo.SimML(16, [(flo, fhi)], walls, prm=mk(), **kw)
G = driver.VardenAMR(16, [(flo, fhi)], walls, params=mk(), **kw)
for it in range(1, ns + 1):
    O.step(); G.step()
    if it % 20 == 0:
        out = []
        for n in range(2):
            g = G.snew[n].to_numpy(0)[3:-3, 3:-3, 3:-3, 0]
            o = O.snew[n].valid()[..., 0]
            out.append("L%d rho %.4f..%.4f |d| %.1e" % (n, g.min(), g.max(), np.abs(g - o).max()))
        print("step %3d t %.4f/%.4f  %s" % (it, G.time, O.time, "  ".join(out)), flush=True)
