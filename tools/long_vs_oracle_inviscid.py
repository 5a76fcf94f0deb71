"""the bench's own configuration (inviscid bubble between walls, cflfac 0.9) over a long run against the oracle: python tools/long_vs_oracle_inviscid.py [n=128] [nsteps=160].
The heavy blob reaches the floor near t = 0.34; from there the density leaves its initial bounds (no viscosity, no limiter on rho in the reference's update either).
Prints every 10 steps; stops at the first failed solve on either side."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle import voracle as vo
from varden_amd import driver, advance as adv
from varden_amd.capi import default_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 160
W = [[15, 15]] * 3
kw = dict(prob_type=1, grav=-9.8, init_shrink=0.1, init_iter=1)
G = driver.Varden(n, W, default_params(cflfac=0.9), **kw)
O = vo.Sim(n, W, default_params(cflfac=0.9), **kw)
t0 = time.time()
for it in range(ns):
    try:
        G.step()
    except Exception as e:
        print("GPU step %d failed: %s" % (it, str(e)[:160]), flush=True); break
    try:
        O.step()
    except Exception as e:
        print("oracle step %d failed: %s" % (it, str(e)[:160]), flush=True); break
    if it % 10 == 0 or it == ns - 1:
        g, o = G.snew[0].to_numpy()[3:-3, 3:-3, 3:-3], O.snew.valid()
        ug, uo = G.unew[0].to_numpy()[3:-3, 3:-3, 3:-3], O.unew.valid()
        print("step %4d t %.5f / %.5f rho gpu [%.4f, %.4f] oracle [%.4f, %.4f] |u|max %.3f / %.3f  rel diff u %.2e rho %.2e  cycles gpu %r oracle %r  (%.0f s)" %
              (it, G.time, O.time, g[..., 0].min(), g[..., 0].max(), o[..., 0].min(), o[..., 0].max(), np.abs(ug).max(), np.abs(uo).max(),
               np.abs(ug - uo).max() / np.abs(uo).max(), np.abs(g[..., 0] - o[..., 0]).max() / np.abs(o[..., 0]).max(),
               (adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0]), (int(O.mgstat[0].cycles), int(O.mgstat[1].cycles)), time.time() - t0), flush=True)
