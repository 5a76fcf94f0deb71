import sys, time
sys.path.insert(0, ".")
import numpy as np
from varden_amd import advance as adv, driver
from varden_amd.capi import default_params
G = driver.Varden(256, [[15, 15]] * 3, default_params(cflfac=0.9, visc_coef=0.001), prob_type=1, grav=-9.8, init_shrink=0.1, init_iter=1, swap_state=True)
t0 = time.time()
for it in range(120):
    G.step()
    if it % 20 == 0 or it == 119:
        s = G.sold[0].to_numpy()[3:-3, 3:-3, 3:-3]
        u = G.uold[0].to_numpy()[3:-3, 3:-3, 3:-3]
        print("step %d t %.4f dt %.2e rho [%.4f, %.4f] |u|max %.3f cycles %r finite %r (%.1f s)" % (it, G.time, G.dt, s[..., 0].min(), s[..., 0].max(), np.abs(u).max(), (adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0]), bool(np.isfinite(u).all()), time.time() - t0), flush=True)
