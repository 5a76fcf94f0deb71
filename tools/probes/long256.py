"""the 256^3 bench bubble stepped until something gives: every 10 steps time, dt, min / max rho, max |u|, the projections' cycle counts (argv: nsteps [n])"""
import sys
sys.path.insert(0, ".")
import numpy as np
from varden_amd import driver, advance as adv
from varden_amd.capi import default_params
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
extra = {}
for a in sys.argv[3:]:                         # further vdn_params fields as name=value (e.g. hg_omega_pre1=0 hg_omega_pre2=0: plain damping in the pre-smoothing sweeps)
    k, v = a.split("="); extra[k] = float(v)
G = driver.Varden(n, [[15, 15]] * 3, default_params(cflfac=0.9, **extra), init_shrink=0.1, init_iter=1, swap_state=True)
for it in range(ns):
    try:
        G.step()
    except Exception as e:
        print("step %d failed: %s" % (it, str(e)[:200]), flush=True)
        break
    if it % 10 == 0 or it > ns - 3:
        s = G.sold[0].to_numpy()[3:-3, 3:-3, 3:-3]; u = G.uold[0].to_numpy()[3:-3, 3:-3, 3:-3]
        print("step %4d t %.5f dt %.3e rho [%.4f, %.4f] tracer [%.3f, %.3f] |u|max %.4f cycles %d %d" % (it, G.time, G.dt, s[..., 0].min(), s[..., 0].max(), s[..., 1].min(), s[..., 1].max(), np.abs(u).max(),
              adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0]), flush=True)
s = G.sold[0].to_numpy()[3:-3, 3:-3, 3:-3]; u = G.uold[0].to_numpy()[3:-3, 3:-3, 3:-3]
print("last good state: rho [%.4f, %.4f] |u|max %.4f finite %r" % (s[..., 0].min(), s[..., 0].max(), np.abs(u).max(), bool(np.isfinite(u).all())), flush=True)
