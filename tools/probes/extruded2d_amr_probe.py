"""the reference's 2-D inputs (all adaptive) through the extruded hierarchy: python tools/probes/extruded2d_amr_probe.py <inputs file> [nsteps | 0 = the file's max_step] [print every]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from varden_amd import inputs, advance as adv
text = open(sys.argv[1]).read()
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
every = int(sys.argv[3]) if len(sys.argv) > 3 else 1
t0 = time.perf_counter()
worst = [0.0, 0.0]
def rep(G):
    umax = max(np.abs(G.uold[n].to_numpy(i)[3:-3, 3:-3, 3:-3, :2]).max() for n in range(G.nlev) for i in range(0, G.uold[n].nfabs(), 5))
    zs = max(np.abs(G.uold[n].to_numpy(i)[3:-3, 3:-3, 3:-3] - G.uold[n].to_numpy(i)[3:-3, 3:-3, 3:4]).max() for n in range(G.nlev) for i in range(0, G.uold[n].nfabs(), 5))
    w = max(np.abs(G.uold[n].to_numpy(i)[3:-3, 3:-3, 3:-3, 2]).max() for n in range(G.nlev) for i in range(0, G.uold[n].nfabs(), 5))
    worst[0], worst[1] = max(worst[0], zs / umax), max(worst[1], w / umax)
    if G.istep % every == 0:
        rho = G.slice2d(G.sold)[0][..., 0]
        print("step %3d t %.5f dt %.3e boxes %s FAC mac %2d hg %2d max|u| %.3e |w|/|u| %.1e z-spread/|u| %.1e rho in [%.4f, %.4f] regrids %d  %.1f s" % (G.istep, G.time, G.dt, [len(b) for b in G.boxes],
              adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0], umax, w / umax, zs / umax, np.nanmin(rho), np.nanmax(rho), G.nregrids, time.perf_counter() - t0), flush=True)
nl, G = inputs.run(text, nsteps=nsteps or None, report=rep, outdir="/tmp")
print("done: %d steps, t = %.5f, worst z-spread / |u| %.2e, worst |w| / |u| %.2e, %.1f s" % (G.istep, G.time, worst[0], worst[1], time.perf_counter() - t0))
G.close()
