"""the reference's 2-D inputs (all adaptive) through the extruded hierarchy: python tools/probes/extruded2d_amr_probe.py <inputs file> [nsteps]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from varden_amd import inputs, advance as adv
text = open(sys.argv[1]).read()
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
t0 = time.perf_counter()
def rep(G):
    u = G.slice2d(G.uold); 
    zs = max(np.abs(G.uold[n].to_numpy(i)[3:-3, 3:-3, 3:-3] - G.uold[n].to_numpy(i)[3:-3, 3:-3, 3:4]).max() for n in range(G.nlev) for i in range(0, G.uold[n].nfabs(), 7))
    print("step %3d t %.5f dt %.3e levels %d boxes %s  FAC mac %d hg %d  max|u| %.3e max|w| %.1e z-spread %.1e  %.1f s" % (G.istep, G.time, G.dt, G.nlev, [len(b) for b in G.boxes],
          adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0], max(np.nanmax(np.abs(a[..., :2])) for a in u), max(np.abs(G.uold[n].to_numpy(0)[..., 2]).max() for n in range(G.nlev)), zs, time.perf_counter() - t0), flush=True)
nl, G = inputs.run(text, nsteps=nsteps, report=rep, outdir="/tmp")
G.close()
