"""python tools/run_inputs.py <inputs file> [nsteps]: runs one of the reference's inputs files (exec/test/inputs_*) through the path"""
import os
import sys, time
sys.path.insert(0, ".")
import numpy as np
from varden_amd import advance as adv, inputs


def report(G):
    if G.istep % 5 == 0 or G.istep == 1:
        nb = [len(b) for b in G.boxes] if hasattr(G, "nlev") else [len(G.boxes)]
        print("step %3d  time %.5f  dt %.4e  boxes/level %s  mac %2d hg %2d cycles" % (G.istep, G.time, G.dt, nb, adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0]), flush=True)


text = open(sys.argv[1]).read()
t0 = time.time()
outdir = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/run_inputs"      # plot / checkpoint files land here
os.makedirs(outdir, exist_ok=True)
nl, G = inputs.run(text, int(sys.argv[2]) if len(sys.argv) > 2 else None, report, outdir=outdir)
print("%d steps in %.2f s" % (G.istep, time.time() - t0))
top = G.snew[-1] if hasattr(G, "nlev") else G.snew[0]
a = np.concatenate([top.to_numpy(i).reshape(-1, top.nc) for i in range(top.nfabs())])
print("finest level: rho in [%.6f, %.6f], finite %s" % (a[:, 0].min(), a[:, 0].max(), np.isfinite(a).all()))
