"""inputs_RayleighTaylor_2d: where does the extruded hierarchy leave the rails?  the same input on one level (2-D path) beside it"""
import sys, re
sys.path.insert(0, ".")
import numpy as np
from varden_amd import inputs, advance as adv
text = open("tests/golden/inputs/inputs_RayleighTaylor_2d").read()
mode = sys.argv[1]
if mode == "one":
    text = re.sub(r"max_levs\s*=\s*\d+", "max_levs = 1", text)
    text = re.sub(r"n_cellx\s*=\s*\d+", "n_cellx = %s" % sys.argv[2], text); text = re.sub(r"n_celly\s*=\s*\d+", "n_celly = %s" % sys.argv[2], text)
else:
    text = re.sub(r"max_levs\s*=\s*\d+", "max_levs = %s" % sys.argv[2], text)
text = re.sub(r"plot_int\s*=\s*\d+", "plot_int = 0", text); text = re.sub(r"chk_int\s*=\s*\d+", "chk_int = 0", text)
def rep(G):
    if G.istep % 5 == 0 or G.istep > 18:
        u = G.uold if isinstance(G.uold, list) else [G.uold]
        um = max(np.abs(m.to_numpy(i)).max() for m in G.uold for i in range(m.nfabs()))
        rm = [f(np.concatenate([m.to_numpy(i)[..., 0].ravel() for m in G.sold for i in range(m.nfabs())])) for f in (np.min, np.max)]
        print("step %3d t %.5f dt %.3e max|u| %.3e rho [%.4f, %.4f] FAC %d %d boxes %s" % (G.istep, G.time, G.dt, um, rm[0], rm[1], adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0],
              [len(b) for b in G.boxes] if hasattr(G, "nlev") else 1), flush=True)
try:
    nl, G = inputs.run(text, nsteps=int(sys.argv[3]), report=rep, outdir="/tmp")
except Exception as e:
    print("FAILED:", str(e)[:200])
