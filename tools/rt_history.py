"""density extrema history of inputs_RayleighTaylor_3d: python tools/rt_history.py [max_levs] [nsteps]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from varden_amd import inputs
text = open("tests/golden/inputs/inputs_RayleighTaylor_3d").read().replace("verbose = 1", "verbose = 0")
ml = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 150
import re
text = re.sub(r"max_levs\s*=\s*\d+", "max_levs = %d" % ml, text)
text = re.sub(r"plot_int\s*=\s*\d+", "plot_int = 0", text)
text = re.sub(r"chk_int\s*=\s*\d+", "chk_int = 0", text)


def report(G):
    if G.istep % 10 == 0:
        lv = range(G.nlev) if hasattr(G, "nlev") else [0]
        mm = [(G.snew[n].min_max(0)) for n in lv]
        um = [max(G.unew[n].norm_inf(c, 1) for c in range(3)) for n in lv]
        print("step %3d t %.4f dt %.3e rho %s  |u| %s" % (G.istep, G.time, G.dt, ["%.3f..%.3f" % m for m in mm], ["%.2e" % u for u in um]), flush=True)


inputs.run(text, ns, report, outdir="/tmp")
