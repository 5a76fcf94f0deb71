"""inputs_RayleighTaylor_2d as built (extruded, four levels): the finest level's state WITH its ghost cells after fill_state_ghosts, saved per box -- two processes, then `cmp` mode compares"""
import sys, os
sys.path.insert(0, ".")
import numpy as np
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2], allow_pickle=True).item(), np.load(sys.argv[3], allow_pickle=True).item()
    for k in sorted(a):
        x, y = a[k], b[k]
        d = np.argwhere(x.view(np.uint64) != y.view(np.uint64))
        if len(d):
            lo = d.min(axis=0); hi = d.max(axis=0)
            print("%s: %d entries differ; index range %s .. %s of shape %s; first %s: %r vs %r" % (k, len(d), lo.tolist(), hi.tolist(), x.shape, d[0].tolist(), x[tuple(d[0])], y[tuple(d[0])]))
    print("compared %d arrays" % len(a))
    sys.exit(0)
from varden_amd import inputs
text = open("tests/golden/inputs/inputs_RayleighTaylor_2d").read()
nl, G = inputs.build(text, outdir="/tmp")
G.fill_state_ghosts()
out = {}
for n in range(G.nlev):
    for i in range(G.sold[n].nfabs()):
        out["sold lev %d box %d %s" % (n, i, G.boxes[n][G.local[n][i]])] = G.sold[n].to_numpy(i).copy()
        out["uold lev %d box %d %s" % (n, i, G.boxes[n][G.local[n][i]])] = G.uold[n].to_numpy(i).copy()
np.save(sys.argv[1], out, allow_pickle=True)
G.close()
