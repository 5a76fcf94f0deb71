"""inputs-restart-regt: the run continued from chk00004 against the uninterrupted one -- which fields differ, by how much, from which step on?"""
import sys, os, tempfile
sys.path.insert(0, ".")
import numpy as np
from varden_amd import inputs
text = open("tests/golden/inputs/inputs-restart-regt").read().replace("verbose = 1", "verbose = 0").replace("mg_verbose = 1", "mg_verbose = 0")
for k, v in (a.split("=") for a in sys.argv[1:]):
    import re
    text = re.sub(r"%s\s*=\s*[-\w.]+" % k, "%s = %s" % (k, v), text)
out = tempfile.mkdtemp(dir="/tmp")
def snap(G):
    return {(nm, n, i): m.to_numpy(i).copy() for nm, mfs in (("u", G.uold), ("s", G.sold), ("gp", G.gp), ("p", G.p)) for n, m in enumerate(mfs) for i in range(m.nfabs())}, [list(b) for b in G.boxes], G.time, G.dt
A_hist = {}
def repA(G): A_hist[G.istep] = snap(G)
nl, A = inputs.run(text, None, repA, outdir=out); A.close()
B_hist = {}
def repB(G): B_hist[G.istep] = snap(G)
nl, B = inputs.run(text.replace("&PROBIN", "&PROBIN\n restart = 4"), None, repB, outdir=out); B.close()
for st in sorted(B_hist):
    a, b = A_hist[st], B_hist[st]
    print("step %d: boxes equal %s, time equal %s, dt equal %s" % (st, a[1] == b[1], a[2] == b[2], a[3] == b[3]))
    if a[1] != b[1]: continue
    worst = {}
    for k in a[0]:
        g = 3 if k[0] in ("u", "s") else 1
        x, y = a[0][k][g:-g, g:-g, g:-g], b[0][k][g:-g, g:-g, g:-g]
        d = np.abs(x - y).max() / max(np.abs(x).max(), 1e-300)
        worst[(k[0], k[1])] = max(worst.get((k[0], k[1]), 0.0), d)
    print("   worst relative difference per field and level:", {k: float("%.2e" % v) for k, v in sorted(worst.items())})

import hashlib
for tag, H in (("uninterrupted", A_hist), ("restarted", B_hist)):
    for st in (6, 7, 8):
        h = hashlib.sha256()
        for k in sorted(H[st][0]):
            g = 3 if k[0] in ("u", "s") else 1
            h.update(np.ascontiguousarray(H[st][0][k][g:-g, g:-g, g:-g]).tobytes())
        print("hash of the valid cells, %s step %d: %s" % (tag, st, h.hexdigest()[:16]))

# bitwise: where do the two runs first differ in BITS (signs of zeros included)?
for st in sorted(B_hist):
    a, b = A_hist[st], B_hist[st]
    rep_ = []
    for k in sorted(a[0]):
        g = 3 if k[0] in ("u", "s") else 1
        x, y = a[0][k][g:-g, g:-g, g:-g], b[0][k][g:-g, g:-g, g:-g]
        xb, yb = np.ascontiguousarray(x).view(np.uint64), np.ascontiguousarray(y).view(np.uint64)
        nd = int((xb != yb).sum())
        if nd:
            idx = np.argwhere(xb != yb)[0]
            rep_.append("%s lev %d box %d: %d entries differ in bits (first at %s: %r vs %r)" % (k[0], k[1], k[2], nd, tuple(idx), x[tuple(idx)], y[tuple(idx)]))
    print("step %d: %d arrays differ in bits" % (st, len(rep_)))
    for r in rep_[:6]: print("    " + r)
