"""is a run of inputs-restart-regt (three levels, viscous, regrid_int = 2) reproducible from process to process?  Hash of the valid cells after every step."""
import sys, hashlib, tempfile
sys.path.insert(0, ".")
import numpy as np
from varden_amd import inputs
name = sys.argv[1] if len(sys.argv) > 1 else "inputs-restart-regt"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
text = open("tests/golden/inputs/" + name).read().replace("verbose = 1", "verbose = 0").replace("mg_verbose = 1", "mg_verbose = 0")
import re
text = re.sub(r"plot_int\s*=\s*\d+", "plot_int = 0", text); text = re.sub(r"chk_int\s*=\s*-?\d+", "chk_int = 0", text)
out = []
def rep(G):
    h = hashlib.sha256()
    for mfs, g in ((G.uold, 3), (G.sold, 3), (G.gp, 1), (G.p, 1)):
        for m in (mfs if isinstance(mfs, list) else [mfs]):
            for i in range(m.nfabs()):
                a = m.to_numpy(i)
                h.update(np.ascontiguousarray(a[g:-g, g:-g, g:-g]).tobytes())
    h2 = hashlib.sha256()
    for mfs in (G.uold, G.sold, G.gp, G.p):
        for m in (mfs if isinstance(mfs, list) else [mfs]):
            for i in range(m.nfabs()):
                h2.update(np.ascontiguousarray(m.to_numpy(i)).tobytes())
    out.append(h.hexdigest()[:8] + "/" + h2.hexdigest()[:8])
nl, G = inputs.run(text, nsteps, rep, outdir=tempfile.mkdtemp(dir="/tmp"))
print(name, " ".join(out))
