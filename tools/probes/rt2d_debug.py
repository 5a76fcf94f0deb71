"""inputs_RayleighTaylor_2d on four levels fails at step 31 (composite MAC solve: right-hand side inf): what is non-finite, and which variation avoids it?"""
import sys, re, os
sys.path.insert(0, ".")
import numpy as np
from varden_amd import inputs, advance as adv, driver, boxlib as bl
text = open("tests/golden/inputs/inputs_RayleighTaylor_2d").read()
text = re.sub(r"plot_int\s*=\s*\d+", "plot_int = 0", text); text = re.sub(r"chk_int\s*=\s*\d+", "chk_int = 0", text)
var = sys.argv[1] if len(sys.argv) > 1 else "base"
nz = 16
if var == "regrid2": text = re.sub(r"regrid_int\s*=\s*\d+", "regrid_int = 2", text)
if var == "noregrid": text = re.sub(r"regrid_int\s*=\s*\d+", "regrid_int = -1", text)
if var == "nz8": nz = 8
if var == "lev3": text = re.sub(r"max_levs\s*=\s*\d+", "max_levs = 3", text)
if var == "mgs32": text = text.replace("&PROBIN", "&PROBIN\n max_grid_size = 32")
if var == "inviscid": text = re.sub(r"visc_coef\s*=\s*[\d.]+", "visc_coef = 0.0", text)
nl, G = inputs.build(text, outdir="/tmp", extrude_nz=nz, extrude_zbc=[14, 14] if var == "zslip" else None)
last_boxes = None
try:
    for s in range(int(sys.argv[2]) if len(sys.argv) > 2 else 45):
        boxes_before = [list(b) for b in G.boxes]
        G.step()
        if adv.last_solver_stats("mac")[0] > 20 or adv.last_solver_stats("hg")[0] > 25:
            print("   step %d: FAC mac %d hg %d, boxes %s" % (G.istep, adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0], [len(b) for b in G.boxes]), flush=True)
    print(var, ": all steps fine, t = %.4f, boxes %s" % (G.time, [len(b) for b in G.boxes]))
except Exception as e:
    print(var, ": FAILED at step", G.istep, str(e)[-120:])
    print("  boxes before the step:", [len(b) for b in boxes_before], " now:", [len(b) for b in G.boxes], "regrids", G.nregrids)
    for name, mfs in (("uold", G.uold), ("sold", G.sold), ("gp", G.gp), ("unew", G.unew), ("snew", G.snew), ("p", G.p)):
        for n, m in enumerate(mfs):
            for i in range(m.nfabs()):
                a = m.to_numpy(i)
                g = m.ng
                v = a[g:-g, g:-g, g:-g] if g else a
                if not np.isfinite(v).all():
                    bad = np.argwhere(~np.isfinite(v))
                    print("  %s level %d box %d %s: %d non-finite VALID entries, first at %s" % (name, n, i, G.boxes[n][G.local[n][i]], len(bad), bad[0]))
                elif not np.isfinite(a).all():
                    bad = np.argwhere(~np.isfinite(a))
                    print("  %s level %d box %d %s: %d non-finite GHOST entries, first at %s (array index)" % (name, n, i, G.boxes[n][G.local[n][i]], len(bad), bad[0]))
    for n in range(1, G.nlev):
        print("  level %d boxes: %s" % (n, G.boxes[n]))
    import json
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(dict(boxes=[[[list(b[0]), list(b[1])] for b in lb] for lb in G.boxes], nz=nz, step=G.istep), open("gpurun_out/rt_fail_boxes_%s.json" % var, "w"))
