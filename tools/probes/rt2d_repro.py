"""the box lists on which inputs_RayleighTaylor_2d's composite MAC solve stalls (tools/probes/rt2d_debug.py dumps them), as FIXED grids from the initial data: does the start-up + a few steps stall too?
usage: rt2d_repro.py <json> [levels kept] [bcx: -1 periodic | 15 walls] [zbc: -1 | 14]"""
import sys, json, os
sys.path.insert(0, ".")
import numpy as np
from varden_amd import driver, advance as adv
from varden_amd.capi import default_params
d = json.load(open(sys.argv[1]))
keep = int(sys.argv[2]) if len(sys.argv) > 2 else len(d["boxes"])
bcx = int(sys.argv[3]) if len(sys.argv) > 3 else -1
zbc = int(sys.argv[4]) if len(sys.argv) > 4 else -1
boxes = [[(tuple(b[0]), tuple(b[1])) for b in lb] for lb in d["boxes"]][:keep]
prm = default_params(cflfac=0.9, visc_coef=float(os.environ.get("VISC", "0.01")), abort_on_max_iter=0)
G = driver.VardenAMR((32, 32), boxes[1], [[bcx, bcx], [15, 15]], params=prm, prob_type=3, finer_levels=boxes[2:], init_shrink=0.1, init_iter=3, do_initial_projection=1, extrude2d=d["nz"],
                     extrude_zbc=None if zbc == -1 else [zbc, zbc])
print("levels %d bcx %d zbc %d: after the start-up: MAC %s  HG %s" % (keep, bcx, zbc, adv.last_solver_stats("mac"), adv.last_solver_stats("hg")), flush=True)
nst = int(sys.argv[5]) if len(sys.argv) > 5 else 3
import os
trace_from = int(os.environ.get("TRACE_FROM", "100000"))
for s in range(nst):
    if s + 1 == trace_from:
        os.environ["VDN_MLCC_TRACE"] = "1"
    G.step()
    m, h = adv.last_solver_stats("mac"), adv.last_solver_stats("hg")
    if m[0] > 30 or h[0] > 40:
        print("   FIRST SLOW SOLVE at step %d t %.4f: MAC %s  HG %s" % (G.istep, G.time, m, h), flush=True)
        break
else:
    print("   no slow solve in %d steps (last MAC %s)" % (nst, adv.last_solver_stats("mac")), flush=True)
G.close()
