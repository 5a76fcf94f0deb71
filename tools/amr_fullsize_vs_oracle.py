"""BASELINE.json configs[3] / configs[4] AT FULL SIZE against the box-list oracle (the GPU suite does this on 32^3 .. 128^3 bases; this is the one-off at 256^3):
python tools/amr_fullsize_vs_oracle.py [max_levs=2] [nsteps=1] -- tagged grids of the 256^3 bubble, start-up + nsteps steps on both sides; prints the FAC
counts, dt and the largest relative differences.  Output kept in profiles/r05_amr<max_levs>_fullsize_vs_oracle.txt."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle import voracle as vo
from varden_amd import advance as adv, driver
from varden_amd.capi import default_params
import threading
def _beat():                                  # a line a minute: the oracle's start-up on three levels takes several (gpurun takes silence for a hang)
    while True:
        time.sleep(60); print("  ... %s" % time.strftime("%H:%M:%S"), flush=True)
threading.Thread(target=_beat, daemon=True).start()
ml = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nc, W = 256, [[15, 15]] * 3
prm = lambda: default_params(cflfac=0.9)     # noqa: E731
t0 = time.time()
levels = driver.VardenAMR.tagged_grids(nc, W, prm(), max_levs=ml, max_grid_size=256)
print("grids:", [len(lb) for lb in levels], "boxes on the refined levels", flush=True)
G = driver.VardenAMR(nc, levels[0], W, params=prm(), finer_levels=levels[1:], init_shrink=0.1, init_iter=1, do_initial_projection=1)
print("GPU start-up %.1f s" % (time.time() - t0), flush=True)
t0 = time.time()
O = vo.SimML(nc, levels, W, prm=prm(), init_shrink=0.1, init_iter=1, do_initial_projection=1)
print("oracle start-up %.1f s; initial projection FAC iterations GPU %r oracle %r; dt %r %r" % (time.time() - t0, G.initial_projection_stat[0], O.initial_projection_stat[0], G.dt, O.dt), flush=True)
assert G.initial_projection_stat[0] == O.initial_projection_stat[0] and G.dt == O.dt
for step in range(ns):
    t0 = time.time(); O.step(); to = time.time() - t0
    t0 = time.time(); G.step(); tg = time.time() - t0
    cg = (adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0]); co = (O.mgstat[0].cycles, O.mgstat[1].cycles)
    worst = {}
    for n in range(O.nlev):
        olo = O.levels[n].lo
        for nm, gm, om, g in (("u", G.uold[n], O.uold[n], 3), ("s", G.sold[n], O.sold[n], 3), ("gp", G.gp[n], O.gp[n], 1)):
            scale = max(float(np.abs(om.valid()).max()), 1e-300)
            for i in range(gm.nfabs()):
                lo, hi = gm.get_box(i)
                a = gm.to_numpy(i)[g:-g, g:-g, g:-g]
                b = om.valid()[tuple(slice(lo[d] - olo[d], hi[d] - olo[d] + 1) for d in range(3))]
                worst[nm] = max(worst.get(nm, 0.0), float(np.abs(a - b).max()) / scale)
    print("step %d: oracle %.1f s, GPU %.3f s; FAC iterations (MAC, HG) GPU %r oracle %r; dt equal %r; max rel. difference %s" %
          (step, to, tg, cg, co, G.dt == O.dt, {k: "%.2e" % v for k, v in worst.items()}), flush=True)
    assert cg == co and G.dt == O.dt and worst["u"] <= 1e-9 and worst["s"] <= 1e-9 and worst["gp"] <= 1e-6
print("OK")
