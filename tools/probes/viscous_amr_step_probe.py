"""a VISCOUS step of the tagged three-level hierarchy (256^3 base, visc_coef = 0.001 as every 3-D input of exec/test has it) beside the inviscid one bench.py times:
usage (under rocprofv3 --kernel-trace --stats for the kernel split): python tools/probes/viscous_amr_step_probe.py [n=256] [nsteps=3] [max_levs=3] [visc=0.001]"""
import sys, time
sys.path.insert(0, ".")
import torch
from varden_amd import advance as adv, driver
from varden_amd.capi import default_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ml = int(sys.argv[3]) if len(sys.argv) > 3 else 3
viscs = [float(sys.argv[4])] if len(sys.argv) > 4 else [0.0, 0.001]
W = [[15, 15]] * 3
for visc in viscs:
    prm = default_params(cflfac=0.9, visc_coef=visc)
    levels = driver.VardenAMR.tagged_grids(n, W, prm, max_levs=ml, max_grid_size=256)
    G = driver.VardenAMR(n, levels[0], W, params=prm, finer_levels=levels[1:], init_shrink=0.1, init_iter=1, do_initial_projection=1, max_grid_size=256, swap_state=True)
    for _ in range(2):
        G.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ph = {}
    for _ in range(ns):
        G.step()
        for k, v in adv.last_step_timing().items():
            ph[k] = ph.get(k, 0.0) + v
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("visc_coef %g, %d levels: %.2f ms per step; phases %s; solver stats %s" % (visc, ml, 1e3 * el / ns, {k: round(1e3 * v / ns, 2) for k, v in ph.items()},
          {w: adv.last_solver_stats(w) for w in ("mac", "hg")}), flush=True)
    G.close()
