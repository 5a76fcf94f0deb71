"""where do the seconds of one regrid go (tools/probes/regrid_cost_probe.py: 7 s per regrid of the 256^3 three-level bubble)?  cProfile around VardenAMR.regrid and around the
step after it (the first step on new grids rebuilds every cached plan, descriptor set and graph), host wall time per Python-level call of the library."""
import cProfile
import pstats
import sys
import time
sys.path.insert(0, ".")
import torch
from varden_amd import driver
from varden_amd.capi import default_params
W = [[15, 15]] * 3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
visc = float(sys.argv[2]) if len(sys.argv) > 2 else 0.001
prm = default_params(cflfac=0.9, visc_coef=visc)
levels = driver.VardenAMR.tagged_grids(n, W, prm, max_levs=3, max_grid_size=256)
G = driver.VardenAMR(n, levels[0], W, params=prm, finer_levels=levels[1:], init_shrink=0.1, init_iter=1, do_initial_projection=1, max_grid_size=256, swap_state=True,
                     regrid_int=-1, max_levs=3)
for _ in range(2):
    G.step()
torch.cuda.synchronize()


def timed(label, fn):
    pr = cProfile.Profile()
    t0 = time.perf_counter(); pr.enable(); fn(); torch.cuda.synchronize(); pr.disable()
    print("== %s: %.1f ms" % (label, 1e3 * (time.perf_counter() - t0)), flush=True)
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)


timed("step on old grids", G.step)
timed("regrid", G.regrid)
print("boxes per level after the regrid: %s" % [len(b) for b in G.boxes], flush=True)
timed("first step on new grids", G.step)
timed("second step on new grids", G.step)
timed("regrid again", G.regrid)
timed("first step after it", G.step)
G.close()
