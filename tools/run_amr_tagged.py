"""BASELINE.json configs[3] / [4] with the grids built from the tagged bubble (tag_boxes + make_new_grids, fixed afterwards):
base nc^3, max_levs levels; prints the boxes, then times advance_timestep on the hierarchy"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from varden_amd import advance as adv, capi
from varden_amd.driver import VardenAMR
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
max_levs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
mgs = int(sys.argv[4]) if len(sys.argv) > 4 else 256
walls = [[15, 15]] * 3
t0 = time.time()
levels = VardenAMR.tagged_grids(nc, walls, capi.default_params(cflfac=0.9), max_levs=max_levs, max_grid_size=mgs)
print("grid generation %.2f s" % (time.time() - t0))
cells = nc ** 3
for n, lb in enumerate(levels):
    c = sum(int(np.prod([b[1][d] - b[0][d] + 1 for d in range(3)])) for b in lb)
    cells += c
    print("level %d: %d boxes, %d cells (%.1f%% of its domain)" % (n + 1, len(lb), c, 100.0 * c / (nc << (n + 1)) ** 3))
    for b in lb[:12]:
        print("   ", b)
G = VardenAMR(nc, levels[0], walls, params=capi.default_params(cflfac=0.9), finer_levels=levels[1:])
print("dt %.4e, %d cells on %d levels" % (G.dt, cells, G.nlev), flush=True)
for it in range(nsteps):
    t0 = time.time(); G.step(); capi.load().vdn_device_synchronize(); t1 = time.time()
    tm = adv.last_step_timing()
    print("step %d: %.1f ms (mac %.1f hg %.1f scalar %.1f velocity %.1f)  FAC iterations mac %d hg %d  -> %.3e cells*steps/s" % (
        it, 1e3 * (t1 - t0), 1e3 * tm["mac"], 1e3 * tm["hg"], 1e3 * tm["scalar"], 1e3 * tm["velocity"],
        adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0], cells / (t1 - t0)), flush=True)
u = [G.unew[n].to_numpy(i) for n in range(G.nlev) for i in range(G.unew[n].nfabs())]
print("max |u| %.4e  finite %s" % (max(np.abs(a).max() for a in u), all(np.isfinite(a).all() for a in u)))
