"""what does creating / destroying a state field cost (vdn_multifab_create: address range + physical memory + map + a clearing memset)?  On the three-level 256^3 hierarchy."""
import sys, time
sys.path.insert(0, ".")
import torch
from varden_amd import driver, boxlib as bl
from varden_amd.capi import default_params
W = [[15, 15]] * 3
prm = default_params(cflfac=0.9)
levels = driver.VardenAMR.tagged_grids(256, W, prm, max_levs=3, max_grid_size=256)
G = driver.VardenAMR(256, levels[0], W, params=prm, finer_levels=levels[1:], init_shrink=0.1, init_iter=1, do_initial_projection=0, max_grid_size=256, swap_state=True)
G.step(); torch.cuda.synchronize()
for rep in range(2):
    for lev in range(3):
        for nc, ng in ((1, 1), (3, 3), (3, 1)):
            t0 = time.perf_counter(); m = bl.MultiFab(G.mla, lev, nc, ng); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            m.destroy(); t3 = time.perf_counter()
            print("rep %d level %d nc %d ng %d: create returns after %.3f ms, its memset done after %.3f ms more, destroy %.3f ms" % (rep, lev, nc, ng, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)), flush=True)
G.close()
