import sys, time
sys.path.insert(0, '.')
import numpy as np
from varden_amd import driver, advance as adv, capi
from varden_amd.capi import default_params
W=[[15,15]]*3
for ml in (2,3):
    prm=default_params(cflfac=0.9)
    levels=driver.VardenAMR.tagged_grids(256, W, prm, max_levs=ml, max_grid_size=256)
    G=driver.VardenAMR(256, levels[0], W, params=prm, finer_levels=levels[1:], init_shrink=0.1, init_iter=1, do_initial_projection=1, swap_state=True)
    G.step(); capi.load().vdn_device_synchronize()
    ph=dict(scalar=0.0, velocity=0.0, mac=0.0, hg=0.0, total=0.0); n=5
    t0=time.perf_counter()
    for _ in range(n):
        G.step()
        for k,v in adv.last_step_timing().items(): ph[k]+=v
    capi.load().vdn_device_synchronize()
    print(ml, "ms/step %.2f"%(1e3*(time.perf_counter()-t0)/n), {k: round(1e3*v/n,2) for k,v in ph.items()}, adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0], flush=True)
    G.close()
