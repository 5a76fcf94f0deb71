"""does a long regridding run keep its memory?  256^3 base, three levels, visc_coef = 0.001, regrid_int = 2, N steps: free device memory (hipMemGetInfo through torch) and the arena every ten steps."""
import sys, time, ctypes as C
sys.path.insert(0, ".")
import torch
from varden_amd import driver, capi
from varden_amd.capi import default_params
W = [[15, 15]] * 3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
prm = default_params(cflfac=0.9, visc_coef=0.001)
levels = driver.VardenAMR.tagged_grids(n, W, prm, max_levs=3, max_grid_size=256)
G = driver.VardenAMR(n, levels[0], W, params=prm, finer_levels=levels[1:], init_shrink=0.1, init_iter=1, do_initial_projection=1, max_grid_size=256, swap_state=True, regrid_int=2, max_levs=3)
t0 = time.perf_counter()
for s in range(1, N + 1):
    G.step()
    if s % 10 == 0:
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        rb, pk = C.c_size_t(), C.c_size_t(); capi.load().vdn_arena_stats(C.byref(rb), C.byref(pk))
        print("step %3d: t = %.4f, %d regrids, boxes %s, device memory in use %.2f GB (arena %.1f GB mapped), %.1f s so far" %
              (s, G.time, G.nregrids, [len(b) for b in G.boxes], (total - free) / 2**30, rb.value / 2**30, time.perf_counter() - t0), flush=True)
G.close()
