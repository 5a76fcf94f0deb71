"""what does regridding cost a run of configs[4]'s shape?  The reference's 3-D regression input regrids every SECOND step (exec/test/inputs_3d-regt: regrid_int = 2).  256^3 base, three levels,
visc_coef = 0.001: ten steps without and ten with regrid_int = 2, wall time per step and per regrid (tagging on the device, clustering, new layouts, fillpatch, every cached plan / descriptor set rebuilt)."""
import sys, time
sys.path.insert(0, ".")
import torch
from varden_amd import driver
from varden_amd.capi import default_params
W = [[15, 15]] * 3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ris = [int(a) for a in sys.argv[2:]] or [-1, 2]          # e.g. `... 256 2` under rocprofv3 --hip-trace --stats: the regridding run alone
for regrid_int in ris:
    prm = default_params(cflfac=0.9, visc_coef=0.001)
    levels = driver.VardenAMR.tagged_grids(n, W, prm, max_levs=3, max_grid_size=256)
    G = driver.VardenAMR(n, levels[0], W, params=prm, finer_levels=levels[1:], init_shrink=0.1, init_iter=1, do_initial_projection=1, max_grid_size=256, swap_state=True,
                         regrid_int=regrid_int, max_levs=3)
    G.step(); torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); G.step(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print("regrid_int %2d: steps (ms) %s  mean %.1f" % (regrid_int, " ".join("%.0f" % t for t in ts), sum(ts) / len(ts)), flush=True)
    import ctypes as C
    from varden_amd import capi
    rb, pk = C.c_size_t(), C.c_size_t(); capi.load().vdn_arena_stats(C.byref(rb), C.byref(pk))
    print("   arena: %.1f GB backed by memory, high-water mark %.1f GB; boxes per level now %s" % (rb.value / 2**30, pk.value / 2**30, [len(b) for b in G.boxes]), flush=True)
    G.close()
