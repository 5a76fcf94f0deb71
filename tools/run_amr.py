"""BASELINE.json configs[3]: base nc^3 + one refined level over the bubble (fixed grids), timing of advance_timestep on both levels"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from varden_amd import advance as adv, capi
from varden_amd.driver import VardenAMR
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
q = nc // 4
flo, fhi = (2 * q,) * 3, (2 * (nc - q) - 1,) * 3          # refine the central half of the domain in each direction
G = VardenAMR(nc, [(flo, fhi)], [[15, 15]] * 3, params=capi.default_params(cflfac=0.9))
cells = nc ** 3 + (fhi[0] - flo[0] + 1) ** 3
print("levels: %d^3 + %d^3 fine cells, dt %.4e" % (nc, fhi[0] - flo[0] + 1, G.dt), flush=True)
for it in range(nsteps):
    t0 = time.time(); G.step(); capi.load().vdn_device_synchronize(); t1 = time.time()
    tm = adv.last_step_timing()
    print("step %d: %.1f ms (mac %.1f hg %.1f scalar %.1f velocity %.1f)  FAC iterations mac %d hg %d  -> %.3e cells*steps/s" % (
        it, 1e3 * (t1 - t0), 1e3 * tm["mac"], 1e3 * tm["hg"], 1e3 * tm["scalar"], 1e3 * tm["velocity"],
        adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0], cells / (t1 - t0)), flush=True)
s1 = G.snew[1].to_numpy()[3:-3, 3:-3, 3:-3, 0]
print("fine rho range %.6f %.6f  symmetry %.2e" % (s1.min(), s1.max(), np.abs(s1 - s1[::-1]).max()))
