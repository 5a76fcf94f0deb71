"""precursor of BASELINE.json configs[4] on one GPU: base nc^3 + two nested refined levels over the bubble (fixed grids), timing of
advance_timestep on the three levels"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from varden_amd import advance as adv, capi
from varden_amd.driver import VardenAMR
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
q = nc // 4
b1 = ((2 * q,) * 3, (2 * (nc - q) - 1,) * 3)              # level 1: the central half of the domain in each direction (nc^3 cells)
n1 = 2 * nc; q1 = 3 * n1 // 8
b2 = ((2 * q1,) * 3, (2 * (n1 - q1) - 1,) * 3)            # level 2: the central quarter (nc^3 cells again), 2 q-cells inside level 1
G = VardenAMR(nc, [b1], [[15, 15]] * 3, params=capi.default_params(cflfac=0.9), finer_levels=[[b2]])
sz = [nc, b1[1][0] - b1[0][0] + 1, b2[1][0] - b2[0][0] + 1]
cells = sum(s ** 3 for s in sz)
print("levels: %d^3 + %d^3 + %d^3 cells, dt %.4e" % (sz[0], sz[1], sz[2], G.dt), flush=True)
for it in range(nsteps):
    t0 = time.time(); G.step(); capi.load().vdn_device_synchronize(); t1 = time.time()
    tm = adv.last_step_timing()
    print("step %d: %.1f ms (mac %.1f hg %.1f scalar %.1f velocity %.1f)  FAC iterations mac %d hg %d  -> %.3e cells*steps/s" % (
        it, 1e3 * (t1 - t0), 1e3 * tm["mac"], 1e3 * tm["hg"], 1e3 * tm["scalar"], 1e3 * tm["velocity"],
        adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0], cells / (t1 - t0)), flush=True)
s2 = G.snew[2].to_numpy()[3:-3, 3:-3, 3:-3, 0]
print("finest rho range %.6f %.6f  symmetry %.2e" % (s2.min(), s2.max(), np.abs(s2 - s2[::-1]).max()))
