"""the reference's own test input exec/test/inputs_bubble_3d, parameter for parameter: 32^3 base, max_levs = 2, regrid_int = 2,
max_grid_size = 16, cluster_min_eff 0.9 / min_width 4 / blocking_factor 4, init_iter = 1, do_initial_projection = 1, cflfac 0.9,
init_shrink 0.1, visc_coef 0.001, grav -9.8, all walls (bc 15).  Not reproduced: plot/checkpoint files (out of scope)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from varden_amd import advance as adv, capi
from varden_amd.driver import VardenAMR
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
prm = capi.default_params(cflfac=0.9, visc_coef=0.001, diff_coef=0.0)
walls = [[15, 15]] * 3
levels = VardenAMR.tagged_grids(32, walls, prm, prob_type=1, max_levs=2, buf_wid=2, max_grid_size=16)
G = VardenAMR(32, levels[0], walls, params=prm, prob_type=1, grav=-9.8, init_shrink=0.1, regrid_int=2, max_levs=2, max_grid_size=16,
              init_iter=1, do_initial_projection=1)
print("level 1: %d boxes; initial projection %d iterations; dt %.6e" % (len(G.boxes[1]), G.initial_projection_stat[0], G.dt), flush=True)
t0 = time.time()
for it in range(nsteps):
    G.step()
    if it % 5 == 4 or it == 0:
        u = np.concatenate([G.unew[1].to_numpy(i)[3:-3, 3:-3, 3:-3].reshape(-1, 3) for i in range(G.unew[1].nfabs())])
        print("step %3d  time %.5f  dt %.4e  boxes %3d  regrids %2d  FAC mac %2d hg %2d  max|w| %.5e" % (
            G.istep, G.time, G.dt, len(G.boxes[1]), G.nregrids, adv.last_solver_stats("mac")[0], adv.last_solver_stats("hg")[0], np.abs(u[:, 2]).max()), flush=True)
capi.load().vdn_device_synchronize()
print("%d steps in %.2f s" % (nsteps, time.time() - t0))
s0 = G.snew[0].to_numpy(0)[3:-3, 3:-3, 3:-3, 0]
print("coarse rho range %.6f %.6f  x-symmetry %.2e  y-symmetry %.2e" % (s0.min(), s0.max(), np.abs(s0 - s0[::-1]).max(), np.abs(s0 - s0[:, ::-1]).max()))
