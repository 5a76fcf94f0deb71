"""a hierarchy whose level 1 covers the WHOLE domain, run as the extruded copy, against the 2-D one-level run at the fine resolution: the fine level must be that run (to solver tolerance)"""
import sys
sys.path.insert(0, ".")
import numpy as np
from varden_amd import driver
from varden_amd.capi import default_params
nc, nz = 16, 8
for bc, prob, visc in (([[15, 15], [15, 15]], 1, 0.0), ([[-1, -1], [15, 15]], 1, 0.001), ([[11, 12], [15, 15]], 2, 0.001)):
    def prm(dm):
        p = default_params(dm=dm, cflfac=0.9, visc_coef=visc) if dm == 2 else default_params(cflfac=0.9, visc_coef=visc)
        for d in range(2):
            for s in range(2):
                if bc[d][s] == 11:
                    [p.u_bc, p.v_bc][d][d][s] = 1.0 if s == 0 else -1.0
                    p.rho_bc[d][s] = 1.0; p.trac_bc[d][s] = 0.5
        return p
    G2 = driver.Varden(2 * nc, [bc[0], bc[1], [0, 0]], prm(2), prob_type=prob, init_shrink=0.1, init_iter=1)
    for _ in range(3): G2.step()
    u2 = G2.gather_valid(G2.uold[0])[:, :, 0, :]; s2 = G2.gather_valid(G2.sold[0])[:, :, 0, :]; dt2 = G2.dt
    G2.close()
    fine = [((0, 0, 0), (2 * nc - 1, 2 * nc - 1, 2 * nz - 1))]
    G = driver.VardenAMR((nc, nc), fine, bc, params=prm(3), prob_type=prob, init_shrink=0.1, init_iter=1, do_initial_projection=1, extrude2d=nz)
    for _ in range(3): G.step()
    u = G.slice2d(G.uold)[1]; s = G.slice2d(G.sold)[1]
    print("bc %s prob %d visc %g: dt %r vs %r; fine level against the 2-D run: u %.2e, rho %.2e (max|u| %.2e)" % (bc, prob, visc, dt2, G.dt, np.abs(u[..., :2] - u2).max(), np.abs(s - s2).max(), np.abs(u2).max()), flush=True)
    G.close()
