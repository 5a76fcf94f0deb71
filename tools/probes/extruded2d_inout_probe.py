"""where does the z-uniform 3-D run leave the 2-D run with an inflow / outflow pair (tools/probes/extruded2d_probe.py: 5e-4 after four steps, 3e-13 between walls)?"""
import sys
sys.path.insert(0, ".")
import numpy as np
from varden_amd import driver
from varden_amd.capi import default_params
n, nz = 32, 8
def prm(dm, bc):
    p = default_params(dm=dm, cflfac=0.9) if dm == 2 else default_params(cflfac=0.9)
    for d in range(2):
        for s in range(2):
            if bc[d][s] == 11:
                [p.u_bc, p.v_bc][d][d][s] = 1.0 if s == 0 else -1.0
                p.rho_bc[d][s] = 1.0; p.trac_bc[d][s] = 0.5
    return p
for bc in ([[11, 12], [15, 15]], [[11, 12], [14, 14]], [[12, 12], [15, 15]], [[15, 15], [12, 12]], [[11, 12], [-1, -1]]):
    for nsteps in (0, 1, 3):
        kw = dict(prob_type=2, init_shrink=0.1, init_iter=1)
        G2 = driver.Varden(n, [bc[0], bc[1], [0, 0]], prm(2, bc), **kw)
        for _ in range(nsteps): G2.step()
        u2 = G2.gather_valid(G2.uold[0])[:, :, 0, :]; s2 = G2.gather_valid(G2.sold[0])[:, :, 0, :]; g2 = G2.gather_valid(G2.gp[0])[:, :, 0, :]; dt2 = G2.dt
        G2.close()
        u0_2, s0_2 = driver.initdata_numpy((n, n), [1.0 / n] * 2, 2, 3, 2, dm=2)
        u0 = np.zeros((n + 6, n + 6, nz + 6, 3), order="F"); s0 = np.zeros((n + 6, n + 6, nz + 6, 2), order="F")
        u0[..., :2] = u0_2[:, :, 0, None, :]; s0[...] = s0_2[:, :, 0, None, :]
        G3 = driver.Varden((n, n, nz), [bc[0], bc[1], [-1, -1]], prm(3, bc), prob_hi=(1.0, 1.0, nz / float(n)), u0=u0, s0=s0, grav_dir=1, extruded2d=True, **kw)
        for _ in range(nsteps): G3.step()
        u3 = G3.gather_valid(G3.uold[0]); s3 = G3.gather_valid(G3.sold[0]); g3 = G3.gather_valid(G3.gp[0]); dt3 = G3.dt
        G3.close()
        du = np.abs(u3[:, :, 0, :2] - u2); ij = np.unravel_index(du.argmax(), du.shape)
        print("bc %s steps %d: dt %s; u %.2e at %s, rho %.2e, gp %.2e" % (bc, nsteps, "equal" if dt2 == dt3 else "%r vs %r" % (dt2, dt3), du.max() / np.abs(u2).max(), ij,
              np.abs(s3[:, :, 0, :] - s2).max(), np.abs(g3[:, :, 0, :2] - g2).max() / max(np.abs(g2).max(), 1e-300)), flush=True)
