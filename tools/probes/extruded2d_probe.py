"""is a 2-D run the xy-section of the z-uniform, z-periodic 3-D run of the same data?  (the 2-D hierarchies of the reference's four 2-D inputs: could they run on the 3-D machinery?)
One level: 2-D n^2 through dim2.hip against n x n x nz, periodic in z, gravity along y, the 2-D initial data extruded; a few steps; u, v, rho of plane k = 0, w, and the spread over z."""
import sys
sys.path.insert(0, ".")
import numpy as np
from varden_amd import driver, boxlib as bl
from varden_amd.capi import default_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 16
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
visc = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
for prob, bc in ((1, [[15, 15], [15, 15]]), (1, [[-1, -1], [15, 15]]), (2, [[11, 12], [15, 15]])):
    p2 = default_params(dm=2, cflfac=0.9, visc_coef=visc)
    p3 = default_params(cflfac=0.9, visc_coef=visc)
    for d in range(2):
        for s in range(2):
            if bc[d][s] == 11:
                for p in (p2, p3):
                    [p.u_bc, p.v_bc][d][d][s] = 1.0 if s == 0 else -1.0
                    p.rho_bc[d][s] = 1.0; p.trac_bc[d][s] = 0.5
    G2 = driver.Varden(n, [bc[0], bc[1], [0, 0]], p2, prob_type=prob, init_shrink=0.1, init_iter=1)
    res2 = []
    for _ in range(nsteps):
        G2.step()
    u2 = G2.gather_valid(G2.uold[0])[:, :, 0, :]; s2 = G2.gather_valid(G2.sold[0])[:, :, 0, :]; dt2 = G2.dt
    G2.close()
    # the same data extruded: arrays with 3 ghost layers, z uniform
    u0_2, s0_2 = driver.initdata_numpy((n, n), [1.0 / n] * 2, prob, 3, 2, dm=2)
    u0 = np.zeros((n + 6, n + 6, nz + 6, 3), order="F"); s0 = np.zeros((n + 6, n + 6, nz + 6, 2), order="F")
    u0[..., :2] = u0_2[:, :, 0, None, :]; s0[...] = s0_2[:, :, 0, None, :]
    G3 = driver.Varden((n, n, nz), [bc[0], bc[1], [-1, -1]], p3, prob_type=prob, prob_hi=(1.0, 1.0, nz / float(n)), init_shrink=0.1, init_iter=1, u0=u0, s0=s0, grav_dir=1, extruded2d=True)
    for _ in range(nsteps):
        G3.step()
    u3 = G3.gather_valid(G3.uold[0]); s3 = G3.gather_valid(G3.sold[0]); dt3 = G3.dt
    G3.close()
    sc = max(np.abs(u2).max(), 1e-300)
    print("prob %d bc %s visc %g, %d steps: dt %r vs %r; max|u3(k=0) - u2| / max|u2| = %.3e, rho %.3e; max|w| / max|u| = %.3e; spread over z %.3e" %
          (prob, bc, visc, nsteps, dt2, dt3, np.abs(u3[:, :, 0, :2] - u2).max() / sc, np.abs(s3[:, :, 0, :] - s2).max() / np.abs(s2).max(),
           np.abs(u3[..., 2]).max() / sc, np.abs(u3 - u3[:, :, :1, :]).max() / sc), flush=True)
