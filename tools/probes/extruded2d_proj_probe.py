"""the initial (nodal) projection of a random velocity field, 2-D against z-uniform 3-D, for the boundary pairs of extruded2d_inout_probe.py"""
import sys
sys.path.insert(0, ".")
import numpy as np
from varden_amd import driver
from varden_amd.capi import default_params
n, nz = 32, 8
rng = np.random.default_rng(5)
x = (np.arange(n + 6) - 2.5) / n
X, Y = np.meshgrid(x, x, indexing="ij")
for bc in ([[12, 12], [15, 15]], [[11, 12], [15, 15]], [[15, 15], [12, 12]], [[12, 12], [-1, -1]], [[15, 15], [15, 15]], [[12, 15], [15, 15]], [[12, 12], [14, 14]]):
    u2 = np.zeros((n + 6, n + 6, 1, 2), order="F"); s2 = np.ones((n + 6, n + 6, 1, 2), order="F")
    u2[:, :, 0, 0] = np.sin(3 * X) * np.cos(2 * Y) + 0.3 * X * Y; u2[:, :, 0, 1] = np.cos(2 * X + 1) * np.sin(3 * Y) - 0.2 * X
    s2[:, :, 0, 0] = 1.0 + 0.5 * np.exp(-20 * ((X - 0.6) ** 2 + (Y - 0.7) ** 2))
    res = {}
    for dm in (2, 3):
        p = default_params(dm=dm, cflfac=0.9) if dm == 2 else default_params(cflfac=0.9)
        if dm == 2:
            G = driver.Varden(n, [bc[0], bc[1], [0, 0]], p, prob_type=1, init_shrink=0.1, init_iter=0, do_initial_projection=1, u0=u2, s0=s2)
            res[dm] = G.gather_valid(G.uold[0])[:, :, 0, :]
        else:
            u3 = np.zeros((n + 6, n + 6, nz + 6, 3), order="F"); u3[..., :2] = u2[:, :, 0, None, :]
            s3 = np.zeros((n + 6, n + 6, nz + 6, 2), order="F"); s3[...] = s2[:, :, 0, None, :]
            G = driver.Varden((n, n, nz), [bc[0], bc[1], [-1, -1]], p, prob_type=1, prob_hi=(1.0, 1.0, nz / float(n)), init_shrink=0.1, init_iter=0, do_initial_projection=1, u0=u3, s0=s3, grav_dir=1)
            res[dm] = G.gather_valid(G.uold[0])[:, :, 0, :2]
        G.close()
    d = np.abs(res[2] - res[3])
    print("bc %s: projected u, 2-D vs extruded: %.2e at %s (max|u| %.2f)" % (bc, d.max(), np.unravel_index(d.argmax(), d.shape), np.abs(res[2]).max()), flush=True)
