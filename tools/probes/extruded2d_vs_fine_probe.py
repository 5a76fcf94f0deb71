"""a two-level extruded hierarchy (32^2 base, the bubble refined) against the one-level 2-D runs at 32^2 and 64^2 with the same fixed dt: where the hierarchy is refined it should follow the fine run"""
import sys
sys.path.insert(0, ".")
import numpy as np
from varden_amd import driver
from varden_amd.capi import default_params
bc = [[15, 15], [15, 15]]
dt, nsteps = 2.0e-3, 20
runs = {}
for n in (32, 64):
    G = driver.Varden(n, [bc[0], bc[1], [0, 0]], default_params(dm=2, cflfac=0.9, visc_coef=0.001), prob_type=1, init_shrink=1.0, init_iter=1, fixed_dt=dt)
    for _ in range(nsteps): G.step()
    runs[n] = (G.gather_valid(G.uold[0])[:, :, 0, :], G.gather_valid(G.sold[0])[:, :, 0, :]); G.close()
prm = default_params(cflfac=0.9, visc_coef=0.001)
levels = driver.VardenAMR.tagged_grids((32, 32), bc, prm, prob_type=1, max_levs=2, max_grid_size=32, extrude2d=8)
G = driver.VardenAMR((32, 32), levels[0], bc, params=default_params(cflfac=0.9, visc_coef=0.001), prob_type=1, init_shrink=1.0, init_iter=1, do_initial_projection=1, extrude2d=8, fixed_dt=dt)
for _ in range(nsteps): G.step()
u, s = G.slice2d(G.uold), G.slice2d(G.sold)
m = np.isfinite(u[1][..., 0])
uf, sf = runs[64]
uc, sc = runs[32]
sc_up = np.repeat(np.repeat(sc, 2, axis=0), 2, axis=1); uc_up = np.repeat(np.repeat(uc, 2, axis=0), 2, axis=1)
print("refined cells: %d of %d; dt %g, %d steps, max|u| %.3e" % (m.sum(), m.size, dt, nsteps, np.abs(uf).max()))
print("level 1 of the hierarchy against the 64^2 run: u %.3e  rho %.3e" % (np.abs(u[1][..., :2][m] - uf[m]).max(), np.abs(s[1][..., 0][m] - sf[..., 0][m]).max()))
print("the 32^2 run (piecewise constant on the fine cells) against the 64^2 run, same cells: u %.3e  rho %.3e" % (np.abs(uc_up[m] - uf[m]).max(), np.abs(sc_up[..., 0][m] - sf[..., 0][m]).max()))
G.close()
