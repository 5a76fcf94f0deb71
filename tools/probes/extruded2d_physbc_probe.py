"""multifab_physbc in 2-D against the z-uniform 3-D call: do the corner ghost cells of an outflow / wall pair agree?"""
import sys
sys.path.insert(0, ".")
import numpy as np
from varden_amd import boxlib as bl
from varden_amd.capi import default_params
n, nz = 16, 8
rng = np.random.default_rng(3)
for bc in ([[11, 12], [15, 15]], [[12, 12], [14, 14]], [[15, 15], [12, 12]], [[11, 12], [-1, -1]], [[15, 15], [15, 15]]):
    a2 = rng.standard_normal((n + 6, n + 6, 1, 4))
    out = {}
    for dm in (2, 3):
        p = default_params(dm=dm) if dm == 2 else default_params()
        for d in range(2):
            for s in range(2):
                if bc[d][s] == 11:
                    [p.u_bc, p.v_bc][d][d][s] = 1.0 if s == 0 else -1.0
                    p.rho_bc[d][s] = 1.0; p.trac_bc[d][s] = 0.5
        bl.initialize(p, 0, 1, 0)
        nn = (n, n, 1) if dm == 2 else (n, n, nz)
        phys = [bc[0], bc[1], [0, 0] if dm == 2 else [-1, -1]]
        lo, hi = (0, 0, 0), tuple(x - 1 for x in nn)
        mla = bl.MLLayout([(lo, hi)], [[(lo, hi)]], pmask=tuple(1 if phys[d][0] == -1 else 0 for d in range(3)))
        bct = bl.BCTower(mla, phys)
        u = bl.MultiFab(mla, 0, dm, 3); s = bl.MultiFab(mla, 0, 2, 3)
        if dm == 2:
            u.from_numpy(np.asfortranarray(a2[..., :2])); s.from_numpy(np.asfortranarray(a2[..., 2:]))
        else:
            u3 = np.zeros((n + 6, n + 6, nz + 6, 3), order="F"); u3[..., :2] = a2[:, :, 0, None, :2]
            s3 = np.zeros((n + 6, n + 6, nz + 6, 2), order="F"); s3[...] = a2[:, :, 0, None, 2:]
            u.from_numpy(u3); s.from_numpy(s3)
        u.fill_boundary(); s.fill_boundary()
        u.physbc(0, 0, dm, bct); s.physbc(0, dm, 2, bct)
        un, sn = u.to_numpy(0), s.to_numpy(0)
        out[dm] = (un[:, :, 0 if dm == 2 else 3, :2], sn[:, :, 0 if dm == 2 else 3, :])
        u.destroy(); s.destroy(); bct.destroy(); mla.destroy()
    du = np.abs(out[2][0] - out[3][0]); ds = np.abs(out[2][1] - out[3][1])
    print("bc %s: physbc 2-D vs extruded 3-D: u %.2e at %s, s %.2e at %s" % (bc, du.max(), np.unravel_index(du.argmax(), du.shape), ds.max(), np.unravel_index(ds.argmax(), ds.shape)), flush=True)
