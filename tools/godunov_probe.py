"""rocprofv3 target: the Godunov kernels (velpred + mkflux velocity/scalars) at n^3 with wall bcs, smooth data."""
import sys
sys.path.insert(0, ".")
import numpy as np
from varden_amd import advance as adv, boxlib as bl, capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bl.initialize(capi.default_params(), 0, 1, 0)
lo, hi = (0, 0, 0), (n - 1,) * 3
mla = bl.MLLayout([(lo, hi)], [[(lo, hi)]])
bct = bl.BCTower(mla, [[bl.SLIP_WALL] * 2] * 3)
dx = [1.0 / n] * 3
x = (np.arange(-3, n + 3) + 0.5) / n
X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
u = bl.MultiFab(mla, 0, 3, 3); s = bl.MultiFab(mla, 0, 2, 3)
ua = np.stack([np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y), -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y) * np.cos(np.pi * Z), 0.3 * np.sin(np.pi * Z) + 0 * X], axis=-1)
u.from_numpy(ua)
s.from_numpy(np.stack([1.0 + 0.5 * np.exp(-30 * ((X - .5) ** 2 + (Y - .5) ** 2 + (Z - .5) ** 2))] * 2, axis=-1))
umac = [bl.MultiFab(mla, 0, 1, 1, tuple(1 if t == d else 0 for t in range(3))) for d in range(3)]
force3 = bl.MultiFab(mla, 0, 3, 1); force2 = bl.MultiFab(mla, 0, 2, 1); mac_rhs = bl.MultiFab(mla, 0, 1, 1)
force3.setval(0.1, all=True); force2.setval(0.0, all=True); mac_rhs.setval(0.0, all=True)
ue = [bl.MultiFab(mla, 0, 3, 0, tuple(1 if t == d else 0 for t in range(3))) for d in range(3)]
uf = [bl.MultiFab(mla, 0, 3, 0, tuple(1 if t == d else 0 for t in range(3))) for d in range(3)]
se = [bl.MultiFab(mla, 0, 2, 0, tuple(1 if t == d else 0 for t in range(3))) for d in range(3)]
sf = [bl.MultiFab(mla, 0, 2, 0, tuple(1 if t == d else 0 for t in range(3))) for d in range(3)]
dt = 0.4 / n
import time
lib = capi.load()
for it in range(nl):
    lib.vdn_device_synchronize(); t0 = time.time()
    adv.velpred(u, umac, force3, dx, dt, bct)
    for m in umac:
        m.fill_boundary()
    lib.vdn_device_synchronize(); t1 = time.time()
    adv.mkflux(u, ue, uf, umac, force3, mac_rhs, dx, dt, bct, True, [0, 0, 0])
    lib.vdn_device_synchronize(); t2 = time.time()
    adv.mkflux(s, se, sf, umac, force2, mac_rhs, dx, dt, bct, False, [1, 0])
    lib.vdn_device_synchronize(); t3 = time.time()
    print("iter %d: velpred %.3f ms  mkflux(vel) %.3f ms  mkflux(scal) %.3f ms" % (it, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)), flush=True)
print("checksum", float(np.abs(ue[0].to_numpy()).sum()), float(np.abs(se[2].to_numpy()).sum()))
