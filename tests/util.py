"""shared helpers for the parity tests: seeded smooth fields on one box, oracle <-> HIP plumbing"""
import ctypes as C

import numpy as np

from oracle import voracle as vo
from varden_amd import advance as adv
from varden_amd import boxlib as bl
from varden_amd.capi import default_params

WALLS = [[15, 15]] * 3
SLIP = [[14, 14]] * 3
PER = [[-1, -1]] * 3
INOUT = [[11, 12], [14, 14], [15, 15]]           # inlet x-lo, outlet x-hi (inputs_advect_3d), slip y, no-slip z
MIXED = [[12, 11], [15, 14], [-1, -1]]           # outlet x-lo / inlet x-hi, mixed walls, periodic z

BC_SETS = {"walls": WALLS, "slip": SLIP, "periodic": PER, "inout": INOUT, "mixed": MIXED}


def params_for(phys, **kw):
    p = default_params(**kw)
    # inflow data for the INLET faces (inputs_advect_3d:41-49 style)
    for d in range(3):
        for s in range(2):
            if phys[d][s] == 11:
                vel = [p.u_bc, p.v_bc, p.w_bc][d]
                vel[d][s] = 1.0 if s == 0 else -1.0
                p.rho_bc[d][s] = 1.0
                p.trac_bc[d][s] = 0.5
    return p


def smooth_field(rng, shape, amp=1.0, base=0.0):
    """deterministic smooth-ish random field (sum of a few random Fourier modes + small noise)"""
    nx, ny, nz = shape[:3]
    x = (np.arange(nx) + 0.5) / nx
    y = (np.arange(ny) + 0.5) / ny
    z = (np.arange(nz) + 0.5) / nz
    X, Y, Z = np.meshgrid(x, y, z, indexing="ij")
    out = np.zeros(shape, order="F")
    for c in range(shape[3]):
        f = np.zeros((nx, ny, nz))
        for _ in range(4):
            k = rng.integers(1, 4, size=3)
            ph = rng.uniform(0, 2 * np.pi, size=3)
            f += rng.uniform(-1, 1) * np.sin(2 * np.pi * k[0] * X + ph[0]) * np.sin(2 * np.pi * k[1] * Y + ph[1]) * np.sin(2 * np.pi * k[2] * Z + ph[2])
        f += 0.05 * rng.standard_normal((nx, ny, nz))
        out[..., c] = base + amp * f
    return out


class Case:
    """one box [0,n)^3 with given physical bcs, on both the oracle and the GPU"""

    def __init__(self, n, phys, seed=0, prm=None, iso=False, **kw):
        self.n = tuple(n) if hasattr(n, "__len__") else (n, n, n)
        self.phys = phys
        self.prm = prm or params_for(phys, **kw)
        bl.initialize(self.prm, 0, 1, 0)
        self.rng = np.random.default_rng(seed)
        self.lo, self.hi = (0, 0, 0), tuple(x - 1 for x in self.n)
        self.pmask = [1 if phys[d][0] == -1 else 0 for d in range(3)]
        self.obc = vo.make_bc(phys, 3, self.prm.nscal)
        self.opm = vo.ivec(self.pmask)
        self.mla = bl.MLLayout([(self.lo, self.hi)], [[(self.lo, self.hi)]], pmask=self.pmask)
        self.bct = bl.BCTower(self.mla, phys)
        # iso: the same spacing in every direction (the projections' multigrid uses point smoothers and, like the
        # reference's inputs, is meant for dx = dy = dz); otherwise a deliberately anisotropic grid
        self.dx = [1.0 / max(self.n)] * 3 if iso else [1.0 / self.n[d] for d in range(3)]
        self.odx = vo.dvec(self.dx)
        self._mfs = []

    def ofab(self, ng, nc, nodal=(0, 0, 0), val=0.0):
        return vo.Fab(self.lo, self.hi, ng, nc, nodal, val)

    def gmf(self, ofab):
        """GPU multifab with the contents of an oracle fab"""
        mf = bl.MultiFab(self.mla, 0, ofab.nc, ofab.ng, ofab.nodal)
        mf.from_numpy(ofab.a)
        self._mfs.append(mf)
        return mf

    def random_state(self, with_ghost_fill=True, uamp=1.0):
        """u (3,ng3), s (nscal,ng3) with physical + periodic ghosts filled by the ORACLE (inputs to both paths)"""
        ns = self.prm.nscal
        u, s = self.ofab(3, 3), self.ofab(3, ns)
        u.a[...] = smooth_field(self.rng, u.a.shape, uamp)
        s.a[...] = smooth_field(self.rng, s.a.shape, 0.3, 2.0)
        if with_ghost_fill:
            L = vo.lib()
            L.vo_fill_boundary(u.ref, self.opm)
            L.vo_fill_boundary(s.ref, self.opm)
            L.vo_physbc(u.ref, 0, 0, 3, C.byref(self.obc), C.byref(self.prm))
            L.vo_physbc(s.ref, 0, 3, ns, C.byref(self.obc), C.byref(self.prm))
        return u, s

    def close(self):
        for m in self._mfs:
            m.destroy()
        self.bct.destroy()
        self.mla.destroy()


def assert_bits(a, b, what, region=None):
    """bit-exact comparison (NaN-aware) of two arrays on an optional region tuple of slices"""
    if region is not None:
        a, b = a[region], b[region]
    if not np.array_equal(a, b, equal_nan=True):
        d = np.abs(a - b)
        idx = np.unravel_index(np.nanargmax(d), d.shape)
        raise AssertionError("%s: not bit-identical; max |diff| = %.3e at %r (a=%r b=%r), %d mismatches"
                             % (what, np.nanmax(d), idx, a[idx], b[idx], int((a != b).sum())))
