"""Host-side mirror of the FBoxLib containers the reference hot path takes, over the C-ABI.

Names follow the BoxLib Fortran modules the reference ``use``s:
``ml_layout`` (here :class:`MLLayout`), ``multifab`` (:class:`MultiFab`, with ``nfabs``, ``get_box``,
``dataptr``, ``setval``, ``copy_c``, ``norm_inf``, ``fill_boundary``) and ``bc_tower``
(:class:`BCTower`, reference src/define_bc_tower.f90).  All data lives in HBM; ``dataptr`` returns the
device address, ``to_numpy``/``from_numpy`` are explicit host copies in the BoxLib fab layout
``a[i, j, k, comp]`` (Fortran order, index 0 = lo-ng).
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import Box, check

# bc_module constants (values as used by exec/test/inputs_*)
PERIODIC, INTERIOR, INLET, OUTLET, SYMMETRY, SLIP_WALL, NO_SLIP_WALL = -1, 0, 11, 12, 13, 14, 15
REFLECT_ODD, REFLECT_EVEN, FOEXTRAP, EXT_DIR, HOEXTRAP = 20, 21, 22, 23, 24
BC_PER, BC_INT, BC_DIR, BC_NEU = -1, 0, 1, 2
# proj_parameters.f90
INITIAL_PROJECTION, DIVU_ITERS, PRESSURE_ITERS, REGULAR_TIMESTEP = 1, 2, 3, 4

_initialised = False
_dm = 3                      # dim_in of the current run (set by initialize); 2: fabs are one z-plane on the host


def initialize(params=None, rank=0, nranks=1, device=0):
    """boxlib_initialize + probin_init: bind the GPU and hand the runtime parameters over."""
    global _initialised, _dm
    lib = capi.load()
    prm = params if params is not None else capi.default_params()
    check(lib.vdn_init(C.byref(prm), rank, nranks, device))
    _initialised = True
    _dm = int(prm.dm)
    return prm


def set_extruded_2d(on=True):
    """the 3-D kernels run a z-uniform copy of a 2-D problem (include/varden_amd.h: vdn_set_extruded_2d); reset by initialize"""
    check(capi.load().vdn_set_extruded_2d(1 if on else 0))


def comm_get_unique_id():
    """rank 0: the 128-byte RCCL unique id to broadcast to the other ranks"""
    buf = C.create_string_buffer(128)
    check(capi.load().vdn_comm_get_unique_id(buf))
    return buf.raw


_comm_up = False


def comm_init(id128):
    """collective; a second call while the communicator is up is a no-op (drivers built one after another share it)"""
    global _comm_up
    if _comm_up:
        return
    check(capi.load().vdn_comm_init(C.c_char_p(bytes(id128))))
    _comm_up = True


def comm_allreduce_max(a):
    """MAX over the ranks of a float64 numpy array, in place (also the barrier of the file writers)"""
    a = np.ascontiguousarray(a, dtype=np.float64)
    check(capi.load().vdn_comm_allreduce_max(a.ctypes.data_as(C.POINTER(C.c_double)), a.size))
    return a


def comm_nranks():
    """ranks of the live RCCL communicator, read back from RCCL (1 when none is up)"""
    n = C.c_int(0)
    check(capi.load().vdn_comm_nranks(C.byref(n)))
    return int(n.value)


def comm_finalize():
    global _comm_up
    if _comm_up:
        check(capi.load().vdn_comm_finalize())
    _comm_up = False


def finalize():
    global _initialised
    if _initialised:
        check(capi.load().vdn_finalize())
        _initialised = False


def _box(lo, hi):
    b = Box()
    for d in range(3):
        b.lo[d], b.hi[d] = int(lo[d]), int(hi[d])
    return b


class MLLayout:
    """ml_layout: levels, refinement ratios, problem domains, boxes and their owner ranks."""

    def __init__(self, pd, boxes, owner=None, rr=None, pmask=(0, 0, 0)):
        """pd: list (per level) of (lo, hi); boxes: list (per level) of lists of (lo, hi)."""
        lib = capi.load()
        nlev = len(pd)
        pdarr = (Box * nlev)(*[_box(*p) for p in pd])
        flat = [b for lev in boxes for b in lev]
        barr = (Box * len(flat))(*[_box(*b) for b in flat])
        nb = (C.c_int * nlev)(*[len(lev) for lev in boxes])
        own = (C.c_int * len(flat))(*([0] * len(flat) if owner is None else [int(o) for lev in owner for o in lev]))
        rrflat = [] if rr is None else [int(x) for r in rr for x in r]
        rrarr = (C.c_int * max(1, len(rrflat)))(*(rrflat or [2]))
        pm = (C.c_int * 3)(*[int(x) for x in pmask])
        self.h = C.c_void_p()
        check(lib.vdn_layout_create(nlev, rrarr, pdarr, nb, barr, own, pm, C.byref(self.h)))
        self.nlevel, self.dim, self.pmask, self.pd, self.boxes = nlev, 3, tuple(pmask), pd, boxes

    def nlocal(self, lev=0):
        return capi.load().vdn_layout_nlocal(self.h, lev)

    def destroy(self):
        if self.h:
            capi.load().vdn_layout_destroy(self.h)
            self.h = None


class BCTower:
    """bc_tower built from the domain's physical bcs, as define_bc_tower.f90 does."""

    def __init__(self, mla, phys_bc):
        self.mla = mla
        flat = (C.c_int * 6)(*[int(phys_bc[d][s]) for d in range(3) for s in range(2)])
        self.h = C.c_void_p()
        check(capi.load().vdn_bc_tower_create(mla.h, flat, C.byref(self.h)))
        self.domain_bc = [[int(phys_bc[d][s]) for s in range(2)] for d in range(3)]

    def phys(self, lev, grid, d, s):
        return capi.load().vdn_bc_tower_phys(self.h, lev, grid, d, s)

    def adv(self, lev, grid, d, s, comp):
        return capi.load().vdn_bc_tower_adv(self.h, lev, grid, d, s, comp)

    def ell(self, lev, grid, d, s, comp):
        return capi.load().vdn_bc_tower_ell(self.h, lev, grid, d, s, comp)

    def destroy(self):
        if self.h:
            capi.load().vdn_bc_tower_destroy(self.h)
            self.h = None


class MultiFab:
    """multifab of one level, resident in HBM."""

    def __init__(self, mla, lev, nc, ng, nodal=None):
        self.mla, self.lev, self.nc, self.ng = mla, lev, int(nc), int(ng)
        self.nodal = tuple(int(x) for x in (nodal or (0, 0, 0)))
        if _dm == 2:
            self.nodal = self.nodal[:2] + (0,)
        nd = (C.c_int * 3)(*self.nodal)
        self.h = C.c_void_p()
        check(capi.load().vdn_multifab_create(mla.h, lev, self.nc, self.ng, nd, C.byref(self.h)))

    # -- BoxLib names ------------------------------------------------------------------------
    def nfabs(self):
        return capi.load().vdn_multifab_nfabs(self.h)

    def get_box(self, i):
        b = Box()
        check(capi.load().vdn_multifab_get_box(self.h, i, C.byref(b)))
        return tuple(b.lo), tuple(b.hi)

    def dataptr(self, i):
        p = C.c_void_p()
        check(capi.load().vdn_multifab_dataptr(self.h, i, C.byref(p)))
        return p.value

    def setval(self, val, comp=0, nc=None, all=False):
        check(capi.load().vdn_multifab_setval(self.h, float(val), comp, self.nc - comp if nc is None else nc, 1 if all else 0))

    def copy_c(self, dcomp, src, scomp, nc, ng=0):
        check(capi.load().vdn_multifab_copy_c(self.h, dcomp, src.h, scomp, nc, ng))

    def norm_inf(self, comp=0, nc=None):
        out = C.c_double()
        check(capi.load().vdn_multifab_norm_inf(self.h, comp, self.nc - comp if nc is None else nc, C.byref(out)))
        return out.value

    def min_max(self, comp=0):
        a, b = C.c_double(), C.c_double()
        check(capi.load().vdn_multifab_min_max(self.h, comp, C.byref(a), C.byref(b)))
        return a.value, b.value

    def fill_boundary(self):
        check(capi.load().vdn_multifab_fill_boundary(self.h))

    def physbc(self, scomp, bccomp, nc, bct):
        """multifab_physbc(s, start_scomp, start_bccomp, num_comp, bc) with 0-based components"""
        check(capi.load().vdn_multifab_physbc(self.h, scomp, bccomp, nc, bct.h))

    # -- host copies --------------------------------------------------------------------------
    def shape(self, i):
        lo, hi = self.get_box(i)
        gd = (self.ng, self.ng, self.ng if _dm == 3 else 0)      # dm = 2: the BoxLib 2-D layout, one z-plane
        return tuple(hi[d] - lo[d] + 1 + self.nodal[d] + 2 * gd[d] for d in range(3)) + (self.nc,)

    def to_numpy(self, i=0):
        a = np.empty(self.shape(i), dtype=np.float64, order="F")
        check(capi.load().vdn_multifab_copy_to_host(self.h, i, a.ctypes.data))
        return a

    def from_numpy(self, a, i=0):
        a = np.asfortranarray(a, dtype=np.float64)
        if a.shape != self.shape(i):
            raise ValueError("fab %d expects shape %r, got %r" % (i, self.shape(i), a.shape))
        check(capi.load().vdn_multifab_copy_from_host(self.h, i, a.ctypes.data))

    def destroy(self):
        if self.h:
            capi.load().vdn_multifab_destroy(self.h)
            self.h = None


def handle_array(mfs):
    arr = (C.c_void_p * len(mfs))()
    for i, m in enumerate(mfs):
        arr[i] = m.h
    return arr
