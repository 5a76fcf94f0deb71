"""ctypes loader + numpy helpers for the CPU oracle (oracle/libvoracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under varden_amd/ imports this module.  parity unpinned (see oracle/vo.h).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from varden_amd.capi import Params, default_params  # noqa: E402  (public POD only)

LIB = os.path.join(_HERE, "libvoracle.so")
VO_MAXCOMP = 16

PERIODIC, INTERIOR, INLET, OUTLET, SYMMETRY, SLIP_WALL, NO_SLIP_WALL = -1, 0, 11, 12, 13, 14, 15
REFLECT_ODD, REFLECT_EVEN, FOEXTRAP, EXT_DIR, HOEXTRAP = 20, 21, 22, 23, 24
BC_PER, BC_INT, BC_DIR, BC_NEU = -1, 0, 1, 2
INITIAL_PROJECTION, DIVU_ITERS, PRESSURE_ITERS, REGULAR_TIMESTEP = 1, 2, 3, 4


class CFab(C.Structure):
    _fields_ = [("p", C.c_void_p), ("lo", C.c_int * 3), ("hi", C.c_int * 3), ("ng", C.c_int),
                ("nd", C.c_int * 3), ("nc", C.c_int), ("n", C.c_long * 3), ("sc", C.c_long), ("gz", C.c_int), ("dm", C.c_int)]


class CBc(C.Structure):
    _fields_ = [("phys", (C.c_int * 2) * 3), ("adv", ((C.c_int * VO_MAXCOMP) * 2) * 3),
                ("ell", ((C.c_int * VO_MAXCOMP) * 2) * 3), ("ncomp_adv", C.c_int), ("ncomp_ell", C.c_int),
                ("press_comp", C.c_int), ("extrap_comp", C.c_int)]


class CMgStat(C.Structure):
    _fields_ = [("cycles", C.c_int), ("res0", C.c_double), ("res", C.c_double)]


class CState(C.Structure):
    _fields_ = [(k, CFab) for k in ("uold", "sold", "unew", "snew", "gp", "p", "ext_vel_force", "ext_scal_force")]


class CLevel(C.Structure):
    """vo_level (oracle/vo.h): one level of a hierarchy as a box list; built by vo_level_build"""
    _fields_ = [("nbox", C.c_int), ("boxes", C.POINTER(C.c_int)), ("blo", C.c_int * 3), ("bhi", C.c_int * 3), ("mg", C.c_int), ("valid", C.c_void_p)]


class Level:
    """a level = a list of boxes [(lo, hi), ...] in the level's own index space"""

    def __init__(self, boxes):
        self.boxes = [(tuple(int(x) for x in b[0]), tuple(int(x) for x in b[1])) for b in boxes]
        flat = []
        for lo, hi in self.boxes:
            flat += list(lo) + list(hi)
        self._flat = (C.c_int * len(flat))(*flat)
        self.c = CLevel()
        lib().vo_level_build(C.byref(self.c), len(self.boxes), self._flat)
        self.lo, self.hi = tuple(self.c.blo), tuple(self.c.bhi)

    def mask(self):
        """boolean array over the bounding box: True on the cells of the union"""
        m = np.zeros(tuple(self.hi[d] - self.lo[d] + 1 for d in range(3)), dtype=bool)
        for lo, hi in self.boxes:
            m[tuple(slice(lo[d] - self.lo[d], hi[d] - self.lo[d] + 1) for d in range(3))] = True
        return m

    def __del__(self):
        try:
            lib().vo_level_free(C.byref(self.c))
        except Exception:
            pass


def level_ptr_array(levels):
    arr = (C.POINTER(CLevel) * len(levels))()
    for i, l in enumerate(levels):
        arr[i] = C.pointer(l.c)
    return arr


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def set_threads(n):
    """OpenMP threads of the oracle from here on (omp_set_num_threads of the libgomp the process holds)"""
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except (OSError, AttributeError):
        pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        # libgomp sizes its pool from the visible CPUs, which on the GPU boxes is far more than the
        # cgroup share; oversubscribed spinning threads make the many tiny parallel regions crawl.  (With the count right the threads may spin:
        # OMP_WAIT_POLICY=passive costs the many small parallel regions a factor of four here.)
        os.environ.setdefault("OMP_NUM_THREADS", str(min(8, os.cpu_count() or 1)))
        _lib = C.CDLL(LIB)
        # a libgomp that was already in the process (torch brings one and initialises it on import, before the lines above could matter) has read its
        # environment long ago: tell it directly (round 5: the GPU suite collected a module that imports torch first, and the oracle ran on ~200 threads
        # of a 16-core share -- five to ten times slower)
        set_threads(int(os.environ["OMP_NUM_THREADS"]))
        _lib.vo_estdt.restype = C.c_double
        _lib.vo2_estdt.restype = C.c_double
        _lib.vo_cc_solve.restype = C.c_int
        _lib.vo_nd_solve.restype = C.c_int
        _lib.vo_estdt_g.restype = C.c_double
    return _lib


class Fab:
    """one BoxLib-layout fab backed by a numpy array ``a[i, j, k, c]`` (Fortran order).
    index 0 of each axis is lo-ng."""

    def __init__(self, lo, hi, ng=0, nc=1, nodal=(0, 0, 0), val=0.0, dm=3):
        """dm = 2: one z-plane, no ghost cells along z (the BoxLib 2-D layout p(lo1-ng:hi1+ng, lo2-ng:hi2+ng, nc))"""
        self.dm = int(dm)
        self.lo = tuple(int(x) for x in lo[:self.dm]) + (0,) * (3 - self.dm)
        self.hi = tuple(int(x) for x in hi[:self.dm]) + (0,) * (3 - self.dm)
        self.ng, self.nc = int(ng), int(nc)
        self.nodal = tuple(int(x) for x in nodal[:self.dm]) + (0,) * (3 - self.dm)
        self.gd = (self.ng, self.ng, self.ng if self.dm == 3 else 0)
        shp = tuple(self.hi[d] - self.lo[d] + 1 + self.nodal[d] + 2 * self.gd[d] for d in range(3)) + (self.nc,)
        self.a = np.full(shp, val, dtype=np.float64, order="F")
        self.c = CFab()
        self.c.p = self.a.ctypes.data
        for d in range(3):
            self.c.lo[d], self.c.hi[d], self.c.nd[d], self.c.n[d] = self.lo[d], self.hi[d], self.nodal[d], shp[d]
        self.c.ng, self.c.nc = self.ng, self.nc
        self.c.gz, self.c.dm = self.gd[2], self.dm
        self.c.sc = shp[0] * shp[1] * shp[2]

    @property
    def ref(self):
        return C.byref(self.c)

    def valid(self, grow=0):
        """view of the valid region (incl. nodal extra point), optionally grown"""
        sl = tuple(slice(self.gd[d] - min(grow, self.gd[d]), self.a.shape[d] - (self.gd[d] - min(grow, self.gd[d]))) for d in range(3))
        return self.a[sl]

    def like(self, val=0.0):
        return Fab(self.lo, self.hi, self.ng, self.nc, self.nodal, val, self.dm)

    def copy(self):
        f = self.like()
        f.a[...] = self.a
        return f


def fab_ptr_array(fabs):
    arr = (C.POINTER(CFab) * len(fabs))()
    for i, f in enumerate(fabs):
        arr[i] = C.pointer(f.c)
    return arr


def make_bc(phys, dm=3, nscal=2):
    """phys[d][side] -> CBc following define_bc_tower.f90"""
    bc = CBc()
    ph = ((C.c_int * 2) * 3)()
    for d in range(3):
        for s in range(2):
            ph[d][s] = int(phys[d][s])
    lib().vo_bc_build(C.byref(bc), ph, dm, nscal)
    return bc


def ellbc_of(bc):
    e = ((C.c_int * 2) * 3)()
    for d in range(3):
        for s in range(2):
            e[d][s] = bc.ell[d][s][bc.press_comp]
    return e


def dvec(x):
    return (C.c_double * 3)(*[float(v) for v in x])


def ivec(x):
    return (C.c_int * len(x))(*[int(v) for v in x])


class Sim:
    """the reference driver's single-level flow (src/varden.f90:108-345) on ONE box, on the CPU oracle.
    Used for the end-to-end parity tests and bench.py's cpu_baseline."""

    def __init__(self, n, phys, prm=None, prob_type=1, grav=-9.8, prob_hi=(1.0, 1.0, 1.0), init_shrink=1.0,
                 init_iter=4, do_initial_projection=1, dm=3):
        L = lib()
        self.dm = dm
        self.n = tuple(int(x) for x in (n if hasattr(n, "__len__") else (n,) * dm))
        if dm == 2:
            self.n = self.n[:2] + (1,)
        self.prm = prm or default_params()
        self.prm.prob_type = prob_type
        self.prm.dm = dm
        self.phys = [[int(phys[d][s]) for s in range(2)] if d < dm else [INTERIOR, INTERIOR] for d in range(3)]
        self.pmask = ivec([1 if self.phys[d][0] == PERIODIC else 0 for d in range(3)])
        self.bc = make_bc(self.phys, dm, self.prm.nscal)
        lo, hi = (0, 0, 0), tuple(x - 1 for x in self.n)
        self.dx = dvec([prob_hi[d] / self.n[d] for d in range(3)])
        ns = self.prm.nscal
        self.uold, self.sold = Fab(lo, hi, 3, dm, dm=dm), Fab(lo, hi, 3, ns, dm=dm)
        self.unew, self.snew = Fab(lo, hi, 3, dm, dm=dm), Fab(lo, hi, 3, ns, dm=dm)
        self.gp, self.p = Fab(lo, hi, 1, dm, dm=dm), Fab(lo, hi, 1, 1, (1, 1, 1), dm=dm)
        self.ext_vel_force, self.ext_scal_force = Fab(lo, hi, 1, dm, dm=dm), Fab(lo, hi, 1, ns, dm=dm)
        self.ext_vel_force.a[..., dm - 1] = grav                  # varden.f90:428-429
        self.init_shrink, self.init_iter = init_shrink, init_iter
        self.time, self.dt, self.istep = 0.0, 0.0, 0
        self.mgstat = (CMgStat * 2)()
        self.phase = (C.c_double * 4)()
        (L.vo2_initdata if dm == 2 else L.vo_initdata)(self.uold.ref, self.sold.ref, self.dx, prob_type)
        if do_initial_projection:                                 # varden.f90:126-138
            rhohalf = Fab(lo, hi, 1, 1, val=1.0, dm=dm)
            st = CMgStat()
            # fill ghosts first so that create_uvec/divu see the boundary data (initialize.f90 does this)
            self.fill_state_ghosts()
            (L.vo2_hgproject if dm == 2 else L.vo_hgproject)(
                           INITIAL_PROJECTION, self.uold.ref, self.uold.ref, rhohalf.ref, self.p.ref, self.gp.ref,
                           self.dx, C.c_double(1.0), C.byref(self.bc), self.pmask, C.byref(self.prm), C.byref(st))
            self.initial_projection_stat = (st.cycles, st.res0, st.res)
        self.p.a[...] = 0.0
        self.gp.a[...] = 0.0
        self.fill_state_ghosts()                                  # varden.f90:165-178
        self.unew.a[...] = self.uold.a
        self.snew.a[...] = self.sold.a
        self.dt = self.estdt(1.0e20) * init_shrink                # varden.f90:186-194
        for _ in range(init_iter):                                # varden.f90:460-490
            self.advance(PRESSURE_ITERS)

    def fill_state_ghosts(self):
        L = lib()
        L.vo_fill_boundary(self.uold.ref, self.pmask)
        L.vo_fill_boundary(self.sold.ref, self.pmask)
        L.vo_fill_boundary(self.gp.ref, self.pmask)
        L.vo_physbc(self.uold.ref, 0, 0, self.dm, C.byref(self.bc), C.byref(self.prm))
        L.vo_physbc(self.sold.ref, 0, self.dm, self.prm.nscal, C.byref(self.bc), C.byref(self.prm))

    def estdt(self, dtold):
        return (lib().vo2_estdt if self.dm == 2 else lib().vo_estdt)(self.uold.ref, self.sold.ref, self.gp.ref, self.ext_vel_force.ref, self.dx,
                              C.c_double(dtold), C.byref(self.prm))

    def state(self):
        S = CState()
        for k in ("uold", "sold", "unew", "snew", "gp", "p", "ext_vel_force", "ext_scal_force"):
            setattr(S, k, getattr(self, k).c)
        return S

    def advance(self, proj_type=REGULAR_TIMESTEP):
        S = self.state()
        (lib().vo2_advance_timestep if self.dm == 2 else lib().vo_advance_timestep)(C.byref(S), self.dx, C.c_double(self.dt), C.byref(self.bc), self.pmask,
                                  C.byref(self.prm), proj_type, self.mgstat, self.phase)

    def step(self):
        """one pass of the time loop body, varden.f90:291-328"""
        self.istep += 1
        self.fill_state_ghosts()
        if self.istep > 1:
            self.dt = self.estdt(self.dt)
        self.advance(REGULAR_TIMESTEP)
        self.uold.valid()[...] = self.unew.valid()                                # copy_c valid only
        self.sold.valid()[...] = self.snew.valid()
        self.time += self.dt


class SimML:
    """multi-level (fixed properly nested grids) bubble run on the CPU oracle: what src/varden.f90's time loop does around advance_timestep for
    nlevs > 1 (ghost fills by ml_restrict_and_fill, dt = min over levels and boxes of estdt).
    boxes: per refined level either ONE box (lo, hi) or a LIST of boxes [(lo, hi), ...], in the level's own index space.  The fields of a level are
    level arrays over the bounding box of its boxes (oracle/vo.h); `levels[n].mask()` marks the cells that belong to the level."""

    def __init__(self, nc, boxes, phys, prm=None, prob_type=1, grav=-9.8, init_shrink=0.1, init_iter=0, do_initial_projection=0, base_boxes=None, init_fn=None, grav_dir=2):
        """nc: cells of level 0 per direction, an int (a cube) or three ints (dx = dy = dz = 1 / nc[0]: the z-uniform copies of 2-D problems, varden_amd/driver.py: extrude2d;
        grav_dir = 1 puts gravity along y there).  base_boxes: level 0 cut into boxes (what max_grid_size makes of it); default: one box.
        init_fn(level, lo, shape, dx) -> (u, s) with 3 ghost layers over the level array replaces the analytic initial data (as varden_amd/driver.py: VardenAMR)"""
        L = lib()
        self.prm = prm or default_params()
        self.prm.prob_type = prob_type
        self.nc, self.phys, self.grav, self.grav_dir = nc, phys, grav, int(grav_dir)
        self.ncs = tuple(int(x) for x in nc) if hasattr(nc, "__len__") else (int(nc),) * 3
        self.base_boxes = [((0, 0, 0), tuple(c - 1 for c in self.ncs))] if base_boxes is None else [(tuple(b[0]), tuple(b[1])) for b in base_boxes]
        self._set_grids(boxes)
        NL, ns = self.nlev, self.prm.nscal
        los, his = [lv.lo for lv in self.levels], [lv.hi for lv in self.levels]
        self.uold, self.sold, self.gp, self.p = self._mk(3, 3), self._mk(3, ns), self._mk(1, 3), self._mk(1, 1, (1, 1, 1))
        self._alloc_temps()
        for n in range(NL):
            L.vo_initdata(self.uold[n].ref, self.sold[n].ref, dvec(self.dxl[n]), prob_type)
            if init_fn is not None:
                ub, sb = init_fn(n, los[n], tuple(his[n][d] - los[n][d] + 1 for d in range(3)), self.dxl[n])
                self.uold[n].a[...] = ub
                self.sold[n].a[...] = sb
        self.mgstat = (CMgStat * 2)()
        self.time, self.istep = 0.0, 0
        self.fill_state_ghosts()
        if do_initial_projection:                                 # varden.f90:126-138
            rhohalf = [Fab(los[n], his[n], 1, 1, val=1.0) for n in range(NL)]
            st = CMgStat()
            L.vo_ml_hgproject_g(NL, self.lev, INITIAL_PROJECTION, fab_ptr_array(self.uold), fab_ptr_array(self.uold), fab_ptr_array(rhohalf), fab_ptr_array(self.p),
                                fab_ptr_array(self.gp), self.dx, C.c_double(1.0), self.bcs, self.pmask, self.pd, C.byref(self.prm), C.byref(st))
            self.initial_projection_stat = (st.cycles, st.res0, st.res)
            for n in range(NL):
                self.p[n].a[...] = 0.0
                self.gp[n].a[...] = 0.0
            self.fill_state_ghosts()                              # varden.f90:165-178
        for n in range(NL):
            self.unew[n].a[...] = self.uold[n].a
            self.snew[n].a[...] = self.sold[n].a
        self.dt = self.estdt(1.0e20) * init_shrink
        for _ in range(init_iter):                                # varden.f90:460-490
            self._advance(PRESSURE_ITERS)

    def _set_grids(self, boxes):
        """the box lists of the refined levels -> levels, bc tables, domains, spacings"""
        ncs, phys, ns = self.ncs, self.phys, self.prm.nscal
        self.nlev = NL = 1 + len(boxes)
        blists = [list(self.base_boxes)]
        for b in boxes:
            blists.append([b] if (len(b) == 2 and not hasattr(b[0][0], "__len__")) else list(b))
        self.levels = [Level(bl_) for bl_ in blists]
        self.lev = level_ptr_array(self.levels)
        los, his = [lv.lo for lv in self.levels], [lv.hi for lv in self.levels]
        bcl, pd, self.dxl = [make_bc(phys, 3, ns)], [0, 0, 0, ncs[0] - 1, ncs[1] - 1, ncs[2] - 1], [[1.0 / ncs[0]] * 3]
        for n in range(1, NL):
            nd = [c << n for c in ncs]
            bcl.append(make_bc([[phys[d][0] if los[n][d] == 0 else INTERIOR, phys[d][1] if his[n][d] == nd[d] - 1 else INTERIOR] for d in range(3)], 3, ns))
            pd += [0, 0, 0, nd[0] - 1, nd[1] - 1, nd[2] - 1]
            self.dxl.append([1.0 / nd[0]] * 3)
        self.bcs = (CBc * NL)(*bcl)
        self.pmask = ivec([1 if phys[d][0] == PERIODIC else 0 for d in range(3)])      # (round 6: level 0 wraps; refined levels must stay clear of the periodic faces)
        self.pd = ivec(pd)
        self.dx = (C.c_double * (3 * NL))(*sum(self.dxl, []))

    def _mk(self, ng, ncomp, nodal=(0, 0, 0)):
        return [Fab(self.levels[n].lo, self.levels[n].hi, ng, ncomp, nodal) for n in range(self.nlev)]

    def _alloc_temps(self):
        ns = self.prm.nscal
        self.unew, self.snew = self._mk(3, 3), self._mk(3, ns)
        self.ext_vel_force, self.ext_scal_force = self._mk(1, 3), self._mk(1, ns)
        for n in range(self.nlev):
            self.ext_vel_force[n].a[..., self.grav_dir] = self.grav

    def node_mask(self, n):
        """nodes of level n's array that a cell of the level touches"""
        m = self.levels[n].mask()
        out = np.zeros(tuple(x + 1 for x in m.shape), dtype=bool)
        for c in (0, 1):
            for b in (0, 1):
                for a in (0, 1):
                    out[a:a + m.shape[0], b:b + m.shape[1], c:c + m.shape[2]] |= m
        return out

    def regrid(self, boxes):
        """the state moved onto NEW box lists of the refined levels -- build_and_fill_data of src/regrid.f90:269-339, coarsest level first: the ghost cells
        of the levels below filled, every cell of the new level interpolated from the level below (fillpatch), the pressure prolonged node by node
        (ml_nodal_prolongation), then the old level's data copied wherever the old level had cells (multifab_copy_c between the box lists).  The grids
        themselves (tag_boxes + make_new_grids) come from the caller."""
        L = lib()
        old_levels, old_nlev = self.levels, self.nlev
        old = dict(uold=self.uold, sold=self.sold, gp=self.gp, p=self.p)
        old_nmask = [self.node_mask(n) for n in range(old_nlev)]
        self._set_grids(boxes)
        ns = self.prm.nscal
        new = dict(uold=self._mk(3, 3), sold=self._mk(3, ns), gp=self._mk(1, 3), p=self._mk(1, 1, (1, 1, 1)))
        for k in new:
            new[k][0].a[...] = old[k][0].a
        for n in range(1, self.nlev):
            for k, ic, bc_, nc_, same in (("uold", 0, 0, 3, 0), ("sold", 0, 3, ns, 0), ("gp", 0, 3 + ns + 1, 3, 1)):
                L.vo_ml_restrict_and_fill_g(n, self.lev, fab_ptr_array(new[k][:n]), ic, bc_, nc_, same, self.bcs, self.pmask, self.pd, C.byref(self.prm))
                L.vo_fillpatch(new[k][n].ref, new[k][n - 1].ref, 0, nc_)
            L.vo_nodal_prolongation(new["p"][n].ref, new["p"][n - 1].ref)
            if n < old_nlev:                                     # the old level's data where it had cells (nodes: where a cell of it touches)
                ol, nl = old_levels[n], self.levels[n]
                lo = [max(ol.lo[d], nl.lo[d]) for d in range(3)]; hi = [min(ol.hi[d], nl.hi[d]) for d in range(3)]
                if all(lo[d] <= hi[d] for d in range(3)):
                    so = tuple(slice(lo[d] - ol.lo[d], hi[d] - ol.lo[d] + 1) for d in range(3))
                    sn = tuple(slice(lo[d] - nl.lo[d], hi[d] - nl.lo[d] + 1) for d in range(3))
                    both = ol.mask()[so] & nl.mask()[sn]
                    for k in ("uold", "sold", "gp"):
                        dst, src = new[k][n].valid(), old[k][n].valid()
                        dst[sn][both] = src[so][both]
                    so_n = tuple(slice(lo[d] - ol.lo[d], hi[d] - ol.lo[d] + 2) for d in range(3))
                    sn_n = tuple(slice(lo[d] - nl.lo[d], hi[d] - nl.lo[d] + 2) for d in range(3))
                    bothn = old_nmask[n][so_n] & self.node_mask(n)[sn_n]
                    new["p"][n].valid()[sn_n][bothn] = old["p"][n].valid()[so_n][bothn]
        self.uold, self.sold, self.gp, self.p = new["uold"], new["sold"], new["gp"], new["p"]
        self._alloc_temps()
        self.fill_state_ghosts()                                  # regrid.f90:252-254
        for n in range(self.nlev):
            self.unew[n].a[...] = self.uold[n].a
            self.snew[n].a[...] = self.sold[n].a

    def _rf(self, mfs, icomp, bcomp, nc, same=0):
        lib().vo_ml_restrict_and_fill_g(self.nlev, self.lev, fab_ptr_array(mfs), icomp, bcomp, nc, same, self.bcs, self.pmask, self.pd, C.byref(self.prm))

    def fill_state_ghosts(self):
        self._rf(self.uold, 0, 0, 3)
        self._rf(self.sold, 0, 3, self.prm.nscal)
        self._rf(self.gp, 0, 3 + self.prm.nscal + 1, 3, 1)          # extrap_comp, same_boundary

    def estdt(self, dtold):
        L = lib()
        return min(L.vo_estdt_g(self.lev[n], self.uold[n].ref, self.sold[n].ref, self.gp[n].ref, self.ext_vel_force[n].ref, dvec(self.dxl[n]),
                                C.c_double(dtold), C.byref(self.prm)) for n in range(self.nlev))

    def _advance(self, proj_type):
        S = (CState * self.nlev)()
        for n in range(self.nlev):
            for k in ("uold", "sold", "unew", "snew", "gp", "p", "ext_vel_force", "ext_scal_force"):
                setattr(S[n], k, getattr(self, k)[n].c)
        lib().vo_ml_advance_timestep_g(self.nlev, self.lev, S, self.dx, C.c_double(self.dt), self.bcs, self.pmask, self.pd, C.byref(self.prm), proj_type, self.mgstat)

    def step(self):
        self.istep += 1
        self.fill_state_ghosts()
        if self.istep > 1:
            self.dt = self.estdt(self.dt)
        self._advance(REGULAR_TIMESTEP)
        for n in range(self.nlev):                              # (cells outside the level are overwritten by the next ghost fill)
            self.uold[n].valid()[...] = self.unew[n].valid()
            self.sold[n].valid()[...] = self.snew[n].valid()
        self.time += self.dt


class Sim2L(SimML):
    """two levels: SimML with one fine box"""

    def __init__(self, nc, flo, fhi, phys, **kw):
        SimML.__init__(self, nc, [(flo, fhi)], phys, **kw)
